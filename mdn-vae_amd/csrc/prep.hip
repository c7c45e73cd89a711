// Connectome preparation on the device: one raw n x n matrix per subject ->
//   m = raw + raw^T, diagonal := column means of m, crop [lo, hi) on both axes, scale, flatten.
//   reference loops: function/VAE_CNN_NN.py:32-39 (counts / offset), function_FC/VAE_CNN_GCN.py:35-43
//   ((x+1)/2), function/Full_NN.py:106-112 (log(x+1), no crop), function_fc_new/ResNet.py:17-19,33-41
//   (Fisher z with the clip at 1 - 1e-7).  MDNVAE_PREP_RAW skips the symmetrisation and the diagonal
//   (the group-mean matrix A_mat of function/VAE_CNN_NN.py:21 is a mean of raw crops).
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

__global__ __launch_bounds__(256) void prep_connectome_kernel(const float* __restrict__ raw, int n, int lo, int hi,
                                                              int mode, float offset, float* __restrict__ out) {
    extern __shared__ float colmean[];                 // [n]
    const int b = blockIdx.x, t = threadIdx.x;
    const float* r = raw + (long)b * n * n;
    const int w = hi - lo;
    if (mode != MDNVAE_PREP_RAW) {
        // column j of m = raw + raw^T sums to (column j of raw) + (row j of raw); rows in order, fp32
        for (int j = t; j < n; j += 256) {
            float s = 0.0f;
            for (int i = 0; i < n; ++i) s += r[i * n + j] + r[j * n + i];
            colmean[j] = s / (float)n;
        }
        __syncthreads();
    }
    float* o = out + (long)b * w * w;
    for (int e = t; e < w * w; e += 256) {
        const int i = lo + e / w, j = lo + e % w;
        float v;
        if (mode == MDNVAE_PREP_RAW) v = r[i * n + j];
        else v = (i == j) ? colmean[j] : r[i * n + j] + r[j * n + i];
        if (mode == MDNVAE_PREP_DIVIDE) v = v / offset;
        else if (mode == MDNVAE_PREP_UNIT) v = (v + 1.0f) / 2.0f;
        else if (mode == MDNVAE_PREP_LOG1P) v = logf(v + 1.0f);
        else if (mode == MDNVAE_PREP_FISHER) {
            const float lim = 1.0f - 1e-7f;
            v = fminf(fmaxf(v, -lim), lim);
            v = 0.5f * logf((1.0f + v) / (1.0f - v));
        }
        o[e] = v;
    }
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" int mdnvae_prep_connectome(const float* raw, long B, int n, int crop_lo, int crop_hi, int mode, float offset,
                                      float* out, void* stream) {
    MV_REQUIRE(B >= 0 && n >= 1, E_ARG, "mdnvae_prep_connectome: bad size");
    MV_REQUIRE(0 <= crop_lo && crop_lo < crop_hi && crop_hi <= n, E_ARG, "mdnvae_prep_connectome: crop [%d,%d) of %d",
               crop_lo, crop_hi, n);
    MV_REQUIRE(mode >= MDNVAE_PREP_DIVIDE && mode <= MDNVAE_PREP_RAW, E_ARG, "mdnvae_prep_connectome: unknown mode %d",
               mode);
    MV_REQUIRE(mode != MDNVAE_PREP_DIVIDE || offset != 0.0f, E_ARG, "mdnvae_prep_connectome: zero offset");
    MV_REQUIRE(n <= 8192 && B <= 0x7fffffffL, E_SHAPE, "mdnvae_prep_connectome: matrix or batch too large");
    if (B == 0) return 0;
    MV_REQUIRE(raw && out, E_ARG, "mdnvae_prep_connectome: null pointer");
    hipLaunchKernelGGL(prep_connectome_kernel, dim3((unsigned)B), dim3(256), sizeof(float) * n, (hipStream_t)stream, raw,
                       n, crop_lo, crop_hi, mode, offset, out);
    MV_LAUNCH_CHECK("prep_connectome_kernel");
    return 0;
}
