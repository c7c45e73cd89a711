// Batched evaluation metrics on square connectome matrices held one per workgroup in LDS.
//   spectrum: eigenvalues of a symmetric n x n matrix by cyclic Jacobi rotations in the parallel
//     (round-robin) ordering, sorted in descending order, top k kept
//       reference: compute_spectral_features, function/VAE_CNN_NN.py:302-308 (np.linalg.eigvals, real
//       part, sort descending, [:k]) -- equal to it for symmetric input, which is what the scripts feed
//       for real matrices and for the symmetrised decoders;
//   ssim: mean structural similarity, 7x7 uniform window, sample covariance, K1 = 0.01, K2 = 0.03,
//     data_range given (the defaults of the skimage call at function/VAE_CNN_NN.py:291-297).
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

constexpr int EIG_MAXN = 96;
constexpr int EIG_NT = 256;
constexpr int EIG_SWEEPS = 16;

__global__ __launch_bounds__(EIG_NT) void jacobi_spectrum_kernel(const float* __restrict__ mats, int n, int k,
                                                                 float* __restrict__ out) {
    extern __shared__ float smem[];
    const int ld = n + 1;
    float* A = smem;                         // [n][ld]
    float* cs = A + n * ld;                  // [m/2][2] rotation of each pair of the round
    __shared__ int pp[EIG_MAXN / 2], qq[EIG_MAXN / 2];
    __shared__ float red[EIG_NT / 64];
    __shared__ float off_sq, frob_sq;
    const int t = threadIdx.x;
    const float* src = mats + (long)blockIdx.x * n * n;
    for (int e = t; e < n * n; e += EIG_NT) {
        const int i = e / n, j = e - i * n;
        A[i * ld + j] = 0.5f * (src[e] + src[j * n + i]);       // exact for symmetric input
    }
    const int m = (n + 1) & ~1;              // players of the tournament (one dummy when n is odd)
    const int half = m / 2;
    __syncthreads();
    {
        float s = 0.0f;
        for (int e = t; e < n * n; e += EIG_NT) {
            const float v = A[(e / n) * ld + e % n];
            s += v * v;
        }
        s = block_sum<EIG_NT, float>(s, red);
        if (t == 0) frob_sq = s;
        __syncthreads();
    }
    for (int sweep = 0; sweep < EIG_SWEEPS; ++sweep) {
        for (int r = 0; r < m - 1; ++r) {
            // round r of the round-robin: (m-1, r) and ((r+i) mod (m-1), (r-i) mod (m-1)), i = 1..half-1
            if (t < half) {
                int p, q;
                if (t == 0) {
                    p = m - 1;
                    q = r;
                } else {
                    p = (r + t) % (m - 1);
                    q = (r - t + (m - 1)) % (m - 1);
                }
                if (p > q) { const int x = p; p = q; q = x; }
                float c = 1.0f, s = 0.0f;
                if (q < n) {                                     // a pair with the dummy player rests
                    const float apq = A[p * ld + q];
                    if (fabsf(apq) > 1e-30f) {
                        const float tau = (A[q * ld + q] - A[p * ld + p]) / (2.0f * apq);
                        const float tt = (tau >= 0.0f ? 1.0f : -1.0f) / (fabsf(tau) + sqrtf(1.0f + tau * tau));
                        c = 1.0f / sqrtf(1.0f + tt * tt);
                        s = tt * c;
                    }
                } else {
                    q = p;                                       // identity on (p, p)
                }
                pp[t] = p;
                qq[t] = q;
                cs[2 * t] = c;
                cs[2 * t + 1] = s;
            }
            __syncthreads();
            // A <- A J : columns p, q of every row
            for (int w = t; w < n * half; w += EIG_NT) {
                const int i = w / half, pr = w - i * half;
                const int p = pp[pr], q = qq[pr];
                if (p != q) {
                    const float c = cs[2 * pr], s = cs[2 * pr + 1];
                    const float ap = A[i * ld + p], aq = A[i * ld + q];
                    A[i * ld + p] = c * ap - s * aq;
                    A[i * ld + q] = s * ap + c * aq;
                }
            }
            __syncthreads();
            // A <- J^T A : rows p, q of every column
            for (int w = t; w < n * half; w += EIG_NT) {
                const int pr = w / n, j = w - pr * n;
                const int p = pp[pr], q = qq[pr];
                if (p != q) {
                    const float c = cs[2 * pr], s = cs[2 * pr + 1];
                    const float ap = A[p * ld + j], aq = A[q * ld + j];
                    A[p * ld + j] = c * ap - s * aq;
                    A[q * ld + j] = s * ap + c * aq;
                }
            }
            __syncthreads();
        }
        float s = 0.0f;
        for (int e = t; e < n * n; e += EIG_NT) {
            const int i = e / n, j = e - i * n;
            if (i != j) s += A[i * ld + j] * A[i * ld + j];
        }
        s = block_sum<EIG_NT, float>(s, red);
        if (t == 0) off_sq = s;
        __syncthreads();
        if (off_sq <= 1e-14f * frob_sq) break;                   // uniform over the workgroup
    }
    // descending rank of each diagonal entry (ties broken by index)
    for (int i = t; i < n; i += EIG_NT) {
        const float v = A[i * ld + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float u = A[j * ld + j];
            rank += (u > v) || (u == v && j < i);
        }
        if (rank < k) out[(long)blockIdx.x * k + rank] = v;
    }
}

// one workgroup per image pair; both images in LDS; each thread walks valid window positions
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                   float data_range, float* __restrict__ out) {
    extern __shared__ float smem[];
    __shared__ double red[4];
    constexpr int WIN = 7;
    float* ia = smem;
    float* ib = smem + n * n;
    const int t = threadIdx.x;
    for (int e = t; e < n * n; e += 256) {
        ia[e] = a[(long)blockIdx.x * n * n + e];
        ib[e] = b[(long)blockIdx.x * n * n + e];
    }
    __syncthreads();
    const int v = n - WIN + 1;                                   // valid positions per axis
    const float np_ = (float)(WIN * WIN), cov_norm = np_ / (np_ - 1.0f);
    const float c1 = (0.01f * data_range) * (0.01f * data_range), c2 = (0.03f * data_range) * (0.03f * data_range);
    double acc = 0.0;
    for (int e = t; e < v * v; e += 256) {
        const int y0 = e / v, x0 = e - y0 * v;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
        for (int dy = 0; dy < WIN; ++dy)
#pragma unroll
            for (int dx = 0; dx < WIN; ++dx) {
                const float p = ia[(y0 + dy) * n + x0 + dx], q = ib[(y0 + dy) * n + x0 + dx];
                sx += p; sy += q; sxx += p * p; syy += q * q; sxy += p * q;
            }
        const float ux = sx / np_, uy = sy / np_;
        const float vx = cov_norm * (sxx / np_ - ux * ux), vy = cov_norm * (syy / np_ - uy * uy);
        const float vxy = cov_norm * (sxy / np_ - ux * uy);
        const float s = ((2.0f * ux * uy + c1) * (2.0f * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
        acc += (double)s;
    }
    acc = block_sum<256, double>(acc, red);
    if (t == 0) out[blockIdx.x] = (float)(acc / ((double)v * v));
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

int mdnvae_spectrum_topk(const float* mats, long count, int n, int k, float* out, void* stream) {
    MV_REQUIRE(count >= 0 && n >= 2 && k >= 1 && k <= n, E_ARG, "mdnvae_spectrum_topk: bad size (n=%d, k=%d)", n, k);
    MV_REQUIRE(n <= EIG_MAXN, E_SHAPE, "mdnvae_spectrum_topk: n=%d exceeds %d", n, EIG_MAXN);
    MV_REQUIRE(count <= 0x7fffffffL, E_SHAPE, "mdnvae_spectrum_topk: too many matrices for one launch");
    if (count == 0) return 0;
    MV_REQUIRE(mats && out, E_ARG, "mdnvae_spectrum_topk: null pointer");
    const size_t lds = sizeof(float) * ((size_t)n * (n + 1) + (size_t)((n + 1) & ~1));
    hipLaunchKernelGGL(jacobi_spectrum_kernel, dim3((unsigned)count), dim3(EIG_NT), lds, (hipStream_t)stream, mats, n, k,
                       out);
    MV_LAUNCH_CHECK("jacobi_spectrum_kernel");
    return 0;
}

int mdnvae_ssim(const float* a, const float* b, long count, int n, float data_range, float* out, void* stream) {
    MV_REQUIRE(count >= 0 && n >= 7, E_ARG, "mdnvae_ssim: images must be at least 7x7 (n=%d)", n);
    MV_REQUIRE(n <= 128, E_SHAPE, "mdnvae_ssim: n=%d exceeds 128", n);
    MV_REQUIRE(data_range > 0.0f, E_ARG, "mdnvae_ssim: data_range must be positive");
    MV_REQUIRE(count <= 0x7fffffffL, E_SHAPE, "mdnvae_ssim: too many pairs for one launch");
    if (count == 0) return 0;
    MV_REQUIRE(a && b && out, E_ARG, "mdnvae_ssim: null pointer");
    static bool attr_set = false;
    if (!attr_set) {
        // two 128 x 128 images; the kernel's static LDS comes on top of this
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssim_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_ssim: cannot raise the LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(ssim_kernel, dim3((unsigned)count), dim3(256), sizeof(float) * 2 * n * n, (hipStream_t)stream, a, b,
                       n, data_range, out);
    MV_LAUNCH_CHECK("ssim_kernel");
    return 0;
}

}  // extern "C"
