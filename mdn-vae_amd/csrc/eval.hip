// Evaluation-loop companions: the global maximum of the real matrices, division by it with the
// clamp at zero, and per-matrix mean squared error.
//   reference: function/VAE_CNN_NN.py:268-271 (max_value, /max_value, np.maximum(.,0)) and
//   :273-289 (compute_mse / compute_psnr's per-matrix mse_values).
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

constexpr int MAX_PARTS = 1024;

__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max_all(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float r = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    __syncthreads();
    return r;
}

// 16-byte loads at 4-byte alignment (the hardware's unaligned access mode): a view that starts at any
// float of a larger tensor (a test split of 87 x 87 rows) is read as it is.
typedef float max_v4 __attribute__((ext_vector_type(4)));
typedef max_v4 max_v4u __attribute__((aligned(4)));

// np.max semantics: a NaN anywhere makes the maximum NaN (fmaxf alone would drop it).
__global__ __launch_bounds__(256) void max_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
    __shared__ float scratch[4];
    float m = -INFINITY;
    bool nan = false;
    const long n4 = n >> 2;
    const max_v4u* x4 = reinterpret_cast<const max_v4u*>(x);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const max_v4 v = x4[i];
        nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
        m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        nan |= v != v;
        m = fmaxf(m, v);
    }
    m = block_max(m, scratch);
    const int any_nan = __syncthreads_or(nan);
    if (threadIdx.x == 0) part[blockIdx.x] = any_nan ? NAN : m;
}
__global__ __launch_bounds__(256) void max_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
    __shared__ float scratch[4];
    float m = -INFINITY;
    bool nan = false;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        const float v = part[i];
        nan |= v != v;
        m = fmaxf(m, v);
    }
    m = block_max(m, scratch);
    const int any_nan = __syncthreads_or(nan);
    if (threadIdx.x == 0) out[0] = any_nan ? NAN : m;
}

// out = x / denom[0], optionally clamped from below at `lo`
__global__ __launch_bounds__(256) void div_clamp_kernel(const float* __restrict__ x, const float* __restrict__ denom,
                                                        int clamp, float lo, long n, float* __restrict__ out) {
    const float d = denom[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = x[i] / d;
        if (clamp) v = fmaxf(v, lo);
        out[i] = v;
    }
}

// out[r] = mean_c (a[r][c] - b[r][c])^2, one workgroup per matrix, double accumulation
__global__ __launch_bounds__(256) void row_mse_kernel(const float* __restrict__ a, const float* __restrict__ b, int cols,
                                                      float* __restrict__ out) {
    __shared__ double scratch[4];
    const long base = (long)blockIdx.x * cols;
    double s = 0.0;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float d = a[base + c] - b[base + c];
        s += (double)d * (double)d;
    }
    s = block_sum<256, double>(s, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(s / cols);
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

size_t mdnvae_max_workspace_floats(void) { return MAX_PARTS; }

int mdnvae_max(const float* x, long n, float* workspace, float* out, void* stream) {
    MV_REQUIRE(n >= 1, E_ARG, "mdnvae_max: empty input has no maximum");
    MV_REQUIRE(x && workspace && out, E_ARG, "mdnvae_max: null pointer");
    MV_REQUIRE((reinterpret_cast<uintptr_t>(x) & 3u) == 0, E_ALIGN, "mdnvae_max: input must be 4-byte aligned");
    long g = (n / 4 + 255) / 256;
    const int parts = (int)(g > MAX_PARTS ? MAX_PARTS : (g < 1 ? 1 : g));
    hipLaunchKernelGGL(max_partial_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, x, n, workspace);
    MV_LAUNCH_CHECK("max_partial_kernel");
    hipLaunchKernelGGL(max_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, parts, out);
    MV_LAUNCH_CHECK("max_final_kernel");
    return 0;
}

int mdnvae_div_clamp(const float* x, const float* denom, int clamp, float lo, long n, float* out, void* stream) {
    MV_REQUIRE(n >= 0, E_ARG, "mdnvae_div_clamp: negative size");
    if (n == 0) return 0;
    MV_REQUIRE(x && denom && out, E_ARG, "mdnvae_div_clamp: null pointer");
    long g = (n + 255) / 256;
    hipLaunchKernelGGL(div_clamp_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, x,
                       denom, clamp, lo, n, out);
    MV_LAUNCH_CHECK("div_clamp_kernel");
    return 0;
}

int mdnvae_row_mse(const float* a, const float* b, long rows, int cols, float* out, void* stream) {
    MV_REQUIRE(rows >= 0 && cols >= 1, E_ARG, "mdnvae_row_mse: bad shape");
    MV_REQUIRE(rows <= 0x7fffffffL, E_SHAPE, "mdnvae_row_mse: too many rows for one launch");
    if (rows == 0) return 0;
    MV_REQUIRE(a && b && out, E_ARG, "mdnvae_row_mse: null pointer");
    hipLaunchKernelGGL(row_mse_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a, b, cols, out);
    MV_LAUNCH_CHECK("row_mse_kernel");
    return 0;
}

}  // extern "C"
