// Elementwise / reduction companions of the library GEMMs used for the plain dense layers
// (FNN encoder towers function/MM_Gate.py:157-162, cVAE function_FC_SC/SC-FC.py:95-113, the two
// latent heads function/VAE_CNN_NN.py:160-161 and the dense intercept :187): bias + activation in
// place after the GEMM, the activation's backward, and fixed-order column sums for bias gradients.
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

constexpr int A_NONE = 0, A_SIGMOID = 1, A_TANH = 2, A_RELU = 3;

__device__ __forceinline__ float dact(int act, float v) {
    if (act == A_SIGMOID) return sigmoidf_(v);
    if (act == A_TANH) return tanhf(v);
    if (act == A_RELU) return fmaxf(v, 0.0f);
    return v;
}
__device__ __forceinline__ float dslope(int act, float h) {
    if (act == A_SIGMOID) return h * (1.0f - h);
    if (act == A_TANH) return 1.0f - h * h;
    if (act == A_RELU) return h > 0.0f ? 1.0f : 0.0f;
    return 1.0f;
}

// y[r][c] = act(y[r][c] + bias[c])
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias, long rows,
                                                       int cols, int act) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cols);
        float v = y[i];
        if (bias) v += bias[c];
        y[i] = dact(act, v);
    }
}

// g[i] *= act'(y[i])   (y = activation output)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gin,
                                                      float* __restrict__ gout, long n, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        gout[i] = gin[i] * dslope(act, y[i]);
}

// y[i] = act(a[i] + b[i])   (residual joins: function_fc_new/ResNet.py:101-107,150-153)
__global__ __launch_bounds__(256) void add_act_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ y, long n, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = dact(act, a[i] + b[i]);
}

// Column sums in a fixed order.  A workgroup is 4 row groups x 64 columns: row group g adds rows
// g, g+4, g+8, ... of its slab (eight independent loads in flight), the four partial rows meet in
// LDS.  colsum_partial writes one row of partial sums per slab of `rows_per_block` rows; the same
// kernel shape then folds the partial rows into the result.
constexpr int CS_COLS = 64, CS_GROUPS = 4, CS_UNROLL = 8;
__device__ __forceinline__ float colsum_slab(const float* __restrict__ a, int r0, int r1, int D, int c, int g,
                                             float (*sm)[CS_COLS]) {
    float s = 0.f;
    if (c < D) {
        int r = r0 + g;
        for (; r + (CS_UNROLL - 1) * CS_GROUPS < r1; r += CS_UNROLL * CS_GROUPS) {
            float v[CS_UNROLL];
#pragma unroll
            for (int u = 0; u < CS_UNROLL; ++u) v[u] = a[(long)(r + u * CS_GROUPS) * D + c];
#pragma unroll
            for (int u = 0; u < CS_UNROLL; ++u) s += v[u];
        }
        for (; r < r1; r += CS_GROUPS) s += a[(long)r * D + c];
    }
    sm[g][threadIdx.x & (CS_COLS - 1)] = s;
    __syncthreads();
    const int t = threadIdx.x & (CS_COLS - 1);
    return (sm[0][t] + sm[1][t]) + (sm[2][t] + sm[3][t]);
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ a, int B, int D,
                                                             int rows_per_block, float* __restrict__ part) {
    __shared__ float sm[CS_GROUPS][CS_COLS];
    const int c = blockIdx.x * CS_COLS + (threadIdx.x & (CS_COLS - 1)), g = threadIdx.x >> 6;
    const int b0 = blockIdx.y * rows_per_block;
    const float s = colsum_slab(a, b0, min(B, b0 + rows_per_block), D, c, g, sm);
    if (g == 0 && c < D) part[(long)blockIdx.y * D + c] = s;
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nparts, int D,
                                                           int accumulate, float* __restrict__ out) {
    __shared__ float sm[CS_GROUPS][CS_COLS];
    const int c = blockIdx.x * CS_COLS + (threadIdx.x & (CS_COLS - 1)), g = threadIdx.x >> 6;
    const float s = colsum_slab(part, 0, nparts, D, c, g, sm);
    if (g == 0 && c < D) out[c] = accumulate ? out[c] + s : s;
}

// ---- narrow dense layers: at most NL_MAX inputs and NL_MAX outputs ----
// (the gate of the trait-gated mixture head: a handful of traits -> K logits.)  A library GEMM
// spends 150 us on the [5 x 4] weight gradient of such a layer over a batch of 16384 - its tiles
// are 16 wide at the least and the reduction runs over the batch in one workgroup column.  Here a
// thread takes samples, the whole weight lives in LDS (forward) or in registers as accumulators
// (weight gradient), and the partial gradients of the workgroups are added in index order.
constexpr int NL_MAX = 8, NL_BLOCKS = 64, NL_VALS = NL_MAX * NL_MAX + NL_MAX;

__global__ __launch_bounds__(256) void narrow_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, long B, int N, int M, int act,
                                                         float* __restrict__ y) {
    __shared__ float ws[NL_VALS];                  // w [N][M] then the bias at NL_MAX*NL_MAX
    if (threadIdx.x < N * M) ws[threadIdx.x] = w[threadIdx.x];
    if (threadIdx.x < N) ws[NL_MAX * NL_MAX + threadIdx.x] = bias ? bias[threadIdx.x] : 0.0f;
    __syncthreads();
    for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < B; b += (long)gridDim.x * 256) {
        float xv[NL_MAX];
#pragma unroll
        for (int m = 0; m < NL_MAX; ++m) xv[m] = m < M ? x[b * M + m] : 0.0f;
#pragma unroll
        for (int n = 0; n < NL_MAX; ++n) {
            if (n < N) {
                float sum = ws[NL_MAX * NL_MAX + n];
#pragma unroll
                for (int m = 0; m < NL_MAX; ++m)
                    if (m < M) sum = __builtin_fmaf(xv[m], ws[n * M + m], sum);
                y[b * N + n] = dact(act, sum);
            }
        }
    }
}

// part[block][n * NL_MAX + m] = sum over the block's samples of g[b][n] x[b][m]; [.. + 64 + n] = sum of g[b][n]
__global__ __launch_bounds__(256) void narrow_wgrad_partial_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                   long B, int N, int M, float* __restrict__ part) {
    __shared__ float sm[4][NL_VALS];
    float acc[NL_MAX][NL_MAX], accb[NL_MAX];
#pragma unroll
    for (int n = 0; n < NL_MAX; ++n) {
        accb[n] = 0.0f;
#pragma unroll
        for (int m = 0; m < NL_MAX; ++m) acc[n][m] = 0.0f;
    }
    for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < B; b += (long)gridDim.x * 256) {
        float gv[NL_MAX], xv[NL_MAX];
#pragma unroll
        for (int n = 0; n < NL_MAX; ++n) gv[n] = n < N ? g[b * N + n] : 0.0f;
#pragma unroll
        for (int m = 0; m < NL_MAX; ++m) xv[m] = m < M ? x[b * M + m] : 0.0f;
#pragma unroll
        for (int n = 0; n < NL_MAX; ++n) {
            accb[n] += gv[n];
#pragma unroll
            for (int m = 0; m < NL_MAX; ++m) acc[n][m] = __builtin_fmaf(gv[n], xv[m], acc[n][m]);
        }
    }
    // 72 wave sums: on the DPP path (six VALU operations each; the shuffle form goes through the LDS crossbar
    // and made this kernel 17 us for a 5 x 4 weight), the total lands in lane 63
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < NL_MAX; ++n) {
        const float sb = wave_sum_dpp(accb[n]);
        if (lane == 63) sm[wid][NL_MAX * NL_MAX + n] = sb;
#pragma unroll
        for (int m = 0; m < NL_MAX; ++m) {
            const float sw = wave_sum_dpp(acc[n][m]);
            if (lane == 63) sm[wid][n * NL_MAX + m] = sw;
        }
    }
    __syncthreads();
    if (threadIdx.x < NL_VALS)
        part[(long)blockIdx.x * NL_VALS + threadIdx.x] =
            (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

__global__ __launch_bounds__(128) void narrow_wgrad_final_kernel(const float* __restrict__ part, int nparts, int N, int M,
                                                                 int accumulate, float* __restrict__ gw,
                                                                 float* __restrict__ gb) {
    const int t = threadIdx.x;
    if (t >= NL_VALS) return;
    // the partial sums are added in block order; sixteen loads in flight (one at a time, the 64 round trips
    // of this loop were 16 us)
    float sum = 0.0f;
    int p = 0;
    for (; p + 16 <= nparts; p += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = part[(long)(p + u) * NL_VALS + t];
#pragma unroll
        for (int u = 0; u < 16; ++u) sum += v[u];
    }
    for (; p < nparts; ++p) sum += part[(long)p * NL_VALS + t];
    if (t < NL_MAX * NL_MAX) {
        const int n = t / NL_MAX, m = t - n * NL_MAX;
        if (n < N && m < M) gw[n * M + m] = accumulate ? gw[n * M + m] + sum : sum;
    } else if (gb) {
        const int n = t - NL_MAX * NL_MAX;
        if (n < N) gb[n] = accumulate ? gb[n] + sum : sum;
    }
}

}  // namespace mdnvae
using namespace mdnvae;

static int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

extern "C" {

int mdnvae_bias_act(float* y, const float* bias, long rows, int cols, int act, void* stream) {
    MV_REQUIRE(rows >= 0 && cols >= 1, E_ARG, "mdnvae_bias_act: bad shape");
    MV_REQUIRE(act >= 0 && act <= MDNVAE_ACT_RELU, E_ARG, "mdnvae_bias_act: unknown activation %d", act);
    if (rows == 0) return 0;
    MV_REQUIRE(y, E_ARG, "mdnvae_bias_act: null pointer");
    hipLaunchKernelGGL(bias_act_kernel, dim3(ew_grid(rows * cols)), dim3(256), 0, (hipStream_t)stream, y, bias, rows,
                       cols, act);
    MV_LAUNCH_CHECK("bias_act_kernel");
    return 0;
}

int mdnvae_act_bwd(const float* y, const float* grad_y, float* grad_pre, long n, int act, void* stream) {
    MV_REQUIRE(n >= 0, E_ARG, "mdnvae_act_bwd: negative size");
    MV_REQUIRE(act >= 0 && act <= MDNVAE_ACT_RELU, E_ARG, "mdnvae_act_bwd: unknown activation %d", act);
    if (n == 0) return 0;
    MV_REQUIRE(y && grad_y && grad_pre, E_ARG, "mdnvae_act_bwd: null pointer");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, y, grad_y, grad_pre, n, act);
    MV_LAUNCH_CHECK("act_bwd_kernel");
    return 0;
}

int mdnvae_add_act(const float* a, const float* b, float* y, long n, int act, void* stream) {
    MV_REQUIRE(n >= 0, E_ARG, "mdnvae_add_act: negative size");
    MV_REQUIRE(act >= 0 && act <= MDNVAE_ACT_RELU, E_ARG, "mdnvae_add_act: unknown activation %d", act);
    if (n == 0) return 0;
    MV_REQUIRE(a && b && y, E_ARG, "mdnvae_add_act: null pointer");
    hipLaunchKernelGGL(add_act_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n, act);
    MV_LAUNCH_CHECK("add_act_kernel");
    return 0;
}

size_t mdnvae_colsum_workspace_floats(int B, int D) { return (size_t)((B + 63) / 64) * D; }

int mdnvae_colsum(const float* a, int B, int D, float* workspace, float* out, int accumulate, void* stream) {
    MV_REQUIRE(B >= 1 && D >= 1, E_ARG, "mdnvae_colsum: empty input");
    MV_REQUIRE(a && workspace && out, E_ARG, "mdnvae_colsum: null pointer");
    if (B <= 512) {
        // a short batch is one slab: the folding kernel adds its rows straight into the result
        hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((D + CS_COLS - 1) / CS_COLS)), dim3(256), 0,
                           (hipStream_t)stream, a, B, D, accumulate, out);
        MV_LAUNCH_CHECK("colsum_final_kernel");
        return 0;
    }
    // slabs of 64 rows for small batches, longer ones once there are enough slabs to fill the chip
    const int rows = B >= 8192 ? 256 : (B >= 2048 ? 128 : 64), nparts = (B + rows - 1) / rows;
    const unsigned cols = (unsigned)((D + CS_COLS - 1) / CS_COLS);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cols, nparts), dim3(256), 0, (hipStream_t)stream, a, B, D, rows,
                       workspace);
    MV_LAUNCH_CHECK("colsum_partial_kernel");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cols), dim3(256), 0, (hipStream_t)stream, workspace, nparts, D, accumulate,
                       out);
    MV_LAUNCH_CHECK("colsum_final_kernel");
    return 0;
}

static int narrow_blocks(long B) {
    const long g = (B + 255) / 256;
    return (int)(g > NL_BLOCKS ? NL_BLOCKS : (g < 1 ? 1 : g));
}

int mdnvae_narrow_linear_fwd(const float* x, const float* w, const float* bias, long B, int N, int M, int act, float* y,
                             void* stream) {
    MV_REQUIRE(B >= 0, E_ARG, "mdnvae_narrow_linear_fwd: negative batch");
    MV_REQUIRE(N >= 1 && N <= NL_MAX && M >= 1 && M <= NL_MAX, E_SHAPE,
               "mdnvae_narrow_linear_fwd: a [%d x %d] weight is not narrow (both sides 1..%d)", N, M, NL_MAX);
    MV_REQUIRE(act >= 0 && act <= MDNVAE_ACT_RELU, E_ARG, "mdnvae_narrow_linear_fwd: unknown activation %d", act);
    if (B == 0) return 0;
    MV_REQUIRE(x && w && y, E_ARG, "mdnvae_narrow_linear_fwd: null pointer");
    hipLaunchKernelGGL(narrow_fwd_kernel, dim3(ew_grid(B)), dim3(256), 0, (hipStream_t)stream, x, w, bias, B, N, M, act, y);
    MV_LAUNCH_CHECK("narrow_fwd_kernel");
    return 0;
}

size_t mdnvae_narrow_linear_workspace_floats(void) { return (size_t)NL_BLOCKS * NL_VALS; }

int mdnvae_narrow_linear_wgrad(const float* grad_pre, const float* x, long B, int N, int M, float* workspace, float* grad_w,
                               float* grad_b, int accumulate, void* stream) {
    MV_REQUIRE(B >= 1, E_ARG, "mdnvae_narrow_linear_wgrad: empty batch");
    MV_REQUIRE(N >= 1 && N <= NL_MAX && M >= 1 && M <= NL_MAX, E_SHAPE,
               "mdnvae_narrow_linear_wgrad: a [%d x %d] weight is not narrow (both sides 1..%d)", N, M, NL_MAX);
    MV_REQUIRE(grad_pre && x && workspace && grad_w, E_ARG, "mdnvae_narrow_linear_wgrad: null pointer");
    const int blocks = narrow_blocks(B);
    hipLaunchKernelGGL(narrow_wgrad_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grad_pre, x, B, N, M,
                       workspace);
    MV_LAUNCH_CHECK("narrow_wgrad_partial_kernel");
    hipLaunchKernelGGL(narrow_wgrad_final_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, workspace, blocks, N, M,
                       accumulate, grad_w, grad_b);
    MV_LAUNCH_CHECK("narrow_wgrad_final_kernel");
    return 0;
}

}  // extern "C"
