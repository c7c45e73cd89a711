// Weight gradient of a small dense layer over a long batch:  dW [N,M] = g^T . x  with g [B,N] the
// gradient of the layer's pre-activations and x [B,M] its input (the autograd of F.linear in the
// encoder towers and latent heads, function/MM_Gate.py:157-165, function/VAE_CNN_NN.py:160-161).
// The output is small (N*M up to a few hundred thousand) and the reduction dimension is the batch,
// so the work is split over the batch: a workgroup owns a 64 (n) x 128 (m) tile of dW for one slab
// of samples, the slabs' partial tiles are added in slab order by a second kernel (fixed order,
// reproducible).  Rows of x and g are staged through LDS in chunks of 32 samples, fetched into
// registers one chunk ahead; the products run on the fp32 matrix cores with g as the A operand, so
// the accumulator columns run along m and the partial tile is stored in whole cache lines.
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TS_TN = 64, TS_TM = 128, TS_KC = 32, TS_NT = 256;
constexpr int TS_GLD = TS_TN + 4, TS_XLD = TS_TM + 4;        // LDS row strides (floats), 16-byte multiples
constexpr int TS_GQ = TS_KC * TS_TN / 4 / TS_NT;            // float4 per thread per chunk of g (2)
constexpr int TS_XQ = TS_KC * TS_TM / 4 / TS_NT;            // ... of x (4)

__device__ __forceinline__ void ts_fetch(const float* __restrict__ g, const float* __restrict__ x, int N, int M, int b0,
                                         int b_end, int n0, int m0, float4 (&cg)[TS_GQ], float4 (&cx)[TS_XQ]) {
#pragma unroll
    for (int i = 0; i < TS_GQ; ++i) {
        const int q = i * TS_NT + threadIdx.x, r = q / (TS_TN / 4), col = (q - r * (TS_TN / 4)) * 4;
        const int b = b0 + r, n = n0 + col;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < b_end && n < N) v = *reinterpret_cast<const float4*>(g + (long)b * N + n);
        cg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < TS_XQ; ++i) {
        const int q = i * TS_NT + threadIdx.x, r = q / (TS_TM / 4), col = (q - r * (TS_TM / 4)) * 4;
        const int b = b0 + r, m = m0 + col;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < b_end && m < M) v = *reinterpret_cast<const float4*>(x + (long)b * M + m);
        cx[i] = v;
    }
}
__device__ __forceinline__ void ts_commit(const float4 (&cg)[TS_GQ], const float4 (&cx)[TS_XQ], float* __restrict__ gs,
                                          float* __restrict__ xs) {
#pragma unroll
    for (int i = 0; i < TS_GQ; ++i) {
        const int q = i * TS_NT + threadIdx.x, r = q / (TS_TN / 4), col = (q - r * (TS_TN / 4)) * 4;
        *reinterpret_cast<float4*>(gs + r * TS_GLD + col) = cg[i];
    }
#pragma unroll
    for (int i = 0; i < TS_XQ; ++i) {
        const int q = i * TS_NT + threadIdx.x, r = q / (TS_TM / 4), col = (q - r * (TS_TM / 4)) * 4;
        *reinterpret_cast<float4*>(xs + r * TS_XLD + col) = cx[i];
    }
}

// grid: (m tiles, n tiles, slabs); part[slab][n][m].  part_b (or NULL): [slab][n] = the slab's column sums of g,
// the bias gradient of the same layer - every lane adds up the g values it feeds to the matrix cores anyway, the
// workgroups of the first m tile store them.
__global__ __launch_bounds__(TS_NT) void tsgemm_partial_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                               int B, int N, int M, int slab_rows,
                                                               float* __restrict__ part, float* __restrict__ part_b) {
    __shared__ __attribute__((aligned(16))) float gs[2][TS_KC * TS_GLD];
    __shared__ __attribute__((aligned(16))) float xs[2][TS_KC * TS_XLD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = (wave & 1) * 32, wm = (wave >> 1) * 64;       // this wave: 32 rows (n) x 64 columns (m)
    const int m0 = blockIdx.x * TS_TM, n0 = blockIdx.y * TS_TN;
    const int b_begin = blockIdx.z * slab_rows, b_end = min(B, b_begin + slab_rows);
    f32x16 acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc0[j] = acc1[j] = 0.0f;
    float bsum = 0.0f;
    float4 cg[TS_GQ], cx[TS_XQ];
    ts_fetch(g, x, N, M, b_begin, b_end, n0, m0, cg, cx);
    ts_commit(cg, cx, gs[0], xs[0]);
    __syncthreads();
    const int h = lane >> 5, l32 = lane & 31;
    int cur = 0;
    for (int b0 = b_begin; b0 < b_end; b0 += TS_KC, cur ^= 1) {
        const bool more = b0 + TS_KC < b_end;
        if (more) ts_fetch(g, x, N, M, b0 + TS_KC, b_end, n0, m0, cg, cx);
        const float* ga = gs[cur] + h * TS_GLD + wn + l32;
        const float* xb = xs[cur] + h * TS_XLD + wm + l32;
#pragma unroll
        for (int k = 0; k < TS_KC; k += 2) {
            const float av = ga[k * TS_GLD];
            bsum += av;
            const float b0v = xb[k * TS_XLD], b1v = xb[k * TS_XLD + 32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1v, acc1, 0, 0, 0);
        }
        if (more) ts_commit(cg, cx, gs[cur ^ 1], xs[cur ^ 1]);
        __syncthreads();
    }
    if (part_b && blockIdx.x == 0 && wm == 0) {
        // lanes l32 and l32 + 32 hold the even and the odd samples of column n0 + wn + l32
        bsum += __shfl_xor(bsum, 32);
        if (h == 0 && n0 + wn + l32 < N) part_b[(long)blockIdx.z * N + n0 + wn + l32] = bsum;
    }
    float* dst = part + (long)blockIdx.z * N * M;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int n = n0 + wn + (j & 3) + 8 * (j >> 2) + 4 * h;
        const int m = m0 + wm + l32;
        if (n < N) {
            if (m < M) dst[(long)n * M + m] = acc0[j];
            if (m + 32 < M) dst[(long)n * M + m + 32] = acc1[j];
        }
    }
}

// out[e] = sum_s part[s][e]: a workgroup covers 64 quads of e with four groups of lanes that each
// add every fourth slab (four loads in flight), the four group sums meet in LDS in a fixed order
// A second, smaller sum (part2, count2, accumulate2, out2: the bias gradient's partial column sums) rides along
// in the blocks past the first one's: one launch for both.
__global__ __launch_bounds__(256) void tsgemm_reduce_kernel(const float* __restrict__ part, int slabs, long count,
                                                            int accumulate, float* __restrict__ out, int blocks1,
                                                            const float* __restrict__ part2, long count2, int accumulate2,
                                                            float* __restrict__ out2) {
    __shared__ float4 sm[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    int blk = blockIdx.x;
    if (blk >= blocks1) {
        blk -= blocks1;
        part = part2;
        count = count2;
        accumulate = accumulate2;
        out = out2;
    }
    const long e4 = (long)blk * 64 + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 * 4 < count) {
        int p = grp;
        for (; p + 12 < slabs; p += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(part + (long)(p + 4 * u) * count + e4 * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; p < slabs; p += 4) {
            const float4 v = *reinterpret_cast<const float4*>(part + (long)p * count + e4 * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sm[grp][col] = s;
    __syncthreads();
    if (grp == 0 && e4 * 4 < count) {
        float4 r = sm[0][col];
#pragma unroll
        for (int u = 1; u < 4; ++u) { r.x += sm[u][col].x; r.y += sm[u][col].y; r.z += sm[u][col].z; r.w += sm[u][col].w; }
        if (accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(out + e4 * 4);
            r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
        }
        *reinterpret_cast<float4*>(out + e4 * 4) = r;
    }
}

// slabs: two workgroups per CU, at least 128 samples per slab (measured at B = 16384: 56 us for the
// [128 x 1024] gradient against 65 us with 1024 workgroups - half the partial tiles to write and add -
// and 65 us with 256; 17 us against 20 us for [68 x 128] with the longer slabs)
static int ts_slabs(int B, int N, int M) {
    const long tiles = (long)((M + TS_TM - 1) / TS_TM) * ((N + TS_TN - 1) / TS_TN);
    long s = (512 + tiles - 1) / tiles;
    const long most = (B + 127) / 128;
    if (s > most) s = most;
    if (s > 256) s = 256;
    return (int)(s < 1 ? 1 : s);
}

// ---- weight gradients of the chain decoder's panels ----
// dwt[rl] [96 x 96] = layer_in[rl]^T . layer_dpre[rl] for all R*NL (channel, layer) pairs of a
// decoder at once: layer_in / layer_dpre are [RL][B][96] (input rows with the constant-1 column,
// gradients of the pre-activations).  Same split over the batch as above; a workgroup owns one whole
// 96 x 96 panel for one slab of samples, its four waves a 48 x 48 quarter each as 3 x 3 tiles of the
// 16x16x4 fp32 matrix instruction (a 96-wide panel does not fill the 64 x 128 tile of the kernel
// above: 56 % of its matrix work would be padding).  Rows of both operands are 96 contiguous floats,
// so a chunk of 32 samples is one contiguous 12 KB run of each, fetched 16 bytes per lane one chunk
// ahead.  LDS row stride 112 floats: the two sample rows that the 32 lanes of one LDS pass read start
// 16 banks apart.
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int PW = 96, PW_LD = 112, PW_KC = 32, PW_NT = 256, PW_Q = PW_KC * PW / 4 / PW_NT;   // 3 float4 per thread

__device__ __forceinline__ void pw_fetch(const float* __restrict__ H, const float* __restrict__ D, int b0, int b_end,
                                         float4 (&ch)[PW_Q], float4 (&cd)[PW_Q]) {
#pragma unroll
    for (int i = 0; i < PW_Q; ++i) {
        const int q = i * PW_NT + threadIdx.x, r = q / (PW / 4), c4 = (q - r * (PW / 4)) * 4;
        const int b = b0 + r;
        float4 vh = make_float4(0.f, 0.f, 0.f, 0.f), vd = vh;
        if (b < b_end) {
            vh = *reinterpret_cast<const float4*>(H + (long)b * PW + c4);
            vd = *reinterpret_cast<const float4*>(D + (long)b * PW + c4);
        }
        ch[i] = vh;
        cd[i] = vd;
    }
}
__device__ __forceinline__ void pw_commit(const float4 (&ch)[PW_Q], const float4 (&cd)[PW_Q], float* __restrict__ hs,
                                          float* __restrict__ ds) {
#pragma unroll
    for (int i = 0; i < PW_Q; ++i) {
        const int q = i * PW_NT + threadIdx.x, r = q / (PW / 4), c4 = (q - r * (PW / 4)) * 4;
        *reinterpret_cast<float4*>(hs + r * PW_LD + c4) = ch[i];
        *reinterpret_cast<float4*>(ds + r * PW_LD + c4) = cd[i];
    }
}

// grid: (slabs, RL); part[slab][rl][96][96]
__global__ __launch_bounds__(PW_NT, 2) void panel_wgrad_partial_kernel(const float* __restrict__ layer_in,
                                                                       const float* __restrict__ layer_dpre, int B,
                                                                       int slab_rows, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float hs[2][PW_KC * PW_LD];
    __shared__ __attribute__((aligned(16))) float ds[2][PW_KC * PW_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 48, wc = (wave & 1) * 48;
    const int rl = blockIdx.y;
    const float* H = layer_in + (long)rl * B * PW;
    const float* D = layer_dpre + (long)rl * B * PW;
    const int b_begin = blockIdx.x * slab_rows, b_end = min(B, b_begin + slab_rows);
    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ch[PW_Q], cd[PW_Q];
    pw_fetch(H, D, b_begin, b_end, ch, cd);
    pw_commit(ch, cd, hs[0], ds[0]);
    __syncthreads();
    const int kk = lane >> 4, l16 = lane & 15;
    int cur = 0;
    for (int b0 = b_begin; b0 < b_end; b0 += PW_KC, cur ^= 1) {
        const bool more = b0 + PW_KC < b_end;
        if (more) pw_fetch(H, D, b0 + PW_KC, b_end, ch, cd);
        const float* ha = hs[cur] + kk * PW_LD + wr + l16;
        const float* db = ds[cur] + kk * PW_LD + wc + l16;
#pragma unroll
        for (int k = 0; k < PW_KC; k += 4) {
            float a[3], b[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                a[i] = ha[k * PW_LD + 16 * i];
                b[i] = db[k * PW_LD + 16 * i];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) pw_commit(ch, cd, hs[cur ^ 1], ds[cur ^ 1]);
        __syncthreads();
    }
    float* dst = part + ((long)blockIdx.x * gridDim.y + rl) * PW * PW;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(wr + 16 * i + 4 * kk + e) * PW + wc + 16 * j + l16] = acc[i][j][e];
}

// slabs: some 640 workgroups in all, at least 64 samples per slab
static int pw_slabs(int B, int count) {
    long s = (640 + count - 1) / count;
    const long most = (B + 63) / 64;
    if (s > most) s = most;
    if (s > 64) s = 64;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

// the slabs' partial tiles, then (for the entry point that also gives the bias gradient) their column sums
size_t mdnvae_linear_wgrad_workspace_floats(int B, int N, int M) {
    if (B < 1 || N < 1 || M < 1) return 0;
    return (size_t)ts_slabs(B, N, M) * ((size_t)N * M + N);
}

static int linear_wgrad_impl(const char* fn, const float* grad_pre, const float* x, int B, int N, int M, float* workspace,
                             float* grad_w, int accumulate, float* grad_b, int accumulate_b, void* stream) {
    MV_REQUIRE(B >= 1 && N >= 1 && M >= 1, E_ARG, "%s: empty operand", fn);
    MV_REQUIRE((N & 3) == 0 && (M & 3) == 0, E_SHAPE, "%s: N = %d and M = %d must be multiples of 4 (16-byte rows)", fn, N, M);
    MV_REQUIRE(grad_pre && x && workspace && grad_w, E_ARG, "%s: null pointer", fn);
    MV_REQUIRE(aligned16(grad_pre) && aligned16(x) && aligned16(workspace) && aligned16(grad_w) && aligned16(grad_b), E_ALIGN,
               "%s: operands must be 16-byte aligned", fn);
    const int slabs = ts_slabs(B, N, M);
    const int slab_rows = ((B + slabs - 1) / slabs + TS_KC - 1) / TS_KC * TS_KC;
    const int used = (B + slab_rows - 1) / slab_rows;
    const dim3 grid((M + TS_TM - 1) / TS_TM, (N + TS_TN - 1) / TS_TN, used);
    MV_REQUIRE(grid.y <= 65535 && grid.z <= 65535, E_SHAPE, "%s: shape too large", fn);
    float* part_b = grad_b ? workspace + (size_t)used * N * M : nullptr;
    hipLaunchKernelGGL(tsgemm_partial_kernel, grid, dim3(TS_NT), 0, (hipStream_t)stream, grad_pre, x, B, N, M, slab_rows,
                       workspace, part_b);
    MV_LAUNCH_CHECK("tsgemm_partial_kernel");
    const long count = (long)N * M;
    const int blocks1 = (int)((count / 4 + 63) / 64), blocks2 = grad_b ? (N / 4 + 63) / 64 : 0;
    hipLaunchKernelGGL(tsgemm_reduce_kernel, dim3((unsigned)(blocks1 + blocks2)), dim3(256), 0, (hipStream_t)stream,
                       workspace, used, count, accumulate, grad_w, blocks1, part_b, (long)N, accumulate_b, grad_b);
    MV_LAUNCH_CHECK("tsgemm_reduce_kernel");
    return 0;
}

int mdnvae_linear_wgrad(const float* grad_pre, const float* x, int B, int N, int M, float* workspace, float* grad_w,
                        int accumulate, void* stream) {
    return linear_wgrad_impl("mdnvae_linear_wgrad", grad_pre, x, B, N, M, workspace, grad_w, accumulate, nullptr, 0, stream);
}

int mdnvae_linear_wgrad_bias(const float* grad_pre, const float* x, int B, int N, int M, float* workspace, float* grad_w,
                             int accumulate, float* grad_b, int accumulate_b, void* stream) {
    MV_REQUIRE(grad_b, E_ARG, "mdnvae_linear_wgrad_bias: grad_b is null");
    return linear_wgrad_impl("mdnvae_linear_wgrad_bias", grad_pre, x, B, N, M, workspace, grad_w, accumulate, grad_b,
                             accumulate_b, stream);
}

size_t mdnvae_decoder_wgrad_workspace_floats(int B, int count) {
    if (B < 1 || count < 1) return 0;
    return (size_t)pw_slabs(B, count) * count * PW * PW;
}

int mdnvae_decoder_wgrad(const float* layer_in, const float* layer_dpre, int B, int count, float* workspace, float* dwt,
                         void* stream) {
    MV_REQUIRE(B >= 1 && count >= 1 && count <= 65535, E_ARG, "mdnvae_decoder_wgrad: bad size");
    MV_REQUIRE(PW == MDNVAE_DEC_KP && PW == MDNVAE_DEC_NP, E_SHAPE, "mdnvae_decoder_wgrad: panels must be 96 x 96");
    MV_REQUIRE(layer_in && layer_dpre && workspace && dwt, E_ARG, "mdnvae_decoder_wgrad: null pointer");
    MV_REQUIRE(aligned16(layer_in) && aligned16(layer_dpre) && aligned16(workspace) && aligned16(dwt), E_ALIGN,
               "mdnvae_decoder_wgrad: operands must be 16-byte aligned");
    const int slabs = pw_slabs(B, count);
    const int slab_rows = ((B + slabs - 1) / slabs + PW_KC - 1) / PW_KC * PW_KC;
    const int used = (B + slab_rows - 1) / slab_rows;
    hipLaunchKernelGGL(panel_wgrad_partial_kernel, dim3(used, count), dim3(PW_NT), 0, (hipStream_t)stream, layer_in,
                       layer_dpre, B, slab_rows, workspace);
    MV_LAUNCH_CHECK("panel_wgrad_partial_kernel");
    const long n = (long)count * PW * PW;
    const int blocks1 = (int)((n / 4 + 63) / 64);
    hipLaunchKernelGGL(tsgemm_reduce_kernel, dim3((unsigned)blocks1), dim3(256), 0, (hipStream_t)stream, workspace, used, n, 0,
                       dwt, blocks1, (const float*)nullptr, 0L, 0, (float*)nullptr);
    MV_LAUNCH_CHECK("tsgemm_reduce_kernel");
    return 0;
}

}  // extern "C"
