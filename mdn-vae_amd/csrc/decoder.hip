// Multi-channel "chain" decoder: R channels, each a short chain of small dense layers on the
// latent vector, whose final node vectors g_r are combined as sum_r g_r g_r^T.
//
// One kernel family serves both decoder styles of the reference:
//   rank-1 "NN" / masked-GCN decoder   function/VAE_CNN_NN.py:169-189, function/VAE-CNN-GCN.py:166-185,
//       chain = Linear(L->N)+act, Linear(N->N, weight*mask)+act; epilogue act(sum) [*w_diag + b]
//   multi-channel k-NN GCN decoder     function/MM_Gate.py:84-93,126-138
//       chain = Linear(L->N), 3 x (scattered k-NN weights + sigmoid); epilogue 0.5*(S+S^T) (== S)
//
// All layer weights live in one packed, zero-padded panel so that every layer is a plain
// [samples x K] . [K x NP] product: Wt[r][l][k][n] (k < K; row KP-1 holds the bias) and the
// same weights transposed Wn[r][l][n][k] for the backward's dh = dpre . W.  The k-NN
// adjacency and the 0/1 masks are applied when the panel is packed (scatter / multiply),
// once per step, which is what the reference redoes on every forward.
//
// The backward saves only the inputs and the final node vectors: one streaming kernel turns the
// output gradient into the gradient of the node vectors, a second one recomputes each chain per
// tile, walks it backwards, and emits per-layer (input activation, pre-activation gradient)
// rows; the weight/bias gradients are then [K x B].[B x N] library GEMMs over those rows.
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

constexpr int NP = MDNVAE_DEC_NP;   // padded node count
constexpr int KP = MDNVAE_DEC_KP;   // padded fan-in; row KP-1 = bias
constexpr int MAXL = 4;             // layers per channel
constexpr int MAXR = 8;
static_assert(NP == KP, "layer outputs are reused as the next layer's input rows");

constexpr int ACT_NONE = 0, ACT_SIGMOID = 1, ACT_TANH = 2;

// 1 / (1 + e^-v) on the hardware exp2 and reciprocal (each within 1 ulp): the chain kernels run one
// wave per SIMD, so the ~40 instructions of the library expf and division are not hidden
__device__ __forceinline__ float sigmoid_fast(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
// tanh without a branch: (1 - e) / (1 + e), e = exp(-2|v|), where the difference keeps its digits, and
// the odd series through v^9 below |v| = 1/4 (its first dropped term is 8e-9 of the value there)
__device__ __forceinline__ float tanh_fast(float v) {
    const float t = fabsf(v), e = __builtin_amdgcn_exp2f(-2.8853900817779268f * t);
    const float big = copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), v);
    const float q = v * v;
    const float small = v * (1.0f + q * (-0.33333333333f + q * (0.13333333333f + q * (-0.05396825397f + q * 0.02186948854f))));
    return t < 0.25f ? small : big;
}
template <int ACT>
__device__ __forceinline__ float act_of(float v) {
    if (ACT == ACT_SIGMOID) return sigmoid_fast(v);
    if (ACT == ACT_TANH) return tanh_fast(v);
    return v;
}
__device__ __forceinline__ float act_apply(int a, float v) {
    if (a == ACT_SIGMOID) return sigmoid_fast(v);
    if (a == ACT_TANH) return tanh_fast(v);
    return v;
}
// derivative expressed through the activation's output h
__device__ __forceinline__ float act_slope(int a, float h) {
    if (a == ACT_SIGMOID) return h * (1.0f - h);
    if (a == ACT_TANH) return 1.0f - h * h;
    return 1.0f;
}
__device__ __forceinline__ int layer_act(int acts, int l) { return (acts >> (2 * l)) & 3; }

// ---------------------------------------------------------------------------------------
// panel packing
// ---------------------------------------------------------------------------------------
// dense (optionally masked) Linear weight [N_out, K] (+bias [N_out]) -> Wt[k][n], Wn[n][k]
__global__ void pack_linear_kernel(const float* __restrict__ w, const float* __restrict__ mask,
                                   const float* __restrict__ bias, int n_out, int K, float* __restrict__ wt,
                                   float* __restrict__ wn) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KP * NP) return;
    const int k = idx / NP, n = idx - k * NP;
    float v = 0.0f;
    if (n < n_out) {
        if (k < K) {
            v = w[n * K + k];
            if (mask) v *= mask[n * K + k];
        } else if (k == KP - 1 && bias) {
            v = bias[n];
        }
    }
    wt[k * NP + n] = v;
    wn[n * KP + k] = (k == KP - 1) ? 0.0f : v;
}

// k-NN GCN layer weight [N, kn] with adjacency [N, kn] (int32): S[u][adj[u][j]] = W[u][j]
// function/MM_Gate.py:84-93 (x_out[:,u] = sum_j x[:,adj[u][j]] * W[u][j] + b[u])
__global__ void pack_gcn_kernel(const float* __restrict__ w, const int* __restrict__ adj,
                                const float* __restrict__ bias, int N, int kn, float* __restrict__ wt,
                                float* __restrict__ wn) {
    // launch with one thread per (k, n) after the panel has been zeroed by the memset the
    // host entry issues; here each thread owns output node u = n and scans its neighbour list
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KP * NP) return;
    const int k = idx / NP, u = idx - k * NP;
    float v = 0.0f;
    if (u < N) {
        if (k == KP - 1) {
            v = bias ? bias[u] : 0.0f;
        } else if (k < N) {
            for (int j = 0; j < kn; ++j)
                if (adj[u * kn + j] == k) v += w[u * kn + j];
        }
    }
    wt[k * NP + u] = v;
    wn[u * KP + k] = (k == KP - 1) ? 0.0f : v;
}

// gradient panel dWt[k][n] (row KP-1 = bias gradient) -> dW [N_out,K] (*mask), db [N_out]
__global__ void unpack_linear_grad_kernel(const float* __restrict__ dwt, const float* __restrict__ mask, int n_out,
                                          int K, float* __restrict__ dw, float* __restrict__ db) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n_out * K) {
        const int n = idx / K, k = idx - n * K;
        float v = dwt[k * NP + n];
        if (mask) v *= mask[idx];
        dw[idx] = v;
    }
    if (db && idx < n_out) db[idx] = dwt[(KP - 1) * NP + idx];
}

__global__ void unpack_gcn_grad_kernel(const float* __restrict__ dwt, const int* __restrict__ adj, int N, int kn,
                                       float* __restrict__ dw, float* __restrict__ db) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < N * kn) {
        const int u = idx / kn;
        dw[idx] = dwt[adj[idx] * NP + u];
    }
    if (db && idx < N) db[idx] = dwt[(KP - 1) * NP + idx];
}

// Every layer of a decoder in one launch: the per-layer pointers travel by value in the kernel
// arguments, blockIdx.y picks the layer.  kind 0 = Linear (aux = mask or NULL), 1 = GCN layer
// (aux = adjacency).
constexpr int MAX_LAYERS = 32;
struct PackTable {
    const float* w[MAX_LAYERS];
    const float* bias[MAX_LAYERS];
    const void* aux[MAX_LAYERS];
    int n_out[MAX_LAYERS], k[MAX_LAYERS], kind[MAX_LAYERS];
};
__global__ __launch_bounds__(256) void pack_layers_kernel(PackTable t, float* __restrict__ wt, float* __restrict__ wn) {
    const int ly = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= KP * NP) return;
    const float* __restrict__ w = t.w[ly];
    const float* __restrict__ bias = t.bias[ly];
    const int n_out = t.n_out[ly], K = t.k[ly];
    const int k = idx / NP, n = idx - k * NP;
    float v = 0.0f;
    if (n < n_out) {
        if (k == KP - 1) {
            v = bias ? bias[n] : 0.0f;
        } else if (t.kind[ly] == 0) {
            if (k < K) {
                v = w[n * K + k];
                const float* mask = static_cast<const float*>(t.aux[ly]);
                if (mask) v *= mask[n * K + k];
            }
        } else if (k < n_out) {
            const int* adj = static_cast<const int*>(t.aux[ly]);
            for (int j = 0; j < K; ++j)
                if (adj[n * K + j] == k) v += w[n * K + j];
        }
    }
    const long base = (long)ly * KP * NP;
    wt[base + k * NP + n] = v;
    wn[base + n * KP + k] = (k == KP - 1) ? 0.0f : v;
}

struct UnpackTable {
    float* dw[MAX_LAYERS];
    float* db[MAX_LAYERS];
    const void* aux[MAX_LAYERS];
    int n_out[MAX_LAYERS], k[MAX_LAYERS], kind[MAX_LAYERS];
};
__global__ __launch_bounds__(256) void unpack_layers_kernel(UnpackTable t, const float* __restrict__ dwt_all,
                                                            int accumulate) {
    const int ly = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const float* __restrict__ dwt = dwt_all + (long)ly * KP * NP;
    const int n_out = t.n_out[ly], K = t.k[ly];
    float* __restrict__ dw = t.dw[ly];
    float* __restrict__ db = t.db[ly];
    if (idx < n_out * K) {
        const int n = idx / K, k = idx - n * K;
        float v;
        if (t.kind[ly] == 0) {
            v = dwt[k * NP + n];
            const float* mask = static_cast<const float*>(t.aux[ly]);
            if (mask) v *= mask[idx];
        } else {
            v = dwt[static_cast<const int*>(t.aux[ly])[idx] * NP + n];
        }
        dw[idx] = accumulate ? dw[idx] + v : v;
    }
    if (db && idx < n_out) {
        const float v = dwt[(KP - 1) * NP + idx];
        db[idx] = accumulate ? db[idx] + v : v;
    }
}

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
// Forward  = chain_fwd_mfma_kernel (z -> final node vectors, x_latent)
//          + outer_epilogue_kernel (node vectors -> out, HBM-write bound).
// Backward = outer_grad_kernel (grad_out, node vectors -> gradient of the node vectors, HBM-read
//            bound) + chain_bwd_mfma_kernel (recompute each chain, walk it backwards).
// The chain kernels run every layer as a [32 samples x 96] . [96 x 96] product on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32); three waves own the three 32-column tiles of the output.
struct DecArgs {
    int B, N, L, R, NL;
    int acts;        // 2 bits per layer
    int epi_act;     // activation on sum_r g g^T
    int has_icpt;    // out = a*wd + bi
    int do_exp;      // out = exp(out)
    int xlat_r;      // x_latent = sum of the first xlat_r channels (0 = none)
    int comp_mode;   // 1: out[b][r][e] = g_r[i]*g_r[j] per channel (mixture components), no epilogue
};

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int CH_TS = 32;            // samples per tile = MFMA rows
constexpr int CH_NT = 192;           // three waves, one per 32-column tile of the layer output
constexpr int CH_LD = CH_TS + 1;     // row stride of the k-major activation buffers (conflict-free)

// sample row of accumulator register j (column = lane & 31)
__device__ __forceinline__ int acc_row(int j, int lane) { return (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5); }

// A wave owns 32 output columns of a layer, and its slice of the layer's panel lives in its
// registers in the matrix-core operand layout: register i of lane (h = lane >> 5, c = lane & 31)
// holds w[2i + h][n0 + c].  The next layer's slice is fetched from global memory (L2 resident: every
// workgroup reads the same panels) around the current product, so LDS carries activations only
// and a workgroup needs little of it - several tiles are resident per CU and hide each other's
// latencies.
constexpr int PANEL_REGS = KP / 2;
__device__ __forceinline__ void panel_fetch(const float* __restrict__ src, int n0, int lane, float (&w)[PANEL_REGS]) {
    const float* p = src + (lane >> 5) * NP + n0 + (lane & 31);
#pragma unroll
    for (int i = 0; i < PANEL_REGS; ++i) w[i] = p[2 * i * NP];
}

// acc[s][n] = sum_k in[k][s] * w[k][n0 + n] over the whole padded panel (row KP-1 of `in` is 1,
// so the bias row of the panel is added by the product itself)
__device__ __forceinline__ f32x16 panel_product(const float* __restrict__ in, const float (&w)[PANEL_REGS], int lane) {
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    const float* ap = in + (lane >> 5) * CH_LD + (lane & 31);
#pragma unroll
    for (int i = 0; i < PANEL_REGS; ++i) {
        // eight activation reads in flight at a time: hoisting all of them costs registers a second
        // resident wave needs
        if ((i & 7) == 0) __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * i * CH_LD], w[i], acc, 0, 0, 0);
    }
    return acc;
}

// z tile -> k-major rows with the ones row at KP-1
__device__ __forceinline__ void load_latent_tile(const float* __restrict__ z, int b0, int B, int L,
                                                 float* __restrict__ zt) {
    for (int i = threadIdx.x; i < KP * CH_TS; i += CH_NT) {
        const int s = i / KP, k = i - s * KP;              // consecutive threads read consecutive k
        float v = 0.0f;
        if (k == KP - 1) v = 1.0f;
        else if (k < L && b0 + s < B) v = z[(long)(b0 + s) * L + k];
        zt[k * CH_LD + s] = v;
    }
}

__global__ __launch_bounds__(CH_NT, 2) void chain_fwd_mfma_kernel(DecArgs a, const float* __restrict__ z,
                                                               const float* __restrict__ wt,
                                                               const float* __restrict__ drop,
                                                               float* __restrict__ nodes, float* __restrict__ xlat,
                                                               float* __restrict__ hin) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* abuf0 = smem;                                // [2][KP][CH_LD]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = wave * 32 + (lane & 31);
    const int N = a.N;
    const int ntiles = (a.B + CH_TS - 1) / CH_TS;
    if ((int)blockIdx.x >= ntiles) return;
    // with few tiles (small batches) the channels are spread over gridDim.y workgroups: the chains
    // of different channels are independent, and a tile's 20 serial layer steps become 4
    const int rper = (a.R + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r_begin = blockIdx.y * rper, r_end = min(a.R, r_begin + rper);
    if (r_begin >= r_end) return;
    const int s_begin = r_begin * a.NL, s_end = r_end * a.NL;      // this workgroup's layer steps per tile
    float wcur[PANEL_REGS];
    panel_fetch(wt + ((long)s_begin * KP) * NP, wave * 32, lane, wcur);
    int cur = 0;
    load_latent_tile(z, blockIdx.x * CH_TS, a.B, a.L, abuf0);
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b0 = tile * CH_TS;
        float xl[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) xl[j] = 0.0f;
        for (int r = r_begin; r < r_end; ++r) {
            for (int l = 0; l < a.NL; ++l) {
                const int s = r * a.NL + l;
                const bool last_step = s == s_end - 1;
                const bool more = !last_step || tile + (int)gridDim.x < ntiles;
                // Every layer's input rows (with the ones column) are kept for the backward pass: they
                // are its weight-gradient operand and, as the previous layer's output, give that
                // layer's activation slope without recomputing the chain.  The first layer's input
                // is the latent tile in LDS; the others are stored from the producing layer's
                // registers below.
                if (hin && l == 0) {
                    const float* src = abuf0 + cur * KP * CH_LD;
                    const long row0 = (long)s * a.B + b0;
                    for (int i = t; i < CH_TS * KP; i += CH_NT) {
                        const int sr = i / KP, k = i - sr * KP;
                        if (b0 + sr < a.B) hin[(row0 + sr) * KP + k] = src[k * CH_LD + sr];
                    }
                }
                const f32x16 acc = panel_product(abuf0 + cur * KP * CH_LD, wcur, lane);
                // One panel slice in registers: the next step's slice is fetched over the one just used,
                // its L2 latency covered by the epilogue below and by the other resident workgroup.  (A
                // second slice fetched ahead of the product costs 48 registers and 48 moves per step
                // and, measured, 25 us of the 174 at B = 16384.)
                if (more) panel_fetch(wt + ((long)(last_step ? s_begin : s + 1) * KP) * NP, wave * 32, lane, wcur);
                const int act = layer_act(a.acts, l);
                float* nxt = abuf0 + (cur ^ 1) * KP * CH_LD;
                if (l == a.NL - 1) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int sr = acc_row(j, lane);
                        if (b0 + sr < a.B) {
                            float v = (n < N) ? act_apply(act, acc[j]) : 0.0f;
                            if (drop && n < N) v *= drop[((long)r * a.B + b0 + sr) * N + n];
                            nodes[((long)(b0 + sr) * a.R + r) * NP + n] = v;
                            if (r < a.xlat_r) xl[j] += v;
                        }
                    }
                    // the buffer this layer would have written is free: it takes the next chain's input
                    if (more) load_latent_tile(z, (last_step ? tile + (int)gridDim.x : tile) * CH_TS, a.B, a.L, nxt);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        float v = (n < N) ? act_apply(act, acc[j]) : 0.0f;
                        if (n == KP - 1) v = 1.0f;
                        const int sr = acc_row(j, lane);
                        nxt[n * CH_LD + sr] = v;
                        if (hin && b0 + sr < a.B) hin[((long)(s + 1) * a.B + b0 + sr) * KP + n] = v;
                    }
                }
                __syncthreads();
                cur ^= 1;
            }
        }
        if (xlat && n < N) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int sr = acc_row(j, lane);
                if (b0 + sr < a.B) xlat[(long)(b0 + sr) * N + n] = xl[j];
            }
        }
    }
}

// Forward for large batches: every wave owns 16 samples through all channels and layers, so activations
// never cross waves - a layer's output goes through a wave-private LDS tile back into the wave's own
// A-operand registers (v_mfma_f32_16x16x4_f32: 16 rows x 4 k per instruction), and no barrier orders
// activations at all.  What the four waves of a workgroup share is the layer's 96 x 96 panel, read-only:
// it arrives by LDS-DMA in the layout it has in memory, one step ahead into the other of two buffers,
// and the single barrier per step is for that copy alone.  Each wave runs 144 MFMAs per step for its
// 16 x 96 outputs with the A operand in 24 registers.  (The 32-sample form above splits a layer's columns
// over three waves, needs every wave's output before the next layer starts, and leaves two of the four
// SIMDs with twice the work.)
constexpr int CR_ROWS = 16;                 // samples per wave
constexpr int CR_WAVES = 4;
constexpr int CR_TS = CR_ROWS * CR_WAVES;   // samples per workgroup
constexpr int CR_LD = 100;                  // row stride of the private tile: 16-byte rows, conflict-free column reads
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int CR_PANEL = KP * NP;                        // floats of one layer's panel
constexpr int CR_PER = CR_ROWS * KP / 64;                // 24 elements of a 16 x 96 tile per lane
constexpr int CR_QPL = CR_ROWS * KP / 4 / 64;            // 6 16-byte pieces of its rows per lane
static_assert(CR_PANEL % (256 * 4) == 0, "the panel is copied in whole 16-byte pieces per thread");

// one panel, global memory -> LDS in the layout it has in memory, by all 256 threads of the workgroup
__device__ __forceinline__ void rows_panel_dma(const float* __restrict__ src, float* dst) {
    const int t = threadIdx.x, wave = t >> 6;
#pragma unroll
    for (int i = 0; i < CR_PANEL / (256 * 4); ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (i * 256 + t) * 4),
                                         (__attribute__((address_space(3))) void*)(dst + (i * 256 + wave * 64) * 4), 16, 0, 0);
}

// rows of a wave's private tile to memory, 16 bytes per lane and 384 contiguous bytes per row
__device__ __forceinline__ void rows_store(const float* priv, float* __restrict__ dst, long row_stride, int b0, int B,
                                           int lane) {
    f32x4_t v[CR_QPL];
#pragma unroll
    for (int i = 0; i < CR_QPL; ++i) {
        const int chunk = i * 64 + lane, r = chunk / (KP / 4), c4 = chunk - r * (KP / 4);
        v[i] = *reinterpret_cast<const f32x4_t*>(priv + r * CR_LD + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < CR_QPL; ++i) {
        const int chunk = i * 64 + lane, r = chunk / (KP / 4), c4 = chunk - r * (KP / 4);
        if (b0 + r < B) *reinterpret_cast<f32x4_t*>(dst + (long)(b0 + r) * row_stride + 4 * c4) = v[i];
    }
}

// acc[ct][r] = sum_k tile[4g + r][k] * pan[k][16 ct + c16]: the tile's rows are the A operand (24 registers),
// the panel's values are read four k steps (24 registers) ahead of the 24 MFMAs that use them - left to
// itself the compiler reads two values, waits for LDS, issues two MFMAs, and so on
__device__ __forceinline__ void rows_product(const float* priv, const float* pan, int c16, int g, f32x4_t (&acc)[6]) {
    float af[KP / 4];
#pragma unroll
    for (int kk = 0; kk < KP / 4; ++kk) af[kk] = priv[c16 * CR_LD + 4 * kk + g];
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) acc[ct] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    float bv[2][24];
    auto load_b = [&](int kq, float (&b)[24]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ct = 0; ct < 6; ++ct) b[e * 6 + ct] = pan[(4 * (4 * kq + e) + g) * NP + ct * 16 + c16];
    };
    load_b(0, bv[0]);
#pragma unroll
    for (int kq = 0; kq < KP / 16; ++kq) {
        if (kq + 1 < KP / 16) load_b(kq + 1, bv[(kq + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ct = 0; ct < 6; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * kq + e], bv[kq & 1][e * 6 + ct], acc[ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(256) void chain_fwd_rows_kernel(DecArgs a, const float* __restrict__ z,
                                                             const float* __restrict__ wt,
                                                             const float* __restrict__ drop, float* __restrict__ nodes,
                                                             float* __restrict__ xlat, float* __restrict__ hin) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PANEL = CR_PANEL, PER = CR_PER;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c16 = lane & 15, g = lane >> 4;            // MFMA lane coordinates: row / column 0..15, k group 0..3
    float* priv = smem + 2 * PANEL + wave * CR_ROWS * CR_LD;
    const int N = a.N, steps = a.R * a.NL;
    const int ntiles = (a.B + CR_TS - 1) / CR_TS;
    auto panel_dma = [&](int s, int buf) { rows_panel_dma(wt + (long)s * PANEL, smem + buf * PANEL); };
    if ((int)blockIdx.x >= ntiles) return;
    panel_dma(0, 0);
    int buf = 0;
    auto store_rows = [&](float* dst, long row_stride, int b0) { rows_store(priv, dst, row_stride, b0, a.B, lane); };
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b0 = tile * CR_TS + wave * CR_ROWS;
        // the latent rows with the ones column and zero padding, kept for all channels.  Every load is
        // issued before the first value is used (clamped addresses, the value dropped where there is
        // none): one load at a time costs a memory round trip per element
        float zl[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = i * 64 + lane, r = idx / KP, k = idx - r * KP;
            const bool ok = k < a.L && b0 + r < a.B;
            const float t = z[(long)(ok ? b0 + r : 0) * a.L + (ok ? k : 0)];
            zl[i] = k == KP - 1 ? 1.0f : (ok ? t : 0.0f);
        }
        float xl[6][4];
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) xl[ct][r] = 0.0f;
        for (int s = 0; s < steps; ++s, buf ^= 1) {
            const int r_ch = s / a.NL, l = s - r_ch * a.NL;
            __syncthreads();                             // this step's panel has landed; the other buffer is free
            const bool more = s + 1 < steps || tile + (int)gridDim.x < ntiles;
            if (more) panel_dma(s + 1 < steps ? s + 1 : 0, buf ^ 1);
            __builtin_amdgcn_wave_barrier();
            if (l == 0) {
                // the tile holds the previous channel's node vectors: out they go, in come the latent rows
                if (s > 0) store_rows(nodes + (long)(r_ch - 1) * NP, (long)a.R * NP, b0);
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int idx = i * 64 + lane, r = idx / KP, k = idx - r * KP;
                    priv[r * CR_LD + k] = zl[i];
                }
            }
            // this layer's input rows: (training) the saved copy for the backward pass, and the A operand
            if (hin) store_rows(hin + (long)s * a.B * KP, KP, b0);
            f32x4_t acc[6];
            rows_product(priv, smem + buf * PANEL, c16, g, acc);
            __builtin_amdgcn_wave_barrier();
            // the layer's output into the private tile, as the next layer's input rows or as the node
            // vectors on their way out; the activation is chosen once per step, outside the element loop
            const bool last = l == a.NL - 1;
            auto epilogue = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
                for (int ct = 0; ct < 6; ++ct) {
                    const int n = ct * 16 + c16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = act_of<ACT>(acc[ct][r]);
                        v = n < N ? v : 0.0f;
                        if (!last && n == KP - 1) v = 1.0f;
                        acc[ct][r] = v;
                    }
                }
            };
            switch (layer_act(a.acts, l)) {
                case ACT_SIGMOID: epilogue(std::integral_constant<int, ACT_SIGMOID>{}); break;
                case ACT_TANH: epilogue(std::integral_constant<int, ACT_TANH>{}); break;
                default: epilogue(std::integral_constant<int, ACT_NONE>{}); break;
            }
            if (last && drop) {
#pragma unroll
                for (int ct = 0; ct < 6; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = ct * 16 + c16, row = 4 * g + r;
                        const bool ok = n < N && b0 + row < a.B;
                        const float d = drop[ok ? ((long)r_ch * a.B + b0 + row) * N + n : 0];
                        acc[ct][r] *= ok ? d : 1.0f;
                    }
            }
            if (last && r_ch < a.xlat_r) {
#pragma unroll
                for (int ct = 0; ct < 6; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xl[ct][r] += acc[ct][r];
            }
#pragma unroll
            for (int ct = 0; ct < 6; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) priv[(4 * g + r) * CR_LD + ct * 16 + c16] = acc[ct][r];
            __builtin_amdgcn_wave_barrier();
        }
        store_rows(nodes + (long)(a.R - 1) * NP, (long)a.R * NP, b0);
        if (xlat) {
#pragma unroll
            for (int ct = 0; ct < 6; ++ct) {
                const int n = ct * 16 + c16;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n < N && b0 + 4 * g + r < a.B) xlat[(long)(b0 + 4 * g + r) * N + n] = xl[ct][r];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The chain with its activations in registers (large batches, no dropout, no x_latent).  The product runs transposed,
// Y^T = W^T X^T: the panel is the A operand, the 16 samples of a wave are the columns.  Register r of lane (c16, g) of
// output tile ct is then feature 16 ct + 4 g + r of sample c16 -- and that is exactly a B operand of the next layer if
// its k step (ct, r) is taken to mean "features 16 ct + 4 g + r for g = 0..3": the order of the reduction is free as
// long as the panel's rows are read in the same order.  Nothing crosses LDS between layers, and no barrier is needed.
// A workgroup serves one channel (blockIdx.y) and keeps that channel's NL panels in LDS for its whole life (row pitch
// NP + 4: the four k groups of a read fall on different banks); its waves walk 16-sample tiles independently, several
// per SIMD, so one wave's MFMA chain hides under another's.
constexpr int CG_LD = NP + 4;
constexpr int CG_MAXL = 4;               // panels that fit: 4 x 96 x 100 floats = 150 KB

__global__ __launch_bounds__(512) void chain_fwd_regs_kernel(DecArgs a, const float* __restrict__ z,
                                                             const float* __restrict__ wt, float* __restrict__ nodes,
                                                             float* __restrict__ hin) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = blockDim.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int r_ch = blockIdx.y;
    const int N = a.N;
    {
        // the channel's panels: 18 16-byte pieces per thread and panel, all requested before the first is stored
        const float* src = wt + (long)r_ch * a.NL * KP * NP;
        constexpr int PER = KP * (NP / 4) / 512 + 1;               // 5 pieces of a panel per thread (512 threads)
        for (int l = 0; l < a.NL; ++l) {
            f32x4_t v[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = t + i * 512;
                if (idx < KP * (NP / 4)) v[i] = *reinterpret_cast<const f32x4_t*>(src + (long)l * KP * NP + idx * 4);
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = t + i * 512;
                const int row = idx / (NP / 4), q = idx - row * (NP / 4);
                if (idx < KP * (NP / 4)) *reinterpret_cast<f32x4_t*>(smem + (l * KP + row) * CG_LD + 4 * q) = v[i];
            }
        }
    }
    __syncthreads();
    // the workgroups of a channel split its 16-sample tiles into even contiguous runs; inside a workgroup wave w takes
    // tiles w, w + 8, ...: waves w and w + 4 share a SIMD, and between them they get an even share whatever the count
    const int ntiles = (a.B + CR_ROWS - 1) / CR_ROWS;
    const int t_lo = (int)((long)ntiles * blockIdx.x / gridDim.x), t_hi = (int)((long)ntiles * (blockIdx.x + 1) / gridDim.x);
    for (int tile = t_lo + wave; tile < t_hi; tile += nwaves) {
        const int b = tile * CR_ROWS + c16;
        const bool live = b < a.B;
        float x[6][4];
#pragma unroll
        for (int kq = 0; kq < 6; ++kq)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * kq + 4 * g + r;
                const bool ok = live && f < a.L;
                const float v = z[(long)(ok ? b : 0) * a.L + (ok ? f : 0)];
                x[kq][r] = f == KP - 1 ? 1.0f : (ok ? v : 0.0f);
            }
        for (int l = 0; l < a.NL; ++l) {
            const int step = r_ch * a.NL + l;
            if (hin && live) {
#pragma unroll
                for (int kq = 0; kq < 6; ++kq)
                    *reinterpret_cast<f32x4_t*>(hin + ((long)step * a.B + b) * KP + 16 * kq + 4 * g) =
                        f32x4_t{x[kq][0], x[kq][1], x[kq][2], x[kq][3]};
            }
            f32x4_t acc[6];
#pragma unroll
            for (int ct = 0; ct < 6; ++ct) acc[ct] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            const float* pan = smem + l * KP * CG_LD + 4 * g * CG_LD + c16;
            float wv[2][24];
            auto load_w = [&](int kq, float (&w)[24]) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ct = 0; ct < 6; ++ct) w[r * 6 + ct] = pan[(16 * kq + r) * CG_LD + ct * 16];
            };
            load_w(0, wv[0]);
#pragma unroll
            for (int kq = 0; kq < 6; ++kq) {
                if (kq + 1 < 6) load_w(kq + 1, wv[(kq + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ct = 0; ct < 6; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kq & 1][r * 6 + ct], x[kq][r], acc[ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            const bool last = l == a.NL - 1;
            auto epilogue = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
                for (int ct = 0; ct < 6; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = 16 * ct + 4 * g + r;
                        float v = act_of<ACT>(acc[ct][r]);
                        v = n < N ? v : 0.0f;
                        if (!last && n == KP - 1) v = 1.0f;
                        x[ct][r] = v;
                    }
            };
            switch (layer_act(a.acts, l)) {
                case ACT_SIGMOID: epilogue(std::integral_constant<int, ACT_SIGMOID>{}); break;
                case ACT_TANH: epilogue(std::integral_constant<int, ACT_TANH>{}); break;
                default: epilogue(std::integral_constant<int, ACT_NONE>{}); break;
            }
        }
        if (live) {
#pragma unroll
            for (int kq = 0; kq < 6; ++kq)
                *reinterpret_cast<f32x4_t*>(nodes + ((long)b * a.R + r_ch) * NP + 16 * kq + 4 * g) =
                    f32x4_t{x[kq][0], x[kq][1], x[kq][2], x[kq][3]};
        }
    }
}

// out[b][i*N+j] = f(sum_r g_r[i] g_r[j])  (one wave per sample), or in component mode
// out[b][r][i*N+j] = g_r[i] g_r[j] (one wave per (sample, channel)).  The wave keeps its node
// vectors in LDS and writes the N*N matrix as one flat run of 16-byte quads (the matrix starts at
// a 4-byte phase when N*N is odd; the hardware takes unaligned 16-byte stores), HBM-write bound.
__global__ __launch_bounds__(256) void outer_epilogue_kernel(DecArgs a, const float* __restrict__ nodes,
                                                             const float* __restrict__ wd,
                                                             const float* __restrict__ bi, unsigned magicN,
                                                             float* __restrict__ out) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef v4f v4fu __attribute__((aligned(4)));
    __shared__ float gl[4][MAXR * NP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long wave = (long)blockIdx.x * 4 + wv, nwaves = (long)gridDim.x * 4;
    const int N = a.N, D = N * N, nq = D >> 2;
    const int per = a.comp_mode ? 1 : a.R;                  // node vectors per output matrix
    const long nout = a.comp_mode ? (long)a.B * a.R : a.B;
    float* g = gl[wv];
    for (long m = wave; m < nout; m += nwaves) {
        const float* gv = nodes + m * per * NP;
        __builtin_amdgcn_wave_barrier();                    // the previous matrix is done with g
        for (int i = lane; i < per * NP; i += 64) g[i] = gv[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* orow = out + m * D;
        auto value = [&](int i, int j, int e) {
            float s = 0.0f;
            for (int r = 0; r < per; ++r) s += g[r * NP + i] * g[r * NP + j];
            if (!a.comp_mode) {
                s = act_apply(a.epi_act, s);
                if (a.has_icpt) s = s * wd[e] + bi[e];
                if (a.do_exp) s = expf(s);
            }
            return s;
        };
        for (int q = lane; q < nq; q += 64) {
            const int e0 = q << 2;
            int i = (int)__umulhi((unsigned)e0, magicN);    // e0 / N  (e0 * N < 2^32)
            int j = e0 - i * N;
            v4f o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                o[c] = value(i, j, e0 + c);
                if (++j == N) { j = 0; ++i; }
            }
            __builtin_nontemporal_store(o, reinterpret_cast<v4fu*>(orow) + q);
        }
        const int e = (nq << 2) + lane;                     // the (at most 3) elements past the last quad
        if (e < D) {
            const int i = (int)__umulhi((unsigned)e, magicN);
            orow[e] = value(i, e - i * N, e);
        }
    }
}

// Gradient of the node vectors.  Per output matrix the pre-activation gradient A (N x N) is
//   component mode: A = grad_out[b][r] as it stands (one wave per (sample, channel));
//   sum mode:       A = grad_out[b] chained through exp / intercept / activation of sum_r g_r g_r^T
//                   (one wave per sample, A feeds every channel; grad_times_act is emitted for the
//                   intercept-weight gradient),
// and  dnode_r[i] = sum_j A[i][j] g_r[j]  (row sums: DPP wave reduction, total in lane 63)
//                 + sum_j A[j][i] g_r[j]  (column sums: accumulated per lane),
// times the dropout multiplier.  A is streamed row by row, OG_ROWS rows in flight; lane c owns
// columns c and c+64 (N <= 95).  No LDS; the grid fills the chip, so the pass is HBM bound.
constexpr int OG_ROWS = 8;
template <bool COMP>
__global__ __launch_bounds__(256) void outer_grad_kernel(DecArgs a, const float* __restrict__ gout,
                                                         const float* __restrict__ nodes,
                                                         const float* __restrict__ wd, const float* __restrict__ bi,
                                                         const float* __restrict__ drop, float* __restrict__ ga,
                                                         float* __restrict__ dnode) {
    constexpr int RR = COMP ? 1 : MAXR;
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const int N = a.N, D = N * N;
    const bool c0 = lane < N, c1 = lane + 64 < N;
    const int per = COMP ? 1 : a.R;
    const long nmat = COMP ? (long)a.B * a.R : a.B;
    for (long m = wave; m < nmat; m += nwaves) {
        const float* G = gout + m * D;
        const float* gv = nodes + m * per * NP;
        float g0[RR], g1[RR], v0[RR], v1[RR], u0[RR], u1[RR];
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            g0[r] = (r < per && c0) ? gv[r * NP + lane] : 0.0f;
            g1[r] = (r < per && c1) ? gv[r * NP + lane + 64] : 0.0f;
            v0[r] = v1[r] = u0[r] = u1[r] = 0.0f;
        }
        for (int i0 = 0; i0 < N; i0 += OG_ROWS) {
            float a0[OG_ROWS], a1[OG_ROWS];
#pragma unroll
            for (int q = 0; q < OG_ROWS; ++q) {
                const bool live_row = i0 + q < N;
                a0[q] = (live_row && c0) ? __builtin_nontemporal_load(G + (i0 + q) * N + lane) : 0.0f;
                a1[q] = (live_row && c1) ? __builtin_nontemporal_load(G + (i0 + q) * N + lane + 64) : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < OG_ROWS; ++q) {
                const int i = i0 + q;                               // uniform over the wave
                if (i < N) {
                    float gi[RR];
#pragma unroll
                    for (int r = 0; r < RR; ++r)
                        gi[r] = (r < per) ? __int_as_float(__builtin_amdgcn_readlane(
                                                __float_as_int(i < 64 ? g0[r] : g1[r]), i & 63))
                                          : 0.0f;
                    float e0 = a0[q], e1 = a1[q];
                    if (!COMP) {
                        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                        for (int r = 0; r < RR; ++r) {
                            s0 += gi[r] * g0[r];
                            s1 += gi[r] * g1[r];
                        }
                        const float av0 = act_apply(a.epi_act, s0), av1 = act_apply(a.epi_act, s1);
                        const int e = i * N + lane;
                        float w0 = 1.0f, w1 = 1.0f;
                        if (a.has_icpt) {
                            w0 = c0 ? wd[e] : 0.0f;
                            w1 = c1 ? wd[e + 64] : 0.0f;
                        }
                        if (a.do_exp) {
                            float o0 = av0, o1 = av1;
                            if (a.has_icpt) {
                                o0 = o0 * w0 + (c0 ? bi[e] : 0.0f);
                                o1 = o1 * w1 + (c1 ? bi[e + 64] : 0.0f);
                            }
                            e0 *= expf(o0);
                            e1 *= expf(o1);
                        }
                        if (a.has_icpt) {
                            if (c0) ga[m * D + e] = e0 * av0;
                            if (c1) ga[m * D + e + 64] = e1 * av1;
                            e0 *= w0;
                            e1 *= w1;
                        }
                        e0 = c0 ? e0 * act_slope(a.epi_act, av0) : 0.0f;
                        e1 = c1 ? e1 * act_slope(a.epi_act, av1) : 0.0f;
                    }
#pragma unroll
                    for (int r = 0; r < RR; ++r) {
                        if (r < per) {
                            v0[r] += e0 * gi[r];
                            v1[r] += e1 * gi[r];
                            const float p = wave_sum_dpp(e0 * g0[r] + e1 * g1[r]);
                            const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 63));
                            if (i < 64) u0[r] = (lane == i) ? tot : u0[r];
                            else u1[r] = (lane == i - 64) ? tot : u1[r];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            if (r < per) {
                float o0 = u0[r] + v0[r], o1 = u1[r] + v1[r];
                const long b = COMP ? m / a.R : m;
                const int ch = COMP ? (int)(m - b * a.R) : r;
                if (drop) {
                    const float* dr = drop + ((long)ch * a.B + b) * N;
                    if (c0) o0 *= dr[lane];
                    if (c1) o1 *= dr[lane + 64];
                }
                float* dst = dnode + (b * a.R + ch) * NP;
                dst[lane] = c0 ? o0 : 0.0f;
                if (lane + 64 < NP) dst[lane + 64] = c1 ? o1 : 0.0f;
            }
        }
    }
}

// Chain backward for a tile of 32 samples: per channel, recompute the layer outputs (k-major in
// LDS), then walk the layers backwards: dpre = dh * act'(h) goes to layer_dpre and, n-major in
// LDS, is the A operand of dh_prev = dpre . W (panel Wn[n][k]); the layer's input rows go to
// layer_in.  The latent gradient accumulates over the channels in the MFMA accumulators.
__global__ __launch_bounds__(CH_NT, 2) void chain_bwd_mfma_kernel(DecArgs a, const float* __restrict__ z,
                                                               const float* __restrict__ wt,
                                                               const float* __restrict__ wn,
                                                               const float* __restrict__ drop,
                                                               const float* __restrict__ dnode,
                                                               const float* __restrict__ gxlat, float* __restrict__ dz,
                                                               float* __restrict__ hin, float* __restrict__ dp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* zt = smem;                                   // [KP][CH_LD]
    float* dpb = zt + KP * CH_LD;                       // [NP][CH_LD]
    float* acts = dpb + NP * CH_LD;                     // [NL][KP][CH_LD]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = wave * 32 + (lane & 31);
    const int N = a.N, R = a.R, NL = a.NL;
    const int ntiles = (a.B + CH_TS - 1) / CH_TS;
    if ((int)blockIdx.x >= ntiles) return;
    const int S = R * 2 * NL;                           // panel steps per tile
    // step q of a tile: channel q / 2NL; first NL steps recompute layers 0..NL-1 (panels Wt),
    // the next NL walk layers NL-1..0 backwards (panels Wn)
    auto panel_of = [&](int q) {
        const int r = q / (2 * NL), i = q - r * 2 * NL;
        return i < NL ? wt + ((long)(r * NL + i) * KP) * NP : wn + ((long)(r * NL + (2 * NL - 1 - i)) * NP) * KP;
    };
    float wcur[PANEL_REGS], wnext[PANEL_REGS];
    panel_fetch(panel_of(0), wave * 32, lane, wcur);
    load_latent_tile(z, blockIdx.x * CH_TS, a.B, a.L, zt);
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b0 = tile * CH_TS;
        const bool more_tiles = tile + (int)gridDim.x < ntiles;
        f32x16 dzacc;
#pragma unroll
        for (int j = 0; j < 16; ++j) dzacc[j] = 0.0f;
        for (int r = 0; r < R; ++r) {
            // recompute the chain of channel r
            for (int l = 0; l < NL; ++l) {
                panel_fetch(panel_of(r * 2 * NL + l + 1), wave * 32, lane, wnext);   // a backward step always follows
                const f32x16 acc = panel_product(l == 0 ? zt : acts + (l - 1) * KP * CH_LD, wcur, lane);
                const int act = layer_act(a.acts, l);
                float* dst = acts + l * KP * CH_LD;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float v = (n < N) ? act_apply(act, acc[j]) : 0.0f;
                    if (n == KP - 1) v = 1.0f;
                    dst[n * CH_LD + acc_row(j, lane)] = v;
                }
#pragma unroll
                for (int i = 0; i < PANEL_REGS; ++i) wcur[i] = wnext[i];
                __syncthreads();
            }
            // gradient w.r.t. the final node vector (dropout already applied to dnode)
            f32x16 dh;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int s = acc_row(j, lane);
                float v = 0.0f;
                if (b0 + s < a.B && n < N) {
                    v = dnode[((long)(b0 + s) * R + r) * NP + n];
                    if (gxlat && r < a.xlat_r) {
                        float gx = gxlat[(long)(b0 + s) * N + n];
                        if (drop) gx *= drop[((long)r * a.B + b0 + s) * N + n];
                        v += gx;
                    }
                }
                dh[j] = v;
            }
            for (int l = NL - 1; l >= 0; --l) {
                const int q = r * 2 * NL + (2 * NL - 1 - l);
                const bool more = q + 1 < S || more_tiles;
                if (more) panel_fetch(panel_of((q + 1) % S), wave * 32, lane, wnext);
                const int act = layer_act(a.acts, l);
                const long row0 = ((long)(r * NL + l) * a.B + b0);
                const float* hl = acts + l * KP * CH_LD;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int s = acc_row(j, lane);
                    const float v = (n < N) ? dh[j] * act_slope(act, hl[n * CH_LD + s]) : 0.0f;
                    dpb[n * CH_LD + s] = v;
                    if (b0 + s < a.B) dp[(row0 + s) * NP + n] = v;
                }
                // the layer's input rows (with the ones column that yields the bias gradient)
                const float* src = (l == 0) ? zt : acts + (l - 1) * KP * CH_LD;
                for (int i = t; i < CH_TS * KP; i += CH_NT) {
                    const int s = i / KP, k = i - s * KP;
                    if (b0 + s < a.B) hin[(row0 + s) * KP + k] = src[k * CH_LD + s];
                }
                __syncthreads();
                // dh_prev[s][k] = sum_n dpre[s][n] * Wn[n][k]
                const f32x16 acc = panel_product(dpb, wcur, lane);
                if (l == 0) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) dzacc[j] += acc[j];
                } else {
                    dh = acc;
                }
                if (more) {
#pragma unroll
                    for (int i = 0; i < PANEL_REGS; ++i) wcur[i] = wnext[i];
                }
                if (q == S - 1 && more_tiles) load_latent_tile(z, (tile + (int)gridDim.x) * CH_TS, a.B, a.L, zt);
                __syncthreads();
            }
        }
        if (n < a.L) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int s = acc_row(j, lane);
                if (b0 + s < a.B) dz[(long)(b0 + s) * a.L + n] = dzacc[j];
            }
        }
    }
}

// Chain backward when the forward pass kept every layer's input rows (layer_in): nothing is
// recomputed.  The activation slope of layer l comes from its output, which is the next layer's
// saved input (the node vectors for the last layer), read from global memory one step ahead; only
// the pre-activation gradient tile lives in LDS, as the A operand of dh_prev = dpre . W.
__global__ __launch_bounds__(CH_NT, 2) void chain_bwd_saved_kernel(DecArgs a, const float* __restrict__ wn,
                                                                const float* __restrict__ dnode,
                                                                const float* __restrict__ gxlat,
                                                                const float* __restrict__ hin,
                                                                const float* __restrict__ nodes,
                                                                float* __restrict__ dz, float* __restrict__ dp) {
    __shared__ __attribute__((aligned(16))) float dpb[NP * CH_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = wave * 32 + (lane & 31);
    const int N = a.N, R = a.R, NL = a.NL;
    const int ntiles = (a.B + CH_TS - 1) / CH_TS;
    if ((int)blockIdx.x >= ntiles) return;
    // with few tiles (small batches) the channels are spread over gridDim.y workgroups, each writing
    // its own partial latent gradient (dz then points at [gridDim.y][B][L], added up afterwards)
    const int rper = (R + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r_begin = blockIdx.y * rper, r_end = min(R, r_begin + rper);
    if (r_begin >= r_end) return;
    dz += (long)blockIdx.y * a.B * a.L;
    const int q_begin = r_begin * NL, S = r_end * NL;   // panel steps q of a tile: channel q / NL, layer NL-1 - q % NL
    auto panel_of = [&](int q) {
        const int r = q / NL, l = NL - 1 - (q - r * NL);
        return wn + ((long)(r * NL + l) * NP) * KP;
    };
    // outputs of layer l of channel r for this lane's 16 (sample, column) pairs
    auto fetch_h = [&](int b0, int r, int l, float (&h)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int s = acc_row(j, lane);
            float v = 0.0f;
            if (b0 + s < a.B && n < N)
                v = (l == NL - 1) ? nodes[((long)(b0 + s) * R + r) * NP + n]
                                  : hin[((long)(r * NL + l + 1) * a.B + b0 + s) * KP + n];
            h[j] = v;
        }
    };
    float wcur[PANEL_REGS], wnext[PANEL_REGS], hcur[16], hnext[16];
    panel_fetch(panel_of(q_begin), wave * 32, lane, wcur);
    fetch_h(blockIdx.x * CH_TS, r_begin, NL - 1, hcur);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b0 = tile * CH_TS;
        const bool more_tiles = tile + (int)gridDim.x < ntiles;
        f32x16 dzacc;
#pragma unroll
        for (int j = 0; j < 16; ++j) dzacc[j] = 0.0f;
        for (int r = r_begin; r < r_end; ++r) {
            f32x16 dh;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int s = acc_row(j, lane);
                float v = 0.0f;
                if (b0 + s < a.B && n < N) {
                    v = dnode[((long)(b0 + s) * R + r) * NP + n];
                    if (gxlat && r < a.xlat_r) v += gxlat[(long)(b0 + s) * N + n];
                }
                dh[j] = v;
            }
            for (int l = NL - 1; l >= 0; --l) {
                const int q = r * NL + (NL - 1 - l);
                const bool more = q + 1 < S || more_tiles;
                if (more) {
                    const int qn = q + 1 < S ? q + 1 : q_begin, rn = qn / NL, ln = NL - 1 - (qn - rn * NL);
                    panel_fetch(panel_of(qn), wave * 32, lane, wnext);
                    fetch_h(q + 1 < S ? b0 : (tile + (int)gridDim.x) * CH_TS, rn, ln, hnext);
                }
                const int act = layer_act(a.acts, l);
                const long row0 = ((long)(r * NL + l) * a.B + b0);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int s = acc_row(j, lane);
                    const float v = (n < N) ? dh[j] * act_slope(act, hcur[j]) : 0.0f;
                    dpb[n * CH_LD + s] = v;
                    if (b0 + s < a.B) dp[(row0 + s) * NP + n] = v;
                }
                __syncthreads();
                // dh_prev[s][k] = sum_n dpre[s][n] * Wn[n][k]
                const f32x16 acc = panel_product(dpb, wcur, lane);
                if (l == 0) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) dzacc[j] += acc[j];
                } else {
                    dh = acc;
                }
                if (more) {
#pragma unroll
                    for (int i = 0; i < PANEL_REGS; ++i) wcur[i] = wnext[i];
#pragma unroll
                    for (int j = 0; j < 16; ++j) hcur[j] = hnext[j];
                }
                __syncthreads();
            }
        }
        if (n < a.L) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int s = acc_row(j, lane);
                if (b0 + s < a.B) dz[(long)(b0 + s) * a.L + n] = dzacc[j];
            }
        }
    }
}

// The saved-rows backward for large batches, in the form of chain_fwd_rows_kernel: every wave walks its own
// 16 samples through all channels, last layer first.  A step turns the gradient of a layer's output into
// the pre-activation gradient (slope from the layer's saved output), leaves that in the wave's private
// tile - from where it goes to memory as whole rows for the weight gradient, and into the A operand - and
// multiplies by the layer's [n][k] panel, which the workgroup's waves share (LDS-DMA, one step ahead).  The
// product is the gradient of the layer below in the register layout the next step's slope multiplies.  The
// saved outputs arrive as whole rows one step ahead and turn into that layout through the private tile.
__global__ __launch_bounds__(256) void chain_bwd_rows_kernel(DecArgs a, const float* __restrict__ wn,
                                                             const float* __restrict__ dnode,
                                                             const float* __restrict__ gxlat,
                                                             const float* __restrict__ hin,
                                                             const float* __restrict__ nodes, float* __restrict__ dz,
                                                             float* __restrict__ dp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PANEL = CR_PANEL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    float* priv = smem + 2 * PANEL + wave * CR_ROWS * CR_LD;
    const int N = a.N, R = a.R, NL = a.NL, steps = R * NL;
    const int ntiles = (a.B + CR_TS - 1) / CR_TS;
    if ((int)blockIdx.x >= ntiles) return;
    // step q of a tile: channel q / NL, layer NL - 1 - q % NL
    auto panel_dma = [&](int q, int buf) {
        const int r = q / NL, l = NL - 1 - (q - r * NL);
        rows_panel_dma(wn + (long)(r * NL + l) * PANEL, smem + buf * PANEL);
    };
    // the outputs of step q's layer, as 16-byte pieces of whole rows (rows past the batch: the last row's)
    auto fetch_h = [&](int q, int b0, f32x4_t (&v)[CR_QPL]) {
        const int r = q / NL, l = NL - 1 - (q - r * NL);
        const float* src = l == NL - 1 ? nodes + (long)r * NP : hin + (long)(r * NL + l + 1) * a.B * KP;
        const long stride = l == NL - 1 ? (long)R * NP : KP;
#pragma unroll
        for (int i = 0; i < CR_QPL; ++i) {
            const int chunk = i * 64 + lane, rr = chunk / (KP / 4), c4 = chunk - rr * (KP / 4);
            const int row = b0 + rr < a.B ? b0 + rr : a.B - 1;
            v[i] = *reinterpret_cast<const f32x4_t*>(src + (long)row * stride + 4 * c4);
        }
    };
    // the gradient of channel r's node vectors in the accumulator layout, zero outside the batch and the nodes
    auto fetch_dnode = [&](int r, int b0, f32x4_t (&v)[6]) {
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int n = ct * 16 + c16, row = b0 + 4 * g + rr;
                const bool ok = n < N && row < a.B;
                float t = dnode[ok ? ((long)row * R + r) * NP + n : 0];
                if (gxlat && r < a.xlat_r) t += gxlat[ok ? (long)row * N + n : 0];
                v[ct][rr] = ok ? t : 0.0f;
            }
    };
    panel_dma(0, 0);
    int buf = 0;
    f32x4_t hq[CR_QPL], dn[6], dh[6];
    {
        const int b0 = blockIdx.x * CR_TS + wave * CR_ROWS;
        fetch_h(0, b0, hq);
        fetch_dnode(0, b0, dn);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b0 = tile * CR_TS + wave * CR_ROWS;
        const bool more_tiles = tile + (int)gridDim.x < ntiles;
        const int b0_next = b0 + (int)gridDim.x * CR_TS;
        f32x4_t dzacc[6];
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) dzacc[ct] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        for (int q = 0; q < steps; ++q, buf ^= 1) {
            const int r = q / NL, l = NL - 1 - (q - r * NL);
            __syncthreads();                             // this step's panel has landed; the other buffer is free
            const bool more = q + 1 < steps || more_tiles;
            if (more) panel_dma(q + 1 < steps ? q + 1 : 0, buf ^ 1);
            __builtin_amdgcn_wave_barrier();
            // the layer's outputs: rows -> private tile -> accumulator layout
#pragma unroll
            for (int i = 0; i < CR_QPL; ++i) {
                const int chunk = i * 64 + lane, rr = chunk / (KP / 4), c4 = chunk - rr * (KP / 4);
                *reinterpret_cast<f32x4_t*>(priv + rr * CR_LD + 4 * c4) = hq[i];
            }
            __builtin_amdgcn_wave_barrier();
            f32x4_t h[6];
#pragma unroll
            for (int ct = 0; ct < 6; ++ct)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) h[ct][rr] = priv[(4 * g + rr) * CR_LD + ct * 16 + c16];
            if (l == NL - 1) {
#pragma unroll
                for (int ct = 0; ct < 6; ++ct) dh[ct] = dn[ct];
            }
            // what the next step needs from memory, under this step's product
            if (more) {
                fetch_h(q + 1 < steps ? q + 1 : 0, q + 1 < steps ? b0 : b0_next, hq);
                if (l == 0) fetch_dnode(q + 1 < steps ? r + 1 : 0, q + 1 < steps ? b0 : b0_next, dn);
            }
            auto slope = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
                for (int ct = 0; ct < 6; ++ct)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float v = dh[ct][rr] * act_slope(ACT, h[ct][rr]);
                        h[ct][rr] = ct * 16 + c16 < N ? v : 0.0f;
                    }
            };
            switch (layer_act(a.acts, l)) {
                case ACT_SIGMOID: slope(std::integral_constant<int, ACT_SIGMOID>{}); break;
                case ACT_TANH: slope(std::integral_constant<int, ACT_TANH>{}); break;
                default: slope(std::integral_constant<int, ACT_NONE>{}); break;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ct = 0; ct < 6; ++ct)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) priv[(4 * g + rr) * CR_LD + ct * 16 + c16] = h[ct][rr];
            __builtin_amdgcn_wave_barrier();
            rows_store(priv, dp + (long)(r * NL + l) * a.B * NP, NP, b0, a.B, lane);
            // gradient of the layer's input rows: dpre . Wn
            f32x4_t acc[6];
            rows_product(priv, smem + buf * PANEL, c16, g, acc);
            if (l == 0) {
#pragma unroll
                for (int ct = 0; ct < 6; ++ct) dzacc[ct] += acc[ct];
            } else {
#pragma unroll
                for (int ct = 0; ct < 6; ++ct) dh[ct] = acc[ct];
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int ct = 0; ct < 6; ++ct)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int n = ct * 16 + c16, row = b0 + 4 * g + rr;
                if (n < a.L && row < a.B) dz[(long)row * a.L + n] = dzacc[ct][rr];
            }
    }
}

// The backward twin of chain_fwd_regs_kernel (large batches, saved layer inputs, no dropout, no x_latent).  With the
// product transposed, dX^T = W dPre^T, the gradient of a layer's output sits in the accumulators as features
// 16 ct + 4 g + r of sample c16 -- the layout in which the layer's saved output and the pre-activation gradient are 16
// contiguous bytes of their rows in memory (no trip through LDS), and, under the permuted reduction order, the B
// operand of the product with the layer's [n][k] panel.  A workgroup serves one channel and keeps its NL panels in LDS;
// the latent gradient of a channel goes to dz_parts [R][B][L], which add_parts_kernel sums in channel order.
__global__ __launch_bounds__(512) void chain_bwd_regs_kernel(DecArgs a, const float* __restrict__ wn,
                                                             const float* __restrict__ dnode, const float* __restrict__ hin,
                                                             const float* __restrict__ nodes, float* __restrict__ dz_parts,
                                                             float* __restrict__ dp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = blockDim.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int r_ch = blockIdx.y;
    const int N = a.N, NL = a.NL;
    {
        const float* src = wn + (long)r_ch * NL * CR_PANEL;
        constexpr int PER = NP * (KP / 4) / 512 + 1;
        for (int l = 0; l < NL; ++l) {
            f32x4_t v[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = t + i * 512;
                if (idx < NP * (KP / 4)) v[i] = *reinterpret_cast<const f32x4_t*>(src + (long)l * CR_PANEL + idx * 4);
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = t + i * 512;
                const int row = idx / (KP / 4), q = idx - row * (KP / 4);
                if (idx < NP * (KP / 4)) *reinterpret_cast<f32x4_t*>(smem + (l * NP + row) * CG_LD + 4 * q) = v[i];
            }
        }
    }
    __syncthreads();
    const int ntiles = (a.B + CR_ROWS - 1) / CR_ROWS;
    const int t_lo = (int)((long)ntiles * blockIdx.x / gridDim.x), t_hi = (int)((long)ntiles * (blockIdx.x + 1) / gridDim.x);
    for (int tile = t_lo + wave; tile < t_hi; tile += nwaves) {
        const int b = tile * CR_ROWS + c16;
        const bool live = b < a.B;
        const long bb = live ? b : 0;
        f32x4_t dh[6];
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) dh[ct] = *reinterpret_cast<const f32x4_t*>(dnode + (bb * a.R + r_ch) * NP + 16 * ct + 4 * g);
        // a layer's saved output: the node vectors, or the input rows of the layer above; fetched one layer ahead, under
        // the product of the layer above
        f32x4_t hn[6];
        auto fetch_h = [&](int l) {
            const float* hsrc = l == NL - 1 ? nodes + (bb * a.R + r_ch) * NP : hin + ((long)(r_ch * NL + l + 1) * a.B + bb) * KP;
#pragma unroll
            for (int ct = 0; ct < 6; ++ct) hn[ct] = *reinterpret_cast<const f32x4_t*>(hsrc + 16 * ct + 4 * g);
        };
        fetch_h(NL - 1);
        for (int l = NL - 1; l >= 0; --l) {
            const int step = r_ch * NL + l;
            f32x4_t h[6];
#pragma unroll
            for (int ct = 0; ct < 6; ++ct) h[ct] = hn[ct];
            if (l > 0) fetch_h(l - 1);
            auto slope = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
                for (int ct = 0; ct < 6; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = dh[ct][r] * act_slope(ACT, h[ct][r]);
                        h[ct][r] = 16 * ct + 4 * g + r < N ? v : 0.0f;
                    }
            };
            switch (layer_act(a.acts, l)) {
                case ACT_SIGMOID: slope(std::integral_constant<int, ACT_SIGMOID>{}); break;
                case ACT_TANH: slope(std::integral_constant<int, ACT_TANH>{}); break;
                default: slope(std::integral_constant<int, ACT_NONE>{}); break;
            }
            if (live) {
#pragma unroll
                for (int ct = 0; ct < 6; ++ct)
                    *reinterpret_cast<f32x4_t*>(dp + ((long)step * a.B + b) * NP + 16 * ct + 4 * g) = h[ct];
            }
            // gradient of the layer's input: rows of the [n][k] panel are the reduction index, read in the order (ct, r, g)
            f32x4_t acc[6];
#pragma unroll
            for (int it = 0; it < 6; ++it) acc[it] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            const float* pan = smem + l * NP * CG_LD + 4 * g * CG_LD + c16;
            float wv[2][24];
            auto load_w = [&](int ct, float (&w)[24]) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int it = 0; it < 6; ++it) w[r * 6 + it] = pan[(16 * ct + r) * CG_LD + it * 16];
            };
            load_w(0, wv[0]);
#pragma unroll
            for (int ct = 0; ct < 6; ++ct) {
                if (ct + 1 < 6) load_w(ct + 1, wv[(ct + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int it = 0; it < 6; ++it)
                        acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ct & 1][r * 6 + it], h[ct][r], acc[it], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int it = 0; it < 6; ++it) dh[it] = acc[it];
        }
        if (live) {
            float* dst = dz_parts + ((long)r_ch * a.B + b) * a.L;
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int f = 16 * it + 4 * g;
                if (f + 3 < a.L && (a.L & 3) == 0) {
                    *reinterpret_cast<f32x4_t*>(dst + f) = dh[it];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (f + r < a.L) dst[f + r] = dh[it][r];
                }
            }
        }
    }
}

// out[i] = sum_p part[p][i] in index order (the per-channel partial latent gradients)
__global__ __launch_bounds__(256) void add_parts_kernel(const float* __restrict__ part, int parts, long n,
                                                        float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = part[i];
    for (int p = 1; p < parts; ++p) s += part[(long)p * n + i];
    out[i] = s;
}

// column sums over the batch: out0[e] = sum_b a[b][e], out1[e] = sum_b c[b][e]  (fixed order)
__global__ __launch_bounds__(256) void colsum2_partial_kernel(const float* __restrict__ a, const float* __restrict__ c,
                                                              int B, int D, int rows_per_block,
                                                              float* __restrict__ part) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int b0 = blockIdx.y * rows_per_block;
    const int b1 = min(B, b0 + rows_per_block);
    if (e >= D) return;
    float s0 = 0.f, s1 = 0.f;
    for (int b = b0; b < b1; ++b) {
        s0 += a[(long)b * D + e];
        s1 += c[(long)b * D + e];
    }
    part[((long)blockIdx.y * 2 + 0) * D + e] = s0;
    part[((long)blockIdx.y * 2 + 1) * D + e] = s1;
}
__global__ __launch_bounds__(256) void colsum2_final_kernel(const float* __restrict__ part, int nparts, int D,
                                                            float* __restrict__ out0, float* __restrict__ out1) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= D) return;
    float s0 = 0.f, s1 = 0.f;
    for (int p = 0; p < nparts; ++p) {
        s0 += part[((long)p * 2 + 0) * D + e];
        s1 += part[((long)p * 2 + 1) * D + e];
    }
    out0[e] = s0;
    out1[e] = s1;
}

// elementwise helpers for the dense-intercept variant (function/VAE_CNN_NN.py:187-188)
__global__ __launch_bounds__(256) void exp_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = expf(x[i]);
}
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = a[i] * b[i];
}

static int check_dec(const char* fn, int B, int N, int L, int R, int NL) {
    MV_REQUIRE(B >= 0, E_ARG, "%s: negative batch", fn);
    MV_REQUIRE(N >= 2 && N <= NP - 1, E_SHAPE, "%s: n_nodes %d outside [2,%d]", fn, N, NP - 1);
    MV_REQUIRE(L >= 1 && L <= KP - 1, E_SHAPE, "%s: latent_dim %d outside [1,%d]", fn, L, KP - 1);
    MV_REQUIRE(R >= 1 && R <= MAXR, E_SHAPE, "%s: channels %d outside [1,%d]", fn, R, MAXR);
    MV_REQUIRE(NL >= 1 && NL <= MAXL, E_SHAPE, "%s: layers %d outside [1,%d]", fn, NL, MAXL);
    return 0;
}

static size_t chain_bwd_lds_bytes(int NL) {
    return sizeof(float) * ((size_t)KP * CH_LD + (size_t)NP * CH_LD + (size_t)NL * KP * CH_LD);
}

static unsigned wave_grid(long items) {          // one wave per item, four waves per workgroup
    const long blocks = (items + 3) / 4;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

}  // namespace mdnvae

using namespace mdnvae;

extern "C" {

size_t mdnvae_decoder_panel_floats(int R, int NL) { return (size_t)R * NL * KP * NP; }

int mdnvae_decoder_pack_linear(const float* w, const float* mask, const float* bias, int n_out, int K, float* wt,
                               float* wn, void* stream) {
    MV_REQUIRE(w && wt && wn, E_ARG, "mdnvae_decoder_pack_linear: null pointer");
    MV_REQUIRE(n_out >= 1 && n_out <= NP - 1 && K >= 1 && K <= KP - 1, E_SHAPE,
               "mdnvae_decoder_pack_linear: weight [%d,%d] outside the %dx%d panel", n_out, K, NP - 1, KP - 1);
    hipLaunchKernelGGL(pack_linear_kernel, dim3((KP * NP + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, mask, bias,
                       n_out, K, wt, wn);
    MV_LAUNCH_CHECK("pack_linear_kernel");
    return 0;
}

int mdnvae_decoder_pack_gcn(const float* w, const int* adj, const float* bias, int N, int kn, float* wt, float* wn,
                            void* stream) {
    MV_REQUIRE(w && adj && wt && wn, E_ARG, "mdnvae_decoder_pack_gcn: null pointer");
    MV_REQUIRE(N >= 2 && N <= NP - 1 && kn >= 1 && kn <= N, E_SHAPE, "mdnvae_decoder_pack_gcn: N=%d k=%d unsupported", N,
               kn);
    hipLaunchKernelGGL(pack_gcn_kernel, dim3((KP * NP + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, adj, bias, N,
                       kn, wt, wn);
    MV_LAUNCH_CHECK("pack_gcn_kernel");
    return 0;
}

int mdnvae_decoder_unpack_linear_grad(const float* dwt, const float* mask, int n_out, int K, float* dw, float* db,
                                      void* stream) {
    MV_REQUIRE(dwt && dw, E_ARG, "mdnvae_decoder_unpack_linear_grad: null pointer");
    MV_REQUIRE(n_out >= 1 && n_out <= NP - 1 && K >= 1 && K <= KP - 1, E_SHAPE,
               "mdnvae_decoder_unpack_linear_grad: weight [%d,%d] outside the panel", n_out, K);
    hipLaunchKernelGGL(unpack_linear_grad_kernel, dim3((n_out * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, dwt,
                       mask, n_out, K, dw, db);
    MV_LAUNCH_CHECK("unpack_linear_grad_kernel");
    return 0;
}

int mdnvae_decoder_unpack_gcn_grad(const float* dwt, const int* adj, int N, int kn, float* dw, float* db,
                                   void* stream) {
    MV_REQUIRE(dwt && adj && dw, E_ARG, "mdnvae_decoder_unpack_gcn_grad: null pointer");
    MV_REQUIRE(N >= 2 && N <= NP - 1 && kn >= 1 && kn <= N, E_SHAPE, "mdnvae_decoder_unpack_gcn_grad: N=%d k=%d", N, kn);
    hipLaunchKernelGGL(unpack_gcn_grad_kernel, dim3((N * kn + 255) / 256), dim3(256), 0, (hipStream_t)stream, dwt, adj,
                       N, kn, dw, db);
    MV_LAUNCH_CHECK("unpack_gcn_grad_kernel");
    return 0;
}

static int check_layer_tables(const char* who, int count, const int* kind, const int* n_out, const int* k) {
    MV_REQUIRE(count >= 1 && count <= MAX_LAYERS, E_SHAPE, "%s: %d layers (1..%d supported)", who, count, MAX_LAYERS);
    MV_REQUIRE(kind && n_out && k, E_ARG, "%s: null table", who);
    for (int i = 0; i < count; ++i) {
        MV_REQUIRE(kind[i] == 0 || kind[i] == 1, E_ARG, "%s: layer %d has kind %d (0 = linear, 1 = gcn)", who, i, kind[i]);
        if (kind[i] == 0)
            MV_REQUIRE(n_out[i] >= 1 && n_out[i] <= NP - 1 && k[i] >= 1 && k[i] <= KP - 1, E_SHAPE,
                       "%s: layer %d weight [%d,%d] outside the %dx%d panel", who, i, n_out[i], k[i], NP - 1, KP - 1);
        else
            MV_REQUIRE(n_out[i] >= 2 && n_out[i] <= NP - 1 && k[i] >= 1 && k[i] <= n_out[i], E_SHAPE,
                       "%s: layer %d N=%d k=%d unsupported", who, i, n_out[i], k[i]);
    }
    return 0;
}

int mdnvae_decoder_pack_layers(int count, const int* kind, const void* const* w, const void* const* bias,
                               const void* const* aux, const int* n_out, const int* k, float* wt, float* wn,
                               void* stream) {
    if (int rc = check_layer_tables("mdnvae_decoder_pack_layers", count, kind, n_out, k)) return rc;
    MV_REQUIRE(w && bias && aux && wt && wn, E_ARG, "mdnvae_decoder_pack_layers: null pointer");
    PackTable t{};
    for (int i = 0; i < count; ++i) {
        MV_REQUIRE(w[i] && (kind[i] == 0 || aux[i]), E_ARG, "mdnvae_decoder_pack_layers: layer %d lacks a weight or adjacency", i);
        t.w[i] = static_cast<const float*>(w[i]);
        t.bias[i] = static_cast<const float*>(bias[i]);
        t.aux[i] = aux[i];
        t.n_out[i] = n_out[i];
        t.k[i] = k[i];
        t.kind[i] = kind[i];
    }
    hipLaunchKernelGGL(pack_layers_kernel, dim3((KP * NP + 255) / 256, count), dim3(256), 0, (hipStream_t)stream, t, wt, wn);
    MV_LAUNCH_CHECK("pack_layers_kernel");
    return 0;
}

int mdnvae_decoder_unpack_layer_grads(int count, const int* kind, const float* dwt, const void* const* aux,
                                      const int* n_out, const int* k, void* const* dw, void* const* db, int accumulate,
                                      void* stream) {
    if (int rc = check_layer_tables("mdnvae_decoder_unpack_layer_grads", count, kind, n_out, k)) return rc;
    MV_REQUIRE(dwt && aux && dw && db, E_ARG, "mdnvae_decoder_unpack_layer_grads: null pointer");
    UnpackTable t{};
    int most = 0;
    for (int i = 0; i < count; ++i) {
        MV_REQUIRE(dw[i] && (kind[i] == 0 || aux[i]), E_ARG,
                   "mdnvae_decoder_unpack_layer_grads: layer %d lacks a gradient buffer or adjacency", i);
        t.dw[i] = static_cast<float*>(dw[i]);
        t.db[i] = static_cast<float*>(db[i]);
        t.aux[i] = aux[i];
        t.n_out[i] = n_out[i];
        t.k[i] = k[i];
        t.kind[i] = kind[i];
        most = n_out[i] * k[i] > most ? n_out[i] * k[i] : most;
    }
    hipLaunchKernelGGL(unpack_layers_kernel, dim3((most + 255) / 256, count), dim3(256), 0, (hipStream_t)stream, t, dwt,
                       accumulate);
    MV_LAUNCH_CHECK("unpack_layers_kernel");
    return 0;
}

int mdnvae_decoder_fwd(const float* z, const float* wt, int B, int N, int L, int R, int NL, int acts, int epi_act,
                       const float* icpt_w, const float* icpt_b, int do_exp, const float* drop, int xlat_channels,
                       int components, float* out, float* xlat, float* nodes, float* layer_in, void* stream) {
    if (int rc = check_dec("mdnvae_decoder_fwd", B, N, L, R, NL)) return rc;
    if (B == 0) return 0;
    MV_REQUIRE(z && wt, E_ARG, "mdnvae_decoder_fwd: null pointer");
    MV_REQUIRE(out || (!icpt_w && !do_exp && !drop), E_ARG,
               "mdnvae_decoder_fwd: node vectors alone (out = NULL) take no intercept, exp or dropout");
    MV_REQUIRE((icpt_w == nullptr) == (icpt_b == nullptr), E_ARG, "mdnvae_decoder_fwd: intercept needs weight and bias");
    MV_REQUIRE(xlat_channels >= 0 && xlat_channels <= R, E_ARG, "mdnvae_decoder_fwd: xlat_channels %d", xlat_channels);
    MV_REQUIRE(!components || (!icpt_w && !do_exp && epi_act == 0), E_ARG,
               "mdnvae_decoder_fwd: component output takes no epilogue");
    MV_REQUIRE(nodes, E_ARG, "mdnvae_decoder_fwd: nodes [B,R,%d] is required", NP);
    MV_REQUIRE(!xlat_channels || xlat, E_ARG, "mdnvae_decoder_fwd: xlat is null");
    DecArgs a{B, N, L, R, NL, acts, epi_act, icpt_w != nullptr, do_exp, xlat_channels, components};
    const int ntiles = (B + CH_TS - 1) / CH_TS;
    const size_t fwd_lds = sizeof(float) * ((size_t)2 * KP * CH_LD);
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fwd_mfma_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_decoder_fwd: cannot raise the LDS limit: %s", hipGetErrorString(e));
    }
    if (B >= CR_TS * 128 && NL <= CG_MAXL && !drop && xlat_channels == 0) {
        // large batches of the plain chain: activations in registers, one channel per workgroup
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fwd_regs_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_decoder_fwd: cannot raise the LDS limit: %s", hipGetErrorString(e));
        const size_t lds = sizeof(float) * (size_t)NL * KP * CG_LD;
        const int tiles16 = (B + CR_ROWS - 1) / CR_ROWS;
        int gx = 256 / R;                                   // about one workgroup per CU over the R channels
        if (gx < 1) gx = 1;
        if (gx * 8 > tiles16) gx = (tiles16 + 7) / 8;
        hipLaunchKernelGGL(chain_fwd_regs_kernel, dim3(gx, R), dim3(512), lds, (hipStream_t)stream, a, z, wt, nodes, layer_in);
        MV_LAUNCH_CHECK("chain_fwd_regs_kernel");
    } else if (B >= CR_TS * 128) {
        // enough 64-sample tiles for half the chip and more: the wave-per-16-samples form
        {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fwd_rows_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_decoder_fwd: cannot raise the LDS limit: %s", hipGetErrorString(e));
        }
        const int rtiles = (B + CR_TS - 1) / CR_TS;
        const size_t lds = sizeof(float) * ((size_t)2 * KP * NP + (size_t)CR_WAVES * CR_ROWS * CR_LD);
        hipLaunchKernelGGL(chain_fwd_rows_kernel, dim3(rtiles < 1024 ? rtiles : 1024), dim3(256), lds, (hipStream_t)stream, a, z,
                           wt, drop, nodes, xlat_channels ? xlat : nullptr, layer_in);
        MV_LAUNCH_CHECK("chain_fwd_rows_kernel");
    } else {
        // x_latent sums channels inside one workgroup, so the channel split is for decoders without it
        const unsigned ysplit = (ntiles <= 128 && xlat_channels == 0) ? (unsigned)R : 1u;
        hipLaunchKernelGGL(chain_fwd_mfma_kernel, dim3(ntiles < 2048 ? ntiles : 2048, ysplit), dim3(CH_NT), fwd_lds,
                           (hipStream_t)stream, a, z, wt, drop, nodes, xlat_channels ? xlat : nullptr, layer_in);
        MV_LAUNCH_CHECK("chain_fwd_mfma_kernel");
    }
    if (!out) return 0;                               // the caller consumes the node vectors itself
    hipLaunchKernelGGL(outer_epilogue_kernel, dim3(wave_grid(components ? (long)B * R : B)), dim3(256), 0,
                       (hipStream_t)stream, a, nodes, icpt_w, icpt_b, (unsigned)(0x100000000ull / (unsigned)N) + 1u, out);
    MV_LAUNCH_CHECK("outer_epilogue_kernel");
    return 0;
}

size_t mdnvae_decoder_bwd_rows_floats(int B, int R, int NL) { return (size_t)R * NL * B * KP; }

int mdnvae_decoder_bwd(const float* z, const float* wt, const float* wn, int B, int N, int L, int R, int NL, int acts,
                       int epi_act, const float* icpt_w, const float* icpt_b, int do_exp, const float* drop,
                       int xlat_channels, int components, const float* grad_out, const float* grad_xlat, float* grad_z,
                       float* layer_in, float* layer_dpre, float* grad_times_act, const float* nodes,
                       float* node_grad, int layer_in_saved, float* grad_z_parts, void* stream) {
    if (int rc = check_dec("mdnvae_decoder_bwd", B, N, L, R, NL)) return rc;
    if (B == 0) return 0;
    MV_REQUIRE(z && wt && wn && grad_z && layer_in && layer_dpre, E_ARG, "mdnvae_decoder_bwd: null pointer");
    MV_REQUIRE(grad_out || (!icpt_w && !do_exp && !drop), E_ARG,
               "mdnvae_decoder_bwd: a node-vector gradient (grad_out = NULL) takes no intercept, exp or dropout");
    MV_REQUIRE((icpt_w == nullptr) == (icpt_b == nullptr), E_ARG, "mdnvae_decoder_bwd: intercept needs weight and bias");
    MV_REQUIRE(!icpt_w || grad_times_act, E_ARG, "mdnvae_decoder_bwd: intercept gradient needs grad_times_act");
    MV_REQUIRE(!components || (!icpt_w && !do_exp && epi_act == 0), E_ARG,
               "mdnvae_decoder_bwd: component output takes no epilogue");
    MV_REQUIRE(nodes && node_grad, E_ARG,
               "mdnvae_decoder_bwd: needs the forward's node vectors and a node_grad workspace, [B,R,%d] each", NP);
    DecArgs a{B, N, L, R, NL, acts, epi_act, icpt_w != nullptr, do_exp, xlat_channels, components};
    const size_t lds = chain_bwd_lds_bytes(NL);
    MV_REQUIRE(lds <= 160 * 1024, E_SHAPE, "mdnvae_decoder_bwd: needs %zu bytes of LDS", lds);
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_bwd_mfma_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_decoder_bwd: cannot raise the LDS limit: %s", hipGetErrorString(e));
    }
    hipStream_t st = (hipStream_t)stream;
    if (!grad_out) {
        // node_grad already holds the gradient of the node vectors
    } else if (components) {
        hipLaunchKernelGGL(outer_grad_kernel<true>, dim3(wave_grid((long)B * R)), dim3(256), 0, st, a, grad_out, nodes,
                           icpt_w, icpt_b, drop, grad_times_act, node_grad);
    } else {
        hipLaunchKernelGGL(outer_grad_kernel<false>, dim3(wave_grid(B)), dim3(256), 0, st, a, grad_out, nodes, icpt_w,
                           icpt_b, drop, grad_times_act, node_grad);
    }
    MV_LAUNCH_CHECK("outer_grad_kernel");
    const int ntiles = (B + CH_TS - 1) / CH_TS;
    if (layer_in_saved && !drop && B >= CR_TS * 128 && NL <= CG_MAXL && xlat_channels == 0 && grad_z_parts) {
        // large batches of the plain chain: gradients in registers, one channel per workgroup
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_bwd_regs_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_decoder_bwd: cannot raise the LDS limit: %s", hipGetErrorString(e));
        const size_t regs_lds = sizeof(float) * (size_t)NL * NP * CG_LD;
        const int tiles16 = (B + CR_ROWS - 1) / CR_ROWS;
        int gx = 256 / R;
        if (gx < 1) gx = 1;
        if (gx * 8 > tiles16) gx = (tiles16 + 7) / 8;
        hipLaunchKernelGGL(chain_bwd_regs_kernel, dim3(gx, R), dim3(512), regs_lds, st, a, wn, node_grad, layer_in, nodes,
                           grad_z_parts, layer_dpre);
        MV_LAUNCH_CHECK("chain_bwd_regs_kernel");
        const long nz = (long)B * L;
        hipLaunchKernelGGL(add_parts_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, grad_z_parts, R, nz, grad_z);
        MV_LAUNCH_CHECK("add_parts_kernel");
        return 0;
    }
    if (layer_in_saved && !drop && B >= CR_TS * 128) {
        {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_bwd_rows_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_decoder_bwd: cannot raise the LDS limit: %s", hipGetErrorString(e));
        }
        const int rtiles = (B + CR_TS - 1) / CR_TS;
        const size_t rows_lds = sizeof(float) * ((size_t)2 * KP * NP + (size_t)CR_WAVES * CR_ROWS * CR_LD);
        hipLaunchKernelGGL(chain_bwd_rows_kernel, dim3(rtiles < 1024 ? rtiles : 1024), dim3(256), rows_lds, st, a, wn,
                           node_grad, xlat_channels ? grad_xlat : nullptr, layer_in, nodes, grad_z, layer_dpre);
        MV_LAUNCH_CHECK("chain_bwd_rows_kernel");
        return 0;
    }
    if (layer_in_saved && !drop) {
        // the forward pass left every layer's input rows in layer_in: no recomputation
        // few tiles: one workgroup per (tile, channel), partial latent gradients added in channel order
        const bool split = grad_z_parts && ntiles <= 128 && R > 1;
        hipLaunchKernelGGL(chain_bwd_saved_kernel, dim3(ntiles < 4096 ? ntiles : 4096, split ? R : 1), dim3(CH_NT), 0, st,
                           a, wn, node_grad, xlat_channels ? grad_xlat : nullptr, layer_in, nodes,
                           split ? grad_z_parts : grad_z, layer_dpre);
        MV_LAUNCH_CHECK("chain_bwd_saved_kernel");
        if (split) {
            const long nz = (long)B * L;
            hipLaunchKernelGGL(add_parts_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, grad_z_parts, R, nz,
                               grad_z);
            MV_LAUNCH_CHECK("add_parts_kernel");
        }
        return 0;
    }
    hipLaunchKernelGGL(chain_bwd_mfma_kernel, dim3(ntiles < 2048 ? ntiles : 2048), dim3(CH_NT), lds, st, a, z, wt, wn,
                       drop, node_grad, xlat_channels ? grad_xlat : nullptr, grad_z, layer_in, layer_dpre);
    MV_LAUNCH_CHECK("chain_bwd_mfma_kernel");
    return 0;
}

int mdnvae_outer_sum_fwd(const float* nodes, int B, int R, int N, int epi_act, float* out, void* stream) {
    MV_REQUIRE(B >= 0 && R >= 1 && R <= MAXR && N >= 1 && N <= NP - 1, E_SHAPE,
               "mdnvae_outer_sum_fwd: unsupported shape (B=%d, R=%d, N=%d)", B, R, N);
    if (B == 0) return 0;
    MV_REQUIRE(nodes && out, E_ARG, "mdnvae_outer_sum_fwd: null pointer");
    DecArgs a{B, N, 0, R, 0, 0, epi_act, 0, 0, 0, 0};
    hipLaunchKernelGGL(outer_epilogue_kernel, dim3(wave_grid(B)), dim3(256), 0, (hipStream_t)stream, a, nodes, nullptr,
                       nullptr, (unsigned)(0x100000000ull / (unsigned)N) + 1u, out);
    MV_LAUNCH_CHECK("outer_epilogue_kernel");
    return 0;
}

int mdnvae_outer_sum_bwd(const float* grad_out, const float* nodes, int B, int R, int N, int epi_act, float* grad_nodes,
                         void* stream) {
    MV_REQUIRE(B >= 0 && R >= 1 && R <= MAXR && N >= 1 && N <= NP - 1, E_SHAPE,
               "mdnvae_outer_sum_bwd: unsupported shape (B=%d, R=%d, N=%d)", B, R, N);
    if (B == 0) return 0;
    MV_REQUIRE(grad_out && nodes && grad_nodes, E_ARG, "mdnvae_outer_sum_bwd: null pointer");
    DecArgs a{B, N, 0, R, 0, 0, epi_act, 0, 0, 0, 0};
    hipLaunchKernelGGL(outer_grad_kernel<false>, dim3(wave_grid(B)), dim3(256), 0, (hipStream_t)stream, a, grad_out, nodes,
                       nullptr, nullptr, nullptr, nullptr, grad_nodes);
    MV_LAUNCH_CHECK("outer_grad_kernel");
    return 0;
}

size_t mdnvae_colsum2_workspace_floats(int B, int D) {
    const int rows = 64;
    return (size_t)((B + rows - 1) / rows) * 2 * D;
}

int mdnvae_colsum2(const float* a, const float* c, int B, int D, float* workspace, float* out_a, float* out_c,
                   void* stream) {
    MV_REQUIRE(B >= 1 && D >= 1, E_ARG, "mdnvae_colsum2: empty input");
    MV_REQUIRE(a && c && workspace && out_a && out_c, E_ARG, "mdnvae_colsum2: null pointer");
    const int rows = 64, nparts = (B + rows - 1) / rows;
    hipLaunchKernelGGL(colsum2_partial_kernel, dim3((D + 255) / 256, nparts), dim3(256), 0, (hipStream_t)stream, a, c, B,
                       D, rows, workspace);
    MV_LAUNCH_CHECK("colsum2_partial_kernel");
    hipLaunchKernelGGL(colsum2_final_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, workspace, nparts,
                       D, out_a, out_c);
    MV_LAUNCH_CHECK("colsum2_final_kernel");
    return 0;
}

int mdnvae_exp(const float* x, float* y, long n, void* stream) {
    MV_REQUIRE(n >= 0, E_ARG, "mdnvae_exp: negative size");
    if (n == 0) return 0;
    MV_REQUIRE(x && y, E_ARG, "mdnvae_exp: null pointer");
    hipLaunchKernelGGL(exp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    MV_LAUNCH_CHECK("exp_kernel");
    return 0;
}

int mdnvae_mul(const float* a, const float* b, float* y, long n, void* stream) {
    MV_REQUIRE(n >= 0, E_ARG, "mdnvae_mul: negative size");
    if (n == 0) return 0;
    MV_REQUIRE(a && b && y, E_ARG, "mdnvae_mul: null pointer");
    hipLaunchKernelGGL(mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
    MV_LAUNCH_CHECK("mul_kernel");
    return 0;
}

}  // extern "C"
