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
// The backward kernel recomputes the chain per tile (saves nothing but inputs), forms the
// symmetric pre-activation gradient of sum_r g_r g_r^T in LDS one sample at a time, walks
// the chains backwards, and emits per-layer (input activation, pre-activation gradient)
// rows; the weight/bias gradients are then [K x B].[B x N] library GEMMs over those rows.
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

constexpr int NP = MDNVAE_DEC_NP;   // padded node count
constexpr int KP = MDNVAE_DEC_KP;   // padded fan-in; row KP-1 = bias
constexpr int MAXL = 4;             // layers per channel
constexpr int MAXR = 8;
static_assert(NP == KP, "layer outputs are reused as the next layer's input rows");
constexpr int DPB_FLOATS = 1024;   // >= BWD_TS*NP and >= MAXR*128

constexpr int ACT_NONE = 0, ACT_SIGMOID = 1, ACT_TANH = 2;

__device__ __forceinline__ float act_apply(int a, float v) {
    if (a == ACT_SIGMOID) return sigmoidf_(v);
    if (a == ACT_TANH) return tanhf(v);
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

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
struct DecArgs {
    int B, N, L, R, NL;
    int acts;        // 2 bits per layer
    int epi_act;     // activation on sum_r g g^T
    int has_icpt;    // out = a*wd + bi
    int do_exp;      // out = exp(out)
    int xlat_r;      // x_latent = sum of the first xlat_r channels (0 = none)
    int comp_mode;   // 1: out[b][r][e] = g_r[i]*g_r[j] per channel (mixture components), no epilogue
};

constexpr int FWD_TS = 16;    // samples per tile
constexpr int FWD_NT = 256;

__global__ __launch_bounds__(FWD_NT) void chain_decoder_fwd_kernel(
    DecArgs a, const float* __restrict__ z, const float* __restrict__ wt, const float* __restrict__ wd,
    const float* __restrict__ bi, const float* __restrict__ drop, float* __restrict__ out,
    float* __restrict__ xlat, float* __restrict__ nodes) {
    __shared__ float abuf[2][KP][FWD_TS];          // layer input, [k][sample]
    __shared__ float gfin[FWD_TS][MAXR][NP];       // final node vectors per channel
    __shared__ __attribute__((aligned(16))) float wl[KP * NP];   // the current layer's weight panel
    const int t = threadIdx.x;
    const int n = t % NP, sg = t / NP;             // chain phase: 2 sample groups of 8 (threads < 192)
    const int N = a.N, L = a.L, D = N * N;
    const int ntiles = (a.B + FWD_TS - 1) / FWD_TS;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b0 = tile * FWD_TS;
        for (int r = 0; r < a.R; ++r) {
            __syncthreads();
            for (int i = t; i < L * FWD_TS; i += FWD_NT) {
                const int s = i / L, k = i - s * L;
                abuf[0][k][s] = (b0 + s < a.B) ? z[(long)(b0 + s) * L + k] : 0.0f;
            }
            __syncthreads();
            int cur = 0;
            for (int l = 0; l < a.NL; ++l) {
                const int K = (l == 0) ? L : N;
                {   // staged by all threads, coalesced: per-thread reads from global were latency bound
                    const float4* src = reinterpret_cast<const float4*>(wt + ((long)(r * a.NL + l) * KP) * NP);
                    for (int i = t; i < KP * NP / 4; i += FWD_NT) reinterpret_cast<float4*>(wl)[i] = src[i];
                }
                __syncthreads();
                const float* w = wl;
                if (t < 2 * NP) {
                    float acc[8];
                    const float bias = w[(KP - 1) * NP + n];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = bias;
#pragma unroll 4
                    for (int k = 0; k < K; ++k) {
                        const float wv = w[k * NP + n];
                        const float4 a0 = *reinterpret_cast<const float4*>(&abuf[cur][k][sg * 8]);
                        const float4 a1 = *reinterpret_cast<const float4*>(&abuf[cur][k][sg * 8 + 4]);
                        acc[0] += wv * a0.x; acc[1] += wv * a0.y; acc[2] += wv * a0.z; acc[3] += wv * a0.w;
                        acc[4] += wv * a1.x; acc[5] += wv * a1.y; acc[6] += wv * a1.z; acc[7] += wv * a1.w;
                    }
                    const int act = layer_act(a.acts, l);
                    if (l == a.NL - 1) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int s = sg * 8 + i;
                            float v = (n < N) ? act_apply(act, acc[i]) : 0.0f;
                            if (drop && n < N && b0 + s < a.B) v *= drop[((long)r * a.B + b0 + s) * N + n];
                            gfin[s][r][n] = v;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) abuf[cur ^ 1][n][sg * 8 + i] = act_apply(act, acc[i]);
                    }
                }
                __syncthreads();
                cur ^= 1;
            }
        }
        if (nodes) {                                   // final node vectors [B][R][NP]
            for (int i = t; i < FWD_TS * a.R * NP; i += FWD_NT) {
                const int s = i / (a.R * NP), rem = i - s * (a.R * NP);
                if (b0 + s < a.B) nodes[(long)(b0 + s) * a.R * NP + rem] = gfin[s][rem / NP][rem % NP];
            }
        }
        // epilogue: out[b][i*N+j] = f(sum_r g_r[i] g_r[j])
        const int di = FWD_NT / N, dj = FWD_NT - di * N;
        for (int s = 0; s < FWD_TS && b0 + s < a.B; ++s) {
            float* orow = out + (long)(b0 + s) * D;
            int i = t / N, j = t - i * N;
            for (int e = t; e < D; e += FWD_NT) {
                if (a.comp_mode) {
                    float* crow = out + (long)(b0 + s) * a.R * D;
                    for (int r = 0; r < a.R; ++r) crow[(long)r * D + e] = gfin[s][r][i] * gfin[s][r][j];
                } else {
                    float sum = 0.0f;
                    for (int r = 0; r < a.R; ++r) sum += gfin[s][r][i] * gfin[s][r][j];
                    float v = act_apply(a.epi_act, sum);
                    if (a.has_icpt) v = v * wd[e] + bi[e];
                    if (a.do_exp) v = expf(v);
                    orow[e] = v;
                }
                i += di;
                j += dj;
                if (j >= N) { j -= N; ++i; }
            }
            if (xlat && t < N) {
                float sum = 0.0f;
                for (int r = 0; r < a.xlat_r; ++r) sum += gfin[s][r][t];
                xlat[(long)(b0 + s) * N + t] = sum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
constexpr int BWD_TS = 8;
constexpr int BWD_NT = 256;


// Component mode, gradient w.r.t. the final node vectors: channel r of sample b has its own
// gradient matrix G = gout[b][r] (N x N), and
//   dnode[i] = sum_j G[i][j] g[j]  (row sums: DPP wave reduction, total in lane 63)
//            + sum_j G[j][i] g[j]  (column sums: accumulated per lane).
// One wave per (sample, channel) streams G row by row, 16 rows (32 loads) in flight; lane c owns
// columns c and c+64 (N <= 95).  No LDS; the grid fills the chip, so the pass is HBM bound.
constexpr int OG_ROWS = 16;
__global__ __launch_bounds__(256) void comp_outer_grad_kernel(const float* __restrict__ gout,
                                                              const float* __restrict__ nodes,
                                                              const float* __restrict__ drop, int B, int R, int N,
                                                              float* __restrict__ dnode) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const bool c0 = lane < N, c1 = lane + 64 < N;
    const int D = N * N;
    for (long pair = wave; pair < (long)B * R; pair += nwaves) {
        const float* G = gout + pair * D;
        const float* gv = nodes + pair * NP;
        const float g0 = c0 ? gv[lane] : 0.0f, g1 = c1 ? gv[lane + 64] : 0.0f;
        float v0 = 0.0f, v1 = 0.0f, u0 = 0.0f, u1 = 0.0f;
        for (int j = 0; j < N; j += OG_ROWS) {
            float a0[OG_ROWS], a1[OG_ROWS];
#pragma unroll
            for (int q = 0; q < OG_ROWS; ++q) {
                const bool live_row = j + q < N;
                a0[q] = (live_row && c0) ? __builtin_nontemporal_load(G + (j + q) * N + lane) : 0.0f;
                a1[q] = (live_row && c1) ? __builtin_nontemporal_load(G + (j + q) * N + lane + 64) : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < OG_ROWS; ++q) {
                const int row = j + q;                                  // uniform over the wave
                if (row < N) {
                    const float gj = __int_as_float(
                        __builtin_amdgcn_readlane(__float_as_int(row < 64 ? g0 : g1), row & 63));
                    v0 += a0[q] * gj;
                    v1 += a1[q] * gj;
                    const float p = wave_sum_dpp(a0[q] * g0 + a1[q] * g1);
                    const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 63));
                    if (row < 64) u0 = (lane == row) ? tot : u0;
                    else u1 = (lane == row - 64) ? tot : u1;
                }
            }
        }
        float o0 = u0 + v0, o1 = u1 + v1;
        if (drop) {
            const int b = (int)(pair / R), r = (int)(pair - (long)b * R);
            const float* dr = drop + ((long)r * B + b) * N;
            if (c0) o0 *= dr[lane];
            if (c1) o1 *= dr[lane + 64];
        }
        dnode[pair * NP + lane] = c0 ? o0 : 0.0f;
        if (lane + 64 < NP) dnode[pair * NP + lane + 64] = c1 ? o1 : 0.0f;
    }
}

// Dynamic LDS layout (floats):
//   acts  [R][NL][TS][NP]   outputs of every layer
//   zt    [TS][KP]          latent rows of the tile
//   dg    [R][TS][NP]       gradient w.r.t. the final node vectors
//   dpb   [TS][NP]          current layer's pre-activation gradient
//   wl    [KP][NP]          weight panel of the layer being worked on (staged by all threads,
//                           coalesced; per-thread reads from global were latency bound)
//   dA    [N][lda]          per-sample gradient of the pre-activation sum (lda odd)
__global__ __launch_bounds__(BWD_NT) void chain_decoder_bwd_kernel(
    DecArgs a, const float* __restrict__ z, const float* __restrict__ wt, const float* __restrict__ wn,
    const float* __restrict__ wd, const float* __restrict__ bi, const float* __restrict__ drop,
    const float* __restrict__ gout, const float* __restrict__ gxlat, float* __restrict__ dz,
    float* __restrict__ hin, float* __restrict__ dp, float* __restrict__ ga, const float* __restrict__ dnode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = a.N, L = a.L, D = N * N, R = a.R, NL = a.NL;
    const int lda = (N & 1) ? N + 2 : N + 1;
    float* acts = smem;
    float* zt = acts + R * NL * BWD_TS * NP;
    float* dg = zt + BWD_TS * KP;
    float* dpb = dg + R * BWD_TS * NP;
    float* wl = dpb + DPB_FLOATS;           // [KP][NP] the current layer's weight panel
    float* dA = wl + KP * NP;               // (sum mode only)
    const int t = threadIdx.x;
    const int n = t % NP, sg = t / NP;      // chain phases: 2 groups of 4 samples (threads < 192)
    const int ntiles = (a.B + BWD_TS - 1) / BWD_TS;
#define ACTS(r, l, s, nn) acts[(((r) * NL + (l)) * BWD_TS + (s)) * NP + (nn)]

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b0 = tile * BWD_TS;
        __syncthreads();
        for (int i = t; i < BWD_TS * KP; i += BWD_NT) {
            const int s = i / KP, k = i - s * KP;
            float v = 0.0f;
            if (b0 + s < a.B) {
                if (k < L) v = z[(long)(b0 + s) * L + k];
                else if (k == KP - 1) v = 1.0f;
            }
            zt[i] = v;
        }
        __syncthreads();
        // ---- phase 1: recompute every chain, keeping all layer outputs
        for (int r = 0; r < R; ++r) {
            for (int l = 0; l < NL; ++l) {
                const int K = (l == 0) ? L : N;
                {
                    const float4* src = reinterpret_cast<const float4*>(wt + ((long)(r * NL + l) * KP) * NP);
                    for (int i = t; i < KP * NP / 4; i += BWD_NT) reinterpret_cast<float4*>(wl)[i] = src[i];
                }
                __syncthreads();
                const float* w = wl;
                if (t < 2 * NP) {
                    float acc[4];
                    const float bias = w[(KP - 1) * NP + n];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = bias;
                    if (l == 0) {
#pragma unroll 4
                        for (int k = 0; k < K; ++k) {
                            const float wv = w[k * NP + n];
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i] += wv * zt[(sg * 4 + i) * KP + k];
                        }
                    } else {
#pragma unroll 4
                        for (int k = 0; k < K; ++k) {
                            const float wv = w[k * NP + n];
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i] += wv * ACTS(r, l - 1, sg * 4 + i, k);
                        }
                    }
                    const int act = layer_act(a.acts, l);
#pragma unroll
                    for (int i = 0; i < 4; ++i) ACTS(r, l, sg * 4 + i, n) = (n < N) ? act_apply(act, acc[i]) : 0.0f;
                }
                __syncthreads();
            }
        }
        // ---- phase 2: per sample, gradient w.r.t. the final node vectors
        if (a.comp_mode) {
            // component mode: computed beforehand by comp_outer_grad_kernel (dropout applied there)
            for (int i = t; i < BWD_TS * R * NP; i += BWD_NT) {
                const int s = i / (R * NP), rem = i - s * (R * NP);
                const int r = rem / NP, nn = rem - r * NP;
                float v = 0.0f;
                if (b0 + s < a.B) {
                    v = dnode[(long)(b0 + s) * R * NP + rem];
                    if (gxlat && r < a.xlat_r && nn < N) {
                        float gx = gxlat[(long)(b0 + s) * N + nn];
                        if (drop) gx *= drop[((long)r * a.B + b0 + s) * N + nn];
                        v += gx;
                    }
                }
                dg[(r * BWD_TS + s) * NP + nn] = v;
            }
            __syncthreads();
        }
        const int di = BWD_NT / N, dj = BWD_NT - di * N;
        for (int s = 0; s < (a.comp_mode ? 0 : BWD_TS); ++s) {
            const bool live = b0 + s < a.B;       // uniform over the workgroup
            // sum mode: one pass with the gradient of the summed matrix feeding every channel;
            // component mode: one pass per channel with that channel's own gradient matrix
            const int npass = a.comp_mode ? R : 1;
            for (int pass = 0; pass < npass; ++pass) {
                if (live) {
                    const float* grow = a.comp_mode ? gout + ((long)(b0 + s) * R + pass) * D : gout + (long)(b0 + s) * D;
                    float* garow = ga ? ga + (long)(b0 + s) * D : nullptr;
                    int i = t / N, j = t - i * N;
                    for (int e = t; e < D; e += BWD_NT) {
                        float g = grow[e];
                        if (!a.comp_mode) {
                            float sum = 0.0f;
                            for (int r = 0; r < R; ++r) {
                                float gi = ACTS(r, NL - 1, s, i), gj = ACTS(r, NL - 1, s, j);
                                if (drop) {
                                    gi *= drop[((long)r * a.B + b0 + s) * N + i];
                                    gj *= drop[((long)r * a.B + b0 + s) * N + j];
                                }
                                sum += gi * gj;
                            }
                            const float av = act_apply(a.epi_act, sum);
                            if (a.do_exp) {
                                float o = av;
                                if (a.has_icpt) o = o * wd[e] + bi[e];
                                g *= expf(o);
                            }
                            if (a.has_icpt) {
                                if (garow) garow[e] = g * av;
                                g *= wd[e];
                            }
                            g *= act_slope(a.epi_act, av);
                        }
                        dA[i * lda + j] = g;
                        i += di;
                        j += dj;
                        if (j >= N) { j -= N; ++i; }
                    }
                }
                __syncthreads();
                if (live) {
                    // dgr[i] = sum_j (dA[i][j] + dA[j][i]) * g_r[j]; two threads per row, j split in halves
                    const int i = t & 127, half = t >> 7;
                    float acc[MAXR];
#pragma unroll
                    for (int r = 0; r < MAXR; ++r) acc[r] = 0.0f;
                    const int r_lo = a.comp_mode ? pass : 0, r_hi = a.comp_mode ? pass + 1 : R;
                    if (i < N) {
                        const int jh = (N + 1) / 2;
                        const int j0 = half * jh, j1 = (half == 0) ? jh : N;
                        for (int j = j0; j < j1; ++j) {
                            const float v = dA[i * lda + j] + dA[j * lda + i];
#pragma unroll
                            for (int r = 0; r < MAXR; ++r)
                                if (r >= r_lo && r < r_hi) {
                                    float gj = ACTS(r, NL - 1, s, j);
                                    if (drop) gj *= drop[((long)r * a.B + b0 + s) * N + j];
                                    acc[r] += v * gj;
                                }
                        }
                    }
                    // combine the two halves through dpb (free at this point)
                    if (half == 1 && i < N)
                        for (int r = r_lo; r < r_hi; ++r) dpb[r * 128 + i] = acc[r];
                    __syncthreads();
                    if (half == 0 && i < N) {
                        for (int r = r_lo; r < r_hi; ++r) {
                            float v = acc[r] + dpb[r * 128 + i];
                            if (gxlat && r < a.xlat_r) v += gxlat[(long)(b0 + s) * N + i];
                            if (drop) v *= drop[((long)r * a.B + b0 + s) * N + i];
                            dg[(r * BWD_TS + s) * NP + i] = v;
                        }
                    }
                } else if (pass == 0) {
                    for (int i = t; i < R * NP; i += BWD_NT) dg[((i / NP) * BWD_TS + s) * NP + (i % NP)] = 0.0f;
                }
                __syncthreads();
            }
        }
        // ---- phase 3: walk each chain backwards
        float dzacc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) {
            float dh[4];
            if (t < 2 * NP) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dh[i] = (n < N) ? dg[(r * BWD_TS + sg * 4 + i) * NP + n] : 0.0f;
            }
            for (int l = NL - 1; l >= 0; --l) {
                const int act = layer_act(a.acts, l);
                const long row0 = ((long)(r * NL + l) * a.B + b0);
                if (t < 2 * NP) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int s = sg * 4 + i;
                        const float h = ACTS(r, l, s, n);
                        const float v = (n < N) ? dh[i] * act_slope(act, h) : 0.0f;
                        dpb[s * NP + n] = v;
                        if (b0 + s < a.B) dp[(row0 + s) * NP + n] = v;
                    }
                }
                // the layer's input rows (with the ones column that yields the bias gradient)
                for (int i = t; i < BWD_TS * KP; i += BWD_NT) {
                    const int s = i / KP, k = i - s * KP;
                    if (b0 + s < a.B) {
                        float v;
                        if (l == 0) v = zt[i];
                        else v = (k == KP - 1) ? 1.0f : ACTS(r, l - 1, s, k);   // KP == NP
                        hin[(row0 + s) * KP + k] = v;
                    }
                }
                {
                    const float4* src = reinterpret_cast<const float4*>(wn + ((long)(r * NL + l) * NP) * KP);
                    for (int i = t; i < NP * KP / 4; i += BWD_NT) reinterpret_cast<float4*>(wl)[i] = src[i];
                }
                __syncthreads();
                // dh_prev[s][k] = sum_n dpb[s][n] * W[k][n]  (Wn[n][k], conflict-free over k)
                if (t < 2 * NP) {
                    const int k = n;
                    const int K = (l == 0) ? L : N;
                    const float* w = wl;
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
                    if (k < K) {
#pragma unroll 4
                        for (int nn = 0; nn < N; ++nn) {
                            const float wv = w[nn * KP + k];
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i] += wv * dpb[(sg * 4 + i) * NP + nn];
                        }
                    }
                    if (l == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) dzacc[i] += acc[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) dh[i] = acc[i];
                    }
                }
                __syncthreads();
            }
        }
        if (t < 2 * NP && n < L) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = sg * 4 + i;
                if (b0 + s < a.B) dz[(long)(b0 + s) * L + n] = dzacc[i];
            }
        }
    }
#undef ACTS
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

static size_t bwd_lds_bytes(int N, int R, int NL) {
    const int lda = (N & 1) ? N + 2 : N + 1;
    size_t fl = (size_t)R * NL * BWD_TS * NP + BWD_TS * KP + (size_t)R * BWD_TS * NP + DPB_FLOATS + (size_t)KP * NP +
                (size_t)N * lda;
    return fl * sizeof(float);
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

int mdnvae_decoder_fwd(const float* z, const float* wt, int B, int N, int L, int R, int NL, int acts, int epi_act,
                       const float* icpt_w, const float* icpt_b, int do_exp, const float* drop, int xlat_channels,
                       int components, float* out, float* xlat, float* nodes, void* stream) {
    if (int rc = check_dec("mdnvae_decoder_fwd", B, N, L, R, NL)) return rc;
    if (B == 0) return 0;
    MV_REQUIRE(z && wt && out, E_ARG, "mdnvae_decoder_fwd: null pointer");
    MV_REQUIRE((icpt_w == nullptr) == (icpt_b == nullptr), E_ARG, "mdnvae_decoder_fwd: intercept needs weight and bias");
    MV_REQUIRE(xlat_channels >= 0 && xlat_channels <= R, E_ARG, "mdnvae_decoder_fwd: xlat_channels %d", xlat_channels);
    MV_REQUIRE(!components || (!icpt_w && !do_exp && epi_act == 0), E_ARG,
               "mdnvae_decoder_fwd: component output takes no epilogue");
    DecArgs a{B, N, L, R, NL, acts, epi_act, icpt_w != nullptr, do_exp, xlat_channels, components};
    const int ntiles = (B + FWD_TS - 1) / FWD_TS;
    const int grid = ntiles < 2048 ? ntiles : 2048;
    hipLaunchKernelGGL(chain_decoder_fwd_kernel, dim3(grid), dim3(FWD_NT), 0, (hipStream_t)stream, a, z, wt, icpt_w,
                       icpt_b, drop, out, xlat_channels ? xlat : nullptr, nodes);
    MV_LAUNCH_CHECK("chain_decoder_fwd_kernel");
    return 0;
}

size_t mdnvae_decoder_bwd_rows_floats(int B, int R, int NL) { return (size_t)R * NL * B * KP; }

int mdnvae_decoder_bwd(const float* z, const float* wt, const float* wn, int B, int N, int L, int R, int NL, int acts,
                       int epi_act, const float* icpt_w, const float* icpt_b, int do_exp, const float* drop,
                       int xlat_channels, int components, const float* grad_out, const float* grad_xlat, float* grad_z,
                       float* layer_in, float* layer_dpre, float* grad_times_act, const float* nodes,
                       float* node_grad, void* stream) {
    if (int rc = check_dec("mdnvae_decoder_bwd", B, N, L, R, NL)) return rc;
    if (B == 0) return 0;
    MV_REQUIRE(z && wt && wn && grad_out && grad_z && layer_in && layer_dpre, E_ARG, "mdnvae_decoder_bwd: null pointer");
    MV_REQUIRE((icpt_w == nullptr) == (icpt_b == nullptr), E_ARG, "mdnvae_decoder_bwd: intercept needs weight and bias");
    MV_REQUIRE(!icpt_w || grad_times_act, E_ARG, "mdnvae_decoder_bwd: intercept gradient needs grad_times_act");
    MV_REQUIRE(!components || (!icpt_w && !do_exp && epi_act == 0), E_ARG,
               "mdnvae_decoder_bwd: component output takes no epilogue");
    MV_REQUIRE(!components || (nodes && node_grad), E_ARG,
               "mdnvae_decoder_bwd: component mode needs the forward's node vectors and a node_grad workspace");
    DecArgs a{B, N, L, R, NL, acts, epi_act, icpt_w != nullptr, do_exp, xlat_channels, components};
    const size_t lds = bwd_lds_bytes(N, R, NL);
    MV_REQUIRE(lds <= 160 * 1024, E_SHAPE, "mdnvae_decoder_bwd: needs %zu bytes of LDS", lds);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(chain_decoder_bwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (components) {
        const long pairs = (long)B * R;
        const unsigned og_grid = (unsigned)((pairs + 3) / 4 < 4096 ? (pairs + 3) / 4 : 4096);
        hipLaunchKernelGGL(comp_outer_grad_kernel, dim3(og_grid), dim3(256), 0, (hipStream_t)stream, grad_out, nodes,
                           drop, B, R, N, node_grad);
        MV_LAUNCH_CHECK("comp_outer_grad_kernel");
    }
    const int ntiles = (B + BWD_TS - 1) / BWD_TS;
    const int grid = ntiles < 1024 ? ntiles : 1024;
    hipLaunchKernelGGL(chain_decoder_bwd_kernel, dim3(grid), dim3(BWD_NT), lds, (hipStream_t)stream, a, z, wt, wn,
                       icpt_w, icpt_b, drop, grad_out, xlat_channels ? grad_xlat : nullptr, grad_z, layer_in,
                       layer_dpre, grad_times_act, node_grad);
    MV_LAUNCH_CHECK("chain_decoder_bwd_kernel");
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
