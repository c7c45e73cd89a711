// Reconstruction loss + KL and their gradients for a decoder whose output is a plain sum of outer
// products, recon[b] = sum_r g_br g_br^T (the multi-channel GCN decoder of function/MM_Gate.py:126-138
// followed by loss_function, function/MM_Gate.py:179-182), taken straight from the node vectors:
// one wave per sample reads x[b] once, MO_ROWS rows in flight, lane c owning columns c and c+64 as a
// packed pair.  Per row i it forms s_ic = sum_r g_ri g_rc, the loss term and its slope e_ic, and
//   v_r[c] = sum_i e_ic g_ri      (column sums, per lane)
//   u_r[i] = sum_c e_ic g_rc      (row sums: the lanes' partial products go to the wave's LDS tile
//                                  [channel * row][lane], lane p adds up row p with 16-byte reads)
// so that the gradient of g_r is u_r + v_r.  The [B,N*N] reconstruction, its gradient and the
// three passes over them (epilogue write, loss read + gradient write, gradient read) do not exist.
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

constexpr int OL_NT = 256;                      // four waves per workgroup, one sample per wave
constexpr int OL_ROWS = 8;
constexpr int OL_NP = MDNVAE_DEC_NP;
constexpr int OL_LD = 68;                       // tile row stride: 16-byte rows, lanes 0..7 cover all banks
constexpr int OL_MAXR = 8;
typedef float ol_v2 __attribute__((ext_vector_type(2)));
typedef float ol_v4 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int ol_wave_floats(int R) { return R * OL_ROWS * OL_LD + R * OL_NP; }

template <int MODE>
__device__ __forceinline__ void ol_term(float s, float x, float& value, float& slope) {
    if (MODE == MDNVAE_RECON_POISSON_RATE) {
        value = s - x * logf(s + 1e-8f);
        slope = 1.0f - x / (s + 1e-8f);
    } else if (MODE == MDNVAE_RECON_POISSON_LOG) {
        const float e = expf(s);
        value = e - x * s;
        slope = e - x;
    } else {
        const float d = s - x;
        value = d * d;
        slope = 2.0f * d;
    }
}

// ---- the sweep over a symmetric x[b] ----
// s and the slope e are then symmetric as well, the row sums equal the column sums the lanes hold,
// and the pass needs no LDS and no cross-lane step.  Lane l owns the adjacent columns 2l and 2l+1
// (one 8-byte load per row) and reads through a buffer descriptor of the sample's N*N floats: a
// lane whose columns do not exist carries an offset past the end, for which the hardware returns 0
// without a branch; the row offset is a scalar.
typedef unsigned ol_u2 __attribute__((ext_vector_type(2)));
template <int MODE, int R, bool FULL>
__device__ __forceinline__ void ol_sym_group(__amdgpu_buffer_rsrc_t rsrc, int off, int row_bytes, bool ok0, bool ok1,
                                             int i0, int rows, const ol_v2 (&g)[R], ol_v2 (&v)[R], float& term) {
    ol_u2 a[OL_ROWS];
#pragma unroll
    for (int q = 0; q < OL_ROWS; ++q)
        if (FULL || q < rows) a[q] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, (i0 + q) * row_bytes, 2 /* nt */);
#pragma unroll
    for (int q = 0; q < OL_ROWS; ++q) {
        if (FULL || q < rows) {
            ol_v2 gi2[R];
            ol_v2 s{0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // i0 is a multiple of OL_ROWS (even): the row's parity is q's
                const float gi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int((q & 1) ? g[r].y : g[r].x),
                                                                         (i0 + q) >> 1));
                gi2[r] = ol_v2{gi, gi};
                s += gi2[r] * g[r];
            }
            float t0, e0, t1, e1;
            ol_term<MODE>(s.x, __uint_as_float(a[q].x), t0, e0);
            ol_term<MODE>(s.y, __uint_as_float(a[q].y), t1, e1);     // (an odd N: the pair's second column of
            const ol_v2 e{ok0 ? e0 : 0.0f, ok1 ? e1 : 0.0f};         //  the last lane is discarded here)
            term += (ok0 ? t0 : 0.0f) + (ok1 ? t1 : 0.0f);
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] += e * gi2[r];
        }
    }
}
template <int MODE, int R>
__device__ __forceinline__ void ol_sym_sweep(const float* X, int N, int lane, const ol_v2 (&g)[R], ol_v2 (&v)[R],
                                             float& term) {
    // the descriptor must be provably wave-uniform: X depends on the wave's index
    const unsigned long addr = reinterpret_cast<unsigned long>(X);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
    void* base = reinterpret_cast<void*>(((unsigned long)hi << 32) | lo);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, N * N * 4, 0x00020000);
    const bool ok0 = 2 * lane < N, ok1 = 2 * lane + 1 < N;
    const int off = ok0 ? lane * 8 : 0x7ffffff0;
    const int groups = N / OL_ROWS, tail = N - groups * OL_ROWS;
    for (int gr = 0; gr < groups; ++gr)
        ol_sym_group<MODE, R, true>(rsrc, off, N * 4, ok0, ok1, gr * OL_ROWS, OL_ROWS, g, v, term);
    if (tail) ol_sym_group<MODE, R, false>(rsrc, off, N * 4, ok0, ok1, groups * OL_ROWS, tail, g, v, term);
}

// SYM: the caller vouches that every x[b] is symmetric (see ol_sym_sweep).
template <int MODE, int R, bool SYM>
__global__ __launch_bounds__(OL_NT) void outer_loss_kernel(const float* __restrict__ nodes, const float* __restrict__ x,
                                                           int B, int N, const float* __restrict__ mu,
                                                           const float* __restrict__ logvar, int L,
                                                           float* __restrict__ gnodes, float* __restrict__ gmu,
                                                           float* __restrict__ glogvar, double* __restrict__ per_sample) {
    __shared__ __attribute__((aligned(16))) float lds[SYM ? 4 : (OL_NT / 64) * ol_wave_floats(R)];
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (OL_NT / 64) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (OL_NT / 64);
    const int D = N * N;
    // columns of the lane's packed pair: (lane, lane + 64), or (2 lane, 2 lane + 1) in the SYM sweep
    const int col0 = SYM ? 2 * lane : lane, col1 = SYM ? 2 * lane + 1 : lane + 64;
    const bool c0 = col0 < N, c1 = col1 < N;
    float* tile = lds + (threadIdx.x >> 6) * ol_wave_floats(R);    // [R * OL_ROWS][OL_LD]
    float* utab = tile + R * OL_ROWS * OL_LD;                       // [R][OL_NP]
    for (int b = wave; b < B; b += nwaves) {
        const float* X = x + (long)b * D;
        const float* gv = nodes + (long)b * R * OL_NP;
        ol_v2 g[R], v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            g[r].x = c0 ? gv[r * OL_NP + col0] : 0.0f;
            g[r].y = c1 ? gv[r * OL_NP + col1] : 0.0f;
            v[r] = ol_v2{0.0f, 0.0f};
        }
        float term = 0.0f;
        if constexpr (SYM) ol_sym_sweep<MODE, R>(X, N, lane, g, v, term);
        for (int i0 = 0; !SYM && i0 < N; i0 += OL_ROWS) {
            ol_v2 a[OL_ROWS];
#pragma unroll
            for (int q = 0; q < OL_ROWS; ++q) {
                const bool live_row = i0 + q < N;
                a[q].x = (live_row && c0) ? __builtin_nontemporal_load(X + (i0 + q) * N + lane) : 0.0f;
                a[q].y = (live_row && c1) ? __builtin_nontemporal_load(X + (i0 + q) * N + lane + 64) : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < OL_ROWS; ++q) {
                const int i = i0 + q;                               // uniform over the wave
                if (i < N) {
                    float gi[R];
                    ol_v2 s{0.0f, 0.0f};
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        gi[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(i < 64 ? g[r].x : g[r].y), i & 63));
                        s += ol_v2{gi[r], gi[r]} * g[r];
                    }
                    // columns that do not exist contribute neither loss nor gradient
                    float t0, e0, t1, e1;
                    ol_term<MODE>(s.x, a[q].x, t0, e0);
                    ol_term<MODE>(s.y, a[q].y, t1, e1);
                    const ol_v2 e{c0 ? e0 : 0.0f, c1 ? e1 : 0.0f};
                    term += (c0 ? t0 : 0.0f) + (c1 ? t1 : 0.0f);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        v[r] += e * ol_v2{gi[r], gi[r]};
                        const ol_v2 pg = e * g[r];
                        tile[(r * OL_ROWS + q) * OL_LD + lane] = pg.x + pg.y;
                    }
                }
            }
            // LDS operations of one wave execute in program order; the fences keep the compiler from
            // moving the reads below ahead of the writes above (and the next group's writes ahead of
            // these reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < R * OL_ROWS) {
                const int r = lane / OL_ROWS, q = lane - r * OL_ROWS;
                if (i0 + q < N) {
                    const ol_v4* row = reinterpret_cast<const ol_v4*>(tile + lane * OL_LD);
                    ol_v4 acc4 = row[0];
#pragma unroll
                    for (int j = 1; j < 16; ++j) acc4 += row[j];
                    utab[r * OL_NP + i0 + q] = (acc4.x + acc4.y) + (acc4.z + acc4.w);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        term = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_dpp(term)), 63));
        float kl = 0.0f;
        for (int i = lane; i < L; i += 64) {
            const long j = (long)b * L + i;
            const float m = mu[j], lv = logvar[j];
            const float e = expf(lv);
            kl += -0.5f * (1.0f + lv - m * m - e);
            if (gmu) {
                gmu[j] = m;
                glogvar[j] = 0.5f * (e - 1.0f);
            }
        }
        kl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_dpp(kl)), 63));
        if (lane == 0) {
            per_sample[2 * b] = (double)term;
            per_sample[2 * b + 1] = (double)kl;
        }
        float* dst = gnodes + (long)b * R * OL_NP;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float u0 = SYM ? v[r].x : (c0 ? utab[r * OL_NP + lane] : 0.0f);
            const float u1 = SYM ? v[r].y : (c1 ? utab[r * OL_NP + lane + 64] : 0.0f);
            if (col0 < OL_NP) dst[r * OL_NP + col0] = c0 ? u0 + v[r].x : 0.0f;
            if (col1 < OL_NP) dst[r * OL_NP + col1] = c1 ? u1 + v[r].y : 0.0f;
        }
    }
}

// out[0] = recon term + KL, out[1] = recon term, out[2] = KL (fixed summation order)
__global__ __launch_bounds__(PAIR_NT) void outer_loss_finalize_kernel(const double* __restrict__ per_sample, int B,
                                                                  float* __restrict__ out) {
    __shared__ double red[PAIR_NT / 64];
    const PairSum t = block_pair_sum(per_sample, B, red);
    if (threadIdx.x == 0) {
        out[0] = (float)(t.a + t.k);
        out[1] = (float)t.a;
        out[2] = (float)t.k;
    }
}

template <int MODE, bool SYM>
static void launch_outer_loss(int R, unsigned grid, hipStream_t st, const float* nodes, const float* x, int B, int N,
                              const float* mu, const float* logvar, int L, float* gnodes, float* gmu, float* glv,
                              double* per_sample) {
#define MV_OL_CASE(RR)                                                                                              \
    case RR:                                                                                                        \
        hipLaunchKernelGGL((outer_loss_kernel<MODE, RR, SYM>), dim3(grid), dim3(OL_NT), 0, st, nodes, x, B, N, mu, logvar, \
                           L, gnodes, gmu, glv, per_sample);                                                        \
        break;
    switch (R) {
        MV_OL_CASE(1) MV_OL_CASE(2) MV_OL_CASE(3) MV_OL_CASE(4) MV_OL_CASE(5) MV_OL_CASE(6) MV_OL_CASE(7)
        default:
            hipLaunchKernelGGL((outer_loss_kernel<MODE, 8, SYM>), dim3(grid), dim3(OL_NT), 0, st, nodes, x, B, N, mu, logvar,
                               L, gnodes, gmu, glv, per_sample);
    }
#undef MV_OL_CASE
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

size_t mdnvae_outer_loss_workspace_bytes(int B) { return sizeof(double) * 2 * (size_t)(B > 0 ? B : 1); }

int mdnvae_outer_loss_fwdbwd(int mode, const float* nodes, const float* x, int B, int R, int N, const float* mu,
                             const float* logvar, int L, float* grad_nodes, float* grad_mu, float* grad_logvar,
                             int x_symmetric, void* workspace, float* loss_out, void* stream) {
    MV_REQUIRE(mode >= 0 && mode <= MDNVAE_RECON_MSE, E_ARG, "mdnvae_outer_loss_fwdbwd: unknown reconstruction mode %d", mode);
    MV_REQUIRE(B >= 0 && L >= 0, E_ARG, "mdnvae_outer_loss_fwdbwd: bad size");
    MV_REQUIRE(N >= 1 && N <= OL_NP - 1, E_SHAPE, "mdnvae_outer_loss_fwdbwd: N = %d outside 1..%d", N, OL_NP - 1);
    MV_REQUIRE(R >= 1 && R <= OL_MAXR, E_SHAPE, "mdnvae_outer_loss_fwdbwd: %d channels (1..%d supported)", R, OL_MAXR);
    MV_REQUIRE(workspace && loss_out, E_ARG, "mdnvae_outer_loss_fwdbwd: workspace/loss_out is null");
    hipStream_t st = (hipStream_t)stream;
    if (B > 0) {
        MV_REQUIRE(nodes && x && grad_nodes, E_ARG, "mdnvae_outer_loss_fwdbwd: null pointer");
        MV_REQUIRE(L == 0 || (mu && logvar), E_ARG, "mdnvae_outer_loss_fwdbwd: mu/logvar is null");
        MV_REQUIRE((grad_mu == nullptr) == (grad_logvar == nullptr), E_ARG,
                   "mdnvae_outer_loss_fwdbwd: grad_mu and grad_logvar come together");
        const int blocks = (B + OL_NT / 64 - 1) / (OL_NT / 64);
        const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
        double* ps = (double*)workspace;
#define MV_OL_MODE(MM)                                                                                          \
    if (x_symmetric)                                                                                            \
        launch_outer_loss<MM, true>(R, grid, st, nodes, x, B, N, mu, logvar, L, grad_nodes, grad_mu, grad_logvar, ps); \
    else                                                                                                        \
        launch_outer_loss<MM, false>(R, grid, st, nodes, x, B, N, mu, logvar, L, grad_nodes, grad_mu, grad_logvar, ps)
        if (mode == MDNVAE_RECON_POISSON_RATE) { MV_OL_MODE(MDNVAE_RECON_POISSON_RATE); }
        else if (mode == MDNVAE_RECON_POISSON_LOG) { MV_OL_MODE(MDNVAE_RECON_POISSON_LOG); }
        else { MV_OL_MODE(MDNVAE_RECON_MSE); }
#undef MV_OL_MODE
        MV_LAUNCH_CHECK("outer_loss_kernel");
    }
    hipLaunchKernelGGL(outer_loss_finalize_kernel, dim3(1), dim3(PAIR_NT), 0, st, (const double*)workspace, B, loss_out);
    MV_LAUNCH_CHECK("outer_loss_finalize_kernel");
    return 0;
}

}  // extern "C"
