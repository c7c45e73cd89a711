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

template <int MODE, int R>
__global__ __launch_bounds__(OL_NT) void outer_loss_kernel(const float* __restrict__ nodes, const float* __restrict__ x,
                                                           int B, int N, const float* __restrict__ mu,
                                                           const float* __restrict__ logvar, int L,
                                                           float* __restrict__ gnodes, float* __restrict__ gmu,
                                                           float* __restrict__ glogvar, double* __restrict__ per_sample) {
    __shared__ __attribute__((aligned(16))) float lds[(OL_NT / 64) * ol_wave_floats(R)];
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (OL_NT / 64) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (OL_NT / 64);
    const int D = N * N;
    const bool c0 = lane < N, c1 = lane + 64 < N;
    float* tile = lds + (threadIdx.x >> 6) * ol_wave_floats(R);    // [R * OL_ROWS][OL_LD]
    float* utab = tile + R * OL_ROWS * OL_LD;                       // [R][OL_NP]
    for (int b = wave; b < B; b += nwaves) {
        const float* X = x + (long)b * D;
        const float* gv = nodes + (long)b * R * OL_NP;
        ol_v2 g[R], v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            g[r].x = c0 ? gv[r * OL_NP + lane] : 0.0f;
            g[r].y = c1 ? gv[r * OL_NP + lane + 64] : 0.0f;
            v[r] = ol_v2{0.0f, 0.0f};
        }
        float term = 0.0f;
        for (int i0 = 0; i0 < N; i0 += OL_ROWS) {
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
            const float u0 = c0 ? utab[r * OL_NP + lane] : 0.0f;
            const float u1 = c1 ? utab[r * OL_NP + lane + 64] : 0.0f;
            dst[r * OL_NP + lane] = c0 ? u0 + v[r].x : 0.0f;
            if (lane + 64 < OL_NP) dst[r * OL_NP + lane + 64] = c1 ? u1 + v[r].y : 0.0f;
        }
    }
}

// out[0] = recon term + KL, out[1] = recon term, out[2] = KL (fixed summation order)
__global__ __launch_bounds__(256) void outer_loss_finalize_kernel(const double* __restrict__ per_sample, int B,
                                                                  float* __restrict__ out) {
    __shared__ double red[4];
    double a = 0.0, k = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) {
        a += per_sample[2 * i];
        k += per_sample[2 * i + 1];
    }
    a = block_sum<256, double>(a, red);
    k = block_sum<256, double>(k, red);
    if (threadIdx.x == 0) {
        out[0] = (float)(a + k);
        out[1] = (float)a;
        out[2] = (float)k;
    }
}

template <int MODE>
static void launch_outer_loss(int R, unsigned grid, hipStream_t st, const float* nodes, const float* x, int B, int N,
                              const float* mu, const float* logvar, int L, float* gnodes, float* gmu, float* glv,
                              double* per_sample) {
#define MV_OL_CASE(RR)                                                                                              \
    case RR:                                                                                                        \
        hipLaunchKernelGGL((outer_loss_kernel<MODE, RR>), dim3(grid), dim3(OL_NT), 0, st, nodes, x, B, N, mu, logvar, L, \
                           gnodes, gmu, glv, per_sample);                                                           \
        break;
    switch (R) {
        MV_OL_CASE(1) MV_OL_CASE(2) MV_OL_CASE(3) MV_OL_CASE(4) MV_OL_CASE(5) MV_OL_CASE(6) MV_OL_CASE(7)
        default:
            hipLaunchKernelGGL((outer_loss_kernel<MODE, 8>), dim3(grid), dim3(OL_NT), 0, st, nodes, x, B, N, mu, logvar, L,
                               gnodes, gmu, glv, per_sample);
    }
#undef MV_OL_CASE
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

size_t mdnvae_outer_loss_workspace_bytes(int B) { return sizeof(double) * 2 * (size_t)(B > 0 ? B : 1); }

int mdnvae_outer_loss_fwdbwd(int mode, const float* nodes, const float* x, int B, int R, int N, const float* mu,
                             const float* logvar, int L, float* grad_nodes, float* grad_mu, float* grad_logvar,
                             void* workspace, float* loss_out, void* stream) {
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
        if (mode == MDNVAE_RECON_POISSON_RATE)
            launch_outer_loss<MDNVAE_RECON_POISSON_RATE>(R, grid, st, nodes, x, B, N, mu, logvar, L, grad_nodes, grad_mu,
                                                         grad_logvar, ps);
        else if (mode == MDNVAE_RECON_POISSON_LOG)
            launch_outer_loss<MDNVAE_RECON_POISSON_LOG>(R, grid, st, nodes, x, B, N, mu, logvar, L, grad_nodes, grad_mu,
                                                        grad_logvar, ps);
        else
            launch_outer_loss<MDNVAE_RECON_MSE>(R, grid, st, nodes, x, B, N, mu, logvar, L, grad_nodes, grad_mu,
                                                grad_logvar, ps);
        MV_LAUNCH_CHECK("outer_loss_kernel");
    }
    hipLaunchKernelGGL(outer_loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, B, loss_out);
    MV_LAUNCH_CHECK("outer_loss_finalize_kernel");
    return 0;
}

}  // extern "C"
