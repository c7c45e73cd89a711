// Fused reparameterisation + KL + trait-gated Gaussian-mixture negative log-likelihood.
//
// BUILD-DEFINED: the reference holds no mixture-density code (its only traces of trait
// conditioning are the unused data/trait.pt and the fragment tools/extract.py:31-43), so the
// model below is this library's own specification:
//   component k of sample b has mean matrix m_bk (the k-th channel's g_k g_k^T of the GCN decoder,
//   function/MM_Gate.py:126-135 before the channels are summed), an isotropic scale sigma_k, and
//   a gate pi_bk = softmax_k(gate_logits_bk) computed from the subject's traits;
//       ll_bk = sum_d [ -0.5 ((x_bd - m_bkd)/sigma_k)^2 - log sigma_k - 0.5 log 2pi ]
//       nll   = - sum_b logsumexp_k ( log pi_bk + ll_bk )
//   loss = nll + KL(q(z|x) || N(0,I))   (KL as in function/MM_Gate.py:181).
//
// One wave streams one sample at a time: x and its K component rows are read once with 16-byte
// non-temporal loads, K running sums of squares per lane; the K sums are reduced by wave
// shuffles, lane k keeps component k, and the log-sum-exp over the K components is again taken
// by shuffles - no LDS and no workgroup barrier.  The same wave does the sample's latent part
// (z, KL).  Backward is a second streaming kernel:
// grad_m_bkd = g * r_bk (m_bkd - x_bd) / sigma_k^2 with the responsibilities r_bk saved by the
// forward.
#include "common.h"
#include "../../include/mdnvae_hip.h"
#include <mutex>

namespace mdnvae {

constexpr int MDN_NT = 256;              // four waves per workgroup, one sample per wave
constexpr int MDN_MAXK = 8;
constexpr float LOG_2PI = 1.8378770664093453f;

// Exponent of component k (the lane's own) of one sample, t_k = u_k / 2 + D log sigma_k with
// u_k = ss_k / sigma_k^2, taken relative to (about) the smallest t over the K owning lanes.  t runs
// to 1e4 and beyond (D = 7569), where one fp32 step is 1e-3: adding the O(1) log mixture weight to
// it directly rounds the weight away, and with it the responsibilities of near-tied components.
// Differences keep the precision: rel_k = (u_k - u_m) / 2 + D (log sigma_k - log sigma_m) with m
// the lane whose fp32 t is smallest; the part common to all components, t_m + (D/2) log 2 pi, joins
// the per-sample total in double as `offset` (nll_b = offset - logsumexp_k(log pi_k - rel_k)).
struct MixtureExponent { float rel; double offset; };
__device__ __forceinline__ MixtureExponent mixture_exponent(bool owner, float ss, float inv_sigma2, float log_sigma,
                                                            int D) {
    const float u = ss * inv_sigma2;
    const float t = owner ? 0.5f * u + (float)D * log_sigma : INFINITY;
    const float tmin = -wave_max_all(-t);
    const unsigned long long at_min = __ballot(owner && t == tmin);
    const int m = at_min ? __builtin_ctzll(at_min) : 0;
    const float um = __shfl(u, m), lsm = __shfl(log_sigma, m);
    return MixtureExponent{owner ? 0.5f * (u - um) + (float)D * (log_sigma - lsm) : 0.0f,
                           0.5 * (double)um + (double)D * ((double)lsm + 0.5 * (double)LOG_2PI)};
}

// Rows of D floats start at arbitrary 4-byte phases when D is not a multiple of 4 (87*87 = 7569);
// the hardware takes 16-byte vector accesses at any 4-byte aligned address, so rows are walked as
// quads of their own elements through this 4-byte aligned vector type.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef v4f v4fu __attribute__((aligned(4)));

#ifndef MDN_NT_LD
#define MDN_NT_LD 1
#endif
#ifndef MDN_NT_ST
#define MDN_NT_ST 1
#endif
__device__ __forceinline__ v4f quad_load(const float* row, int i) {
#if MDN_NT_LD
    return __builtin_nontemporal_load(reinterpret_cast<const v4fu*>(row) + i);
#else
    return *(reinterpret_cast<const v4fu*>(row) + i);
#endif
}
__device__ __forceinline__ void quad_store(float* row, int i, v4f v) {
#if MDN_NT_ST
    __builtin_nontemporal_store(v, reinterpret_cast<v4fu*>(row) + i);
#else
    *(reinterpret_cast<v4fu*>(row) + i) = v;
#endif
}

template <int K>
__global__ __launch_bounds__(MDN_NT) void mdn_fwd_kernel(const float* __restrict__ comp, const float* __restrict__ x,
                                                         const float* __restrict__ logits,
                                                         const float* __restrict__ log_sigma, int B, int D,
                                                         const float* __restrict__ mu, const float* __restrict__ logvar,
                                                         const float* __restrict__ eps, float* __restrict__ z, int L,
                                                         float* __restrict__ resp, float* __restrict__ ss_out,
                                                         double* __restrict__ per_sample) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (MDN_NT / 64) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (MDN_NT / 64);
    const int D4 = D >> 2;
    for (int b = wave; b < B; b += nwaves) {
        const float* xr = x + (long)b * D;
        const float* cr = comp + (long)b * K * D;
        float ss[K];
#pragma unroll
        for (int k = 0; k < K; ++k) ss[k] = 0.0f;
        for (int i = lane; i < D4; i += 64) {
            const v4f xv = quad_load(xr, i);
            v4f mv[K];
#pragma unroll
            for (int k = 0; k < K; ++k)
                mv[k] = quad_load(cr + (long)k * D, i);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const v4f d = xv - mv[k];
                ss[k] += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        for (int i = (D4 << 2) + lane; i < D; i += 64) {          // at most 3 elements
            const float xv = xr[i];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float d = xv - cr[(long)k * D + i];
                ss[k] += d * d;
            }
        }
        // lane k < K ends up owning component k
        float ssk = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float v = wave_sum_all(ss[k]);
            if (lane == k) ssk = v;
        }
        // latent part of this sample
        float kl = 0.0f;
        for (int i = lane; i < L; i += 64) {
            const long j = (long)b * L + i;
            const float m = mu[j], lv = logvar[j];
            kl += -0.5f * (1.0f + lv - m * m - expf(lv));
            if (eps && z) z[j] = m + eps[j] * expf(0.5f * lv);
        }
        kl = wave_sum_all(kl);
        // log-sum-exp across the K owning lanes by shuffles
        float a = -INFINITY, gl = -INFINITY;
        if (lane < K) gl = logits[(long)b * K + lane];
        const float gmax = wave_max_all(gl);
        const float gsum = wave_sum_all(lane < K ? expf(gl - gmax) : 0.0f);
        const float ls = lane < K ? log_sigma[lane] : 0.0f;
        const MixtureExponent me = mixture_exponent(lane < K, ssk, expf(-2.0f * ls), ls, D);
        if (lane < K) a = (gl - gmax - logf(gsum)) - me.rel;
        const float amax = wave_max_all(a);
        const float asum = wave_sum_all(lane < K ? expf(a - amax) : 0.0f);
        const float lse = amax + logf(asum);
        if (lane < K) {
            resp[(long)b * K + lane] = expf(a - lse);
            ss_out[(long)b * K + lane] = ssk;
        }
        if (lane == 0) {
            per_sample[2 * b] = me.offset - (double)lse;
            per_sample[2 * b + 1] = (double)kl;
        }
    }
}

// out[0] = nll + KL, out[1] = nll, out[2] = KL (fixed summation order)
__global__ __launch_bounds__(PAIR_NT) void mdn_finalize_kernel(const double* __restrict__ per_sample, int B,
                                                           float* __restrict__ out) {
    __shared__ double red[PAIR_NT / 64];
    const PairSum t = block_pair_sum(per_sample, B, red);
    if (threadIdx.x == 0) {
        out[0] = (float)(t.a + t.k);
        out[1] = (float)t.a;
        out[2] = (float)t.k;
    }
}

#ifndef MDN_UNROLL_N
#define MDN_UNROLL_N 2
#endif
constexpr int MDN_UNROLL = MDN_UNROLL_N;            // quads per lane per tile
#ifndef MDN_BWD_BLOCKS_N
#define MDN_BWD_BLOCKS_N 8192
#endif

// Backward, pass 1 (one thread per (sample, component)): coef = g r / sigma^2, the gate and scale
// gradients; plus the latent gradients.  gscale[0] scales the mixture term, gscale[1] the KL term
// (NULL = 1,1).
__global__ __launch_bounds__(256) void mdn_bwd_small_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ log_sigma,
                                                            const float* __restrict__ resp, const float* __restrict__ ss,
                                                            int B, int K, int D, const float* __restrict__ mu,
                                                            const float* __restrict__ logvar, int L,
                                                            const float* __restrict__ gscale, float* __restrict__ coef,
                                                            float* __restrict__ glogits, float* __restrict__ glsig_rows,
                                                            float* __restrict__ gmu, float* __restrict__ glogvar) {
    const float g = gscale ? gscale[0] : 1.0f, gk = gscale ? gscale[1] : 1.0f;
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, nth = (long)gridDim.x * 256;
    for (long i = tid; i < (long)B * K; i += nth) {
        const int b = (int)(i / K), k = (int)(i - (long)b * K);
        float gmax = -INFINITY;
        for (int j = 0; j < K; ++j) gmax = fmaxf(gmax, logits[(long)b * K + j]);
        float gsum = 0.0f;
        for (int j = 0; j < K; ++j) gsum += expf(logits[(long)b * K + j] - gmax);
        const float pi = expf(logits[i] - gmax) / gsum;
        const float r = resp[i], is2 = expf(-2.0f * log_sigma[k]);
        coef[i] = g * r * is2;
        glogits[i] = g * (pi - r);
        glsig_rows[i] = g * r * ((float)D - ss[i] * is2);
    }
    if (gmu) {
        for (long j = tid; j < (long)B * L; j += nth) {
            gmu[j] = gk * mu[j];
            glogvar[j] = gk * 0.5f * (expf(logvar[j]) - 1.0f);
        }
    }
}

// Backward, pass 2: grad_comp = coef[row] (comp - x) streamed as ONE flat array of B*K*D floats
// (the same sweep as the reconstruction-loss kernels: consecutive workgroups take consecutive
// 8 KB tiles, so the chip works on a compact window of addresses).  A quad usually lies inside one
// component row; the few that straddle two rows take the per-element path.  x rows are re-read
// once per component from cache (default policy), comp / grad_comp are non-temporal.
__global__ __launch_bounds__(MDN_NT) void mdn_bwd_stream_kernel(const float* __restrict__ comp,
                                                                const float* __restrict__ x,
                                                                const float* __restrict__ coef, unsigned total,
                                                                unsigned D, unsigned K, unsigned magicD,
                                                                unsigned magicK, float* __restrict__ gcomp) {
    // floor(n / d) for n < 2^32 given magic = ceil(2^32 / d) is exact only for small n, so the
    // quotient is corrected by one step either way (d >= 1, n < 2^32).
    auto divide = [](unsigned n, unsigned d, unsigned magic) {
        unsigned q = d == 1 ? n : __umulhi(n, magic);
        const long rem = (long)n - (long)q * d;
        if (rem < 0) --q;
        else if (rem >= (long)d) ++q;
        return q;
    };
    const unsigned nquads = total >> 2;
    constexpr unsigned TILE = MDN_NT * MDN_UNROLL;
    for (unsigned base = blockIdx.x * TILE; base < nquads; base += gridDim.x * TILE) {
        v4f mv[MDN_UNROLL], xv[MDN_UNROLL];
        float cf[MDN_UNROLL];
        bool whole[MDN_UNROLL];
#pragma unroll
        for (int u = 0; u < MDN_UNROLL; ++u) {
            const unsigned q = base + u * MDN_NT + threadIdx.x;
            whole[u] = false;
            if (q < nquads) {
                const unsigned f = q << 2;
                const unsigned row = divide(f, D, magicD);
                const unsigned d = f - row * D;
                whole[u] = d + 3 < D;
                if (whole[u]) {
                    const unsigned b = divide(row, K, magicK);
                    mv[u] = __builtin_nontemporal_load(reinterpret_cast<const v4fu*>(comp) + q);
                    xv[u] = *reinterpret_cast<const v4fu*>(x + (size_t)b * D + d);
                    cf[u] = coef[row];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < MDN_UNROLL; ++u) {
            const unsigned q = base + u * MDN_NT + threadIdx.x;
            if (q < nquads) {
                if (whole[u]) {
                    const v4f o = cf[u] * (mv[u] - xv[u]);
                    __builtin_nontemporal_store(o, reinterpret_cast<v4fu*>(gcomp) + q);
                } else {
                    for (unsigned c = 0; c < 4; ++c) {
                        const unsigned f = (q << 2) + c;
                        const unsigned row = divide(f, D, magicD);
                        const unsigned d = f - row * D;
                        const unsigned b = divide(row, K, magicK);
                        gcomp[f] = coef[row] * (comp[f] - x[(size_t)b * D + d]);
                    }
                }
            }
        }
    }
    // the (at most 3) elements past the last whole quad
    const unsigned f = (nquads << 2) + blockIdx.x * MDN_NT + threadIdx.x;
    if (f < total) {
        const unsigned row = divide(f, D, magicD);
        const unsigned d = f - row * D;
        const unsigned b = divide(row, K, magicK);
        gcomp[f] = coef[row] * (comp[f] - x[(size_t)b * D + d]);
    }
}

// Rank-one components: m_bk = g_bk g_bk^T from the node vectors of the GCN decoder.  One wave per
// sample reads x[b] once, MO_ROWS rows in flight, lane c owning columns c and c+64 as a packed
// pair; for every component the residual r_ij = g_i g_j - x_ij is formed in registers and feeds,
// in the same sweep,
//   ss_k  = sum_ij r_ij^2                      (the exponent of the likelihood),
//   v_k[c] = sum_i r_ic g_i                    (column sums, per lane),
//   u_k[i] = sum_j r_ij g_j                    (row sums, across the lanes),
// so that after the log-sum-exp over the K components the gradient of the node vectors is
// coef_bk (u_k + v_k) with coef_bk = resp_bk / sigma_k^2: the [B,K,N*N] component matrices, their
// gradient and the passes over them do not exist.  The same wave does the sample's KL part.
// Row sums: the lanes' partial products of the K * MO_ROWS (component, row) pairs of a row group go
// to the wave's LDS tile [pair][lane]; lane p then adds up row p of the tile with 16-byte reads
// and leaves the total in the wave's u table, which the lanes read back per column at the end -
// about a quarter of the instructions of one cross-lane reduction per pair.
constexpr int MO_ROWS = 8;
constexpr int MO_NP = MDNVAE_DEC_NP;
constexpr int MO_LD = 68;                       // tile row stride: 16-byte rows, lanes 0..7 cover all banks
typedef float mo_v2 __attribute__((ext_vector_type(2)));
__host__ __device__ constexpr int mo_wave_floats(int K) { return K * MO_ROWS * MO_LD + K * MO_NP; }
// ---- the sweep over a symmetric x[b] ----
// Lane l owns the adjacent columns 2l and 2l+1 (one 8-byte load per row).  Loads go through a
// buffer descriptor of the sample's N*N floats: a lane whose columns do not exist carries an offset
// past the end, for which the hardware returns 0, so the row loads need no branches; the row offset
// is a scalar.  (Measured on MI355X at B = 16384, N = 87: issuing the next group's loads ahead of
// the arithmetic, 4 or 16 rows per group, unpacked FMAs and other cache policies all land within a
// few percent of this form or below it; the loads alone take 0.9 of the kernel's time, and the
// same bytes fetched as one flat run of 16-byte pieces per wave take 0.8 of it.)
typedef unsigned mo_u2 __attribute__((ext_vector_type(2)));
constexpr int MO_SR = 8;                        // rows per group (even: a row's parity is its slot's)
template <bool FULL>
__device__ __forceinline__ void mo_sym_load(__amdgpu_buffer_rsrc_t rsrc, int off, int row_bytes, int i0, int rows,
                                            mo_u2 (&a)[MO_SR]) {
#pragma unroll
    for (int q = 0; q < MO_SR; ++q) {
        if (FULL || q < rows) a[q] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, (i0 + q) * row_bytes, 2 /* nt */);
        else a[q] = mo_u2{0u, 0u};
    }
}
// rows i0 .. i0+rows-1 (i0 a multiple of MO_SR, so the parity of a row is known at compile time)
// against every component; the K components of a row are independent chains, issued side by side
template <int K, bool FULL>
__device__ __forceinline__ void mo_sym_rows(const mo_u2 (&a)[MO_SR], bool second_ok, int i0, int rows,
                                            const mo_v2 (&g)[K], mo_v2 (&v)[K], mo_v2 (&ss2)[K]) {
#pragma unroll
    for (int q = 0; q < MO_SR; ++q) {
        if (FULL || q < rows) {
            // the second column of the last lane of an odd N is the next row's first element
            const mo_v2 xq{__uint_as_float(a[q].x), second_ok ? __uint_as_float(a[q].y) : 0.0f};
            mo_v2 gi2[K], r[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float gi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int((q & 1) ? g[k].y : g[k].x),
                                                                         (i0 + q) >> 1));
                gi2[k] = mo_v2{gi, gi};
            }
#pragma unroll
            for (int k = 0; k < K; ++k) r[k] = gi2[k] * g[k] - xq;
#pragma unroll
            for (int k = 0; k < K; ++k) ss2[k] += r[k] * r[k];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] += r[k] * gi2[k];
        }
    }
}
template <int K>
__device__ __forceinline__ void mo_sym_sweep(const float* X, int N, int lane, const mo_v2 (&g)[K], mo_v2 (&v)[K],
                                             mo_v2 (&ss2)[K]) {
    // the descriptor must be provably wave-uniform: X depends on the wave's index
    const unsigned long addr = reinterpret_cast<unsigned long>(X);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)addr);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
    void* base = reinterpret_cast<void*>(((unsigned long)hi << 32) | lo);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, N * N * 4, 0x00020000);
    const int off = 2 * lane < N ? lane * 8 : 0x7ffffff0;
    const bool second_ok = 2 * lane + 1 < N;
    const int row_bytes = N * 4;
    const int groups = N / MO_SR, tail = N - groups * MO_SR;
    mo_u2 a[MO_SR];
    for (int gr = 0; gr < groups; ++gr) {
        mo_sym_load<true>(rsrc, off, row_bytes, gr * MO_SR, MO_SR, a);
        mo_sym_rows<K, true>(a, second_ok, gr * MO_SR, MO_SR, g, v, ss2);
    }
    if (tail) {
        mo_sym_load<false>(rsrc, off, row_bytes, groups * MO_SR, tail, a);
        mo_sym_rows<K, false>(a, second_ok, groups * MO_SR, tail, g, v, ss2);
    }
}

// SYM: the caller vouches that every x[b] is symmetric.  Then r is symmetric, the row sums equal
// the column sums the lanes already hold, and the sweep needs no LDS and no cross-lane step at all:
// three packed operations per (row, component) on a stream of x.
template <int K, bool SYM>
__global__ __launch_bounds__(MDN_NT) void mdn_outer_kernel(const float* __restrict__ nodes, const float* __restrict__ x,
                                                           const float* __restrict__ logits,
                                                           const float* __restrict__ log_sigma, int B, int N,
                                                           const float* __restrict__ mu, const float* __restrict__ logvar,
                                                           int L, float* __restrict__ gnodes, float* __restrict__ glogits,
                                                           float* __restrict__ glsig_rows, float* __restrict__ gmu,
                                                           float* __restrict__ glogvar, float* __restrict__ resp,
                                                           double* __restrict__ per_sample) {
    __shared__ __attribute__((aligned(16))) float lds[SYM ? 4 : (MDN_NT / 64) * mo_wave_floats(K)];
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (MDN_NT / 64) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (MDN_NT / 64);
    const int D = N * N;
    // columns of the lane's packed pair: (lane, lane + 64), or (2 lane, 2 lane + 1) in the SYM sweep
    const int col0 = SYM ? 2 * lane : lane, col1 = SYM ? 2 * lane + 1 : lane + 64;
    const bool c0 = col0 < N, c1 = col1 < N;
    float* tile = lds + (threadIdx.x >> 6) * mo_wave_floats(K);    // [K * MO_ROWS][MO_LD]
    float* utab = tile + K * MO_ROWS * MO_LD;                       // [K][MO_NP]
    for (int b = wave; b < B; b += nwaves) {
        const float* X = x + (long)b * D;
        const float* gv = nodes + (long)b * K * MO_NP;
        // the lane's two columns (c, c+64) travel as one packed pair: the residual, the sum of
        // squares and the column sums are packed fp32 operations
        mo_v2 g[K], v[K], ss2[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            g[k].x = c0 ? gv[k * MO_NP + col0] : 0.0f;
            g[k].y = c1 ? gv[k * MO_NP + col1] : 0.0f;
            v[k] = ss2[k] = mo_v2{0.0f, 0.0f};
        }
        if constexpr (SYM) mo_sym_sweep<K>(X, N, lane, g, v, ss2);
        for (int i0 = 0; !SYM && i0 < N; i0 += MO_ROWS) {
            mo_v2 a[MO_ROWS];
#pragma unroll
            for (int q = 0; q < MO_ROWS; ++q) {
                const bool live_row = i0 + q < N;
                a[q].x = (live_row && c0) ? __builtin_nontemporal_load(X + (i0 + q) * N + lane) : 0.0f;
                a[q].y = (live_row && c1) ? __builtin_nontemporal_load(X + (i0 + q) * N + lane + 64) : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < MO_ROWS; ++q) {
                const int i = i0 + q;                               // uniform over the wave
                if (i < N) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const float gi = __int_as_float(__builtin_amdgcn_readlane(
                            __float_as_int(i < 64 ? g[k].x : g[k].y), i & 63));
                        const mo_v2 gi2{gi, gi};
                        // dead lanes hold g = 0 and x = 0, so their residual is 0 without a select
                        const mo_v2 r = gi2 * g[k] - a[q];
                        ss2[k] += r * r;
                        v[k] += r * gi2;
                        const mo_v2 pg = r * g[k];
                        tile[(k * MO_ROWS + q) * MO_LD + lane] = pg.x + pg.y;
                    }
                }
            }
            // LDS operations of one wave execute in program order; the fence keeps the compiler
            // from moving the reads below ahead of the writes above (and the next group's writes
            // ahead of these reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < K * MO_ROWS) {
                const int k = lane / MO_ROWS, q = lane - k * MO_ROWS;
                if (i0 + q < N) {
                    const v4f* row = reinterpret_cast<const v4f*>(tile + lane * MO_LD);
                    v4f acc4 = row[0];
#pragma unroll
                    for (int j = 1; j < 16; ++j) acc4 += row[j];
                    utab[k * MO_NP + i0 + q] = (acc4.x + acc4.y) + (acc4.z + acc4.w);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        float ss[K], v0[K], v1[K], u0[K], u1[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            ss[k] = ss2[k].x + ss2[k].y;
            v0[k] = v[k].x;
            v1[k] = v[k].y;
            u0[k] = SYM ? v0[k] : (c0 ? utab[k * MO_NP + lane] : 0.0f);
            u1[k] = SYM ? v1[k] : (c1 ? utab[k * MO_NP + lane + 64] : 0.0f);
        }
        float ssk = 0.0f;                                           // lane k < K owns component k
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float v = wave_sum_all(ss[k]);
            if (lane == k) ssk = v;
        }
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
        kl = wave_sum_all(kl);
        float a = -INFINITY, gl = -INFINITY, is2 = 0.0f, pi = 0.0f;
        if (lane < K) gl = logits[(long)b * K + lane];
        const float gmax = wave_max_all(gl);
        const float gsum = wave_sum_all(lane < K ? expf(gl - gmax) : 0.0f);
        const float ls = lane < K ? log_sigma[lane] : 0.0f;
        if (lane < K) is2 = expf(-2.0f * ls);
        const MixtureExponent me = mixture_exponent(lane < K, ssk, is2, ls, D);
        if (lane < K) {
            const float logpi = gl - gmax - logf(gsum);
            pi = expf(logpi);
            a = logpi - me.rel;
        }
        const float amax = wave_max_all(a);
        const float asum = wave_sum_all(lane < K ? expf(a - amax) : 0.0f);
        const float lse = amax + logf(asum);
        float coef = 0.0f;
        if (lane < K) {
            const float r = expf(a - lse);
            coef = r * is2;
            resp[(long)b * K + lane] = r;
            glogits[(long)b * K + lane] = pi - r;
            glsig_rows[(long)b * K + lane] = r * ((float)D - ssk * is2);
        }
        if (lane == 0) {
            per_sample[2 * b] = me.offset - (double)lse;
            per_sample[2 * b + 1] = (double)kl;
        }
        float* dst = gnodes + (long)b * K * MO_NP;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float ck = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(coef), k));
            if (col0 < MO_NP) dst[k * MO_NP + col0] = c0 ? ck * (u0[k] + v0[k]) : 0.0f;
            if (col1 < MO_NP) dst[k * MO_NP + col1] = c1 ? ck * (u1[k] + v1[k]) : 0.0f;
        }
    }
}

}  // namespace mdnvae
using namespace mdnvae;

#define MV_MDN_DISPATCH(K, CALL) \
    switch (K) {                 \
        case 1: CALL(1); break;  \
        case 2: CALL(2); break;  \
        case 3: CALL(3); break;  \
        case 4: CALL(4); break;  \
        case 5: CALL(5); break;  \
        case 6: CALL(6); break;  \
        case 7: CALL(7); break;  \
        default: CALL(8); break; \
    }

#ifndef MDN_MAX_BLOCKS_N
#define MDN_MAX_BLOCKS_N 1024
#endif
// one wave per sample, every wave resident at once where the batch allows it (a grid sized to the
// kernel's measured occupancy, equal samples per wave, was tried and is no faster)
static unsigned mdn_outer_grid(int B) {
    const int blocks = (B + MDN_NT / 64 - 1) / (MDN_NT / 64);
    return (unsigned)(blocks < 2048 ? blocks : 2048);
}
static unsigned mdn_grid(int B) {
    const int blocks = (B + MDN_NT / 64 - 1) / (MDN_NT / 64);
    return (unsigned)(blocks < MDN_MAX_BLOCKS_N ? blocks : MDN_MAX_BLOCKS_N);
}

// ---- static scratch of the entry point that allocates nothing ----
// Two doubles per sample (likelihood term, KL term) for at most MDN_STATIC_MAX_B samples: 1 MiB that
// every GPU the library runs on holds until the process ends.  In host code the array's name is
// only a stand-in; the device address comes from hipGetSymbolAddress, once per device.
constexpr int MDN_STATIC_MAX_B = 65536;
static __device__ __align__(256) double scratch[2 * MDN_STATIC_MAX_B];

constexpr int MDN_MAX_DEVICES = 64;
static std::mutex g_scratch_mutex;
static double* g_scratch_ptr[MDN_MAX_DEVICES] = {};

// Device address of `scratch` on the current device (looked up on the first call there).
static int static_scratch(const char* fn, double** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        ::mdnvae::set_error("%s: hipGetDevice failed: %s", fn, hipGetErrorString(e));
        return (int)e;
    }
    MV_REQUIRE(dev >= 0 && dev < MDN_MAX_DEVICES, E_ARG, "%s: device ordinal %d outside 0..%d", fn, dev,
               MDN_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            ::mdnvae::set_error("%s: hipGetSymbolAddress(scratch) failed on device %d: %s", fn, dev,
                                hipGetErrorString(e));
            return e != hipSuccess ? (int)e : E_ARG;
        }
        g_scratch_ptr[dev] = static_cast<double*>(p);
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" {

// forward: 2 doubles per sample; backward: one float per (sample, component)
size_t mdnvae_mdn_workspace_bytes(int B) { return sizeof(float) * MDN_MAXK * (size_t)(B > 0 ? B : 1); }

int mdnvae_mdn_loss_fwd(const float* comp_mean, const float* x, const float* gate_logits, const float* log_sigma,
                        int B, int K, int D, const float* mu, const float* logvar, int L, const float* eps, float* z,
                        float* resp, float* sumsq, void* workspace, float* loss_out, void* stream) {
    MV_REQUIRE(B >= 0 && D >= 1 && L >= 0, E_ARG, "mdnvae_mdn_loss_fwd: bad size");
    MV_REQUIRE(K >= 1 && K <= MDN_MAXK, E_SHAPE, "mdnvae_mdn_loss_fwd: %d components (1..%d supported)", K, MDN_MAXK);
    MV_REQUIRE(workspace && loss_out, E_ARG, "mdnvae_mdn_loss_fwd: workspace/loss_out is null");
    hipStream_t st = (hipStream_t)stream;
    if (B > 0) {
        MV_REQUIRE(comp_mean && x && gate_logits && log_sigma && resp && sumsq, E_ARG,
                   "mdnvae_mdn_loss_fwd: null pointer");
        MV_REQUIRE(L == 0 || (mu && logvar), E_ARG, "mdnvae_mdn_loss_fwd: mu/logvar is null");
        MV_REQUIRE(!z || eps, E_ARG, "mdnvae_mdn_loss_fwd: z requested without eps");
#define CALL(KK)                                                                                            \
    hipLaunchKernelGGL((mdn_fwd_kernel<KK>), dim3(mdn_grid(B)), dim3(MDN_NT), 0, st, comp_mean, x, \
                       gate_logits, log_sigma, B, D, mu, logvar, eps, z, L, resp, sumsq, (double*)workspace)
        MV_MDN_DISPATCH(K, CALL)
#undef CALL
        MV_LAUNCH_CHECK("mdn_fwd_kernel");
    }
    hipLaunchKernelGGL(mdn_finalize_kernel, dim3(1), dim3(PAIR_NT), 0, st, (const double*)workspace, B, loss_out);
    MV_LAUNCH_CHECK("mdn_finalize_kernel");
    return 0;
}

// Shared body of the two rank-one entry points: `per_sample` (2 doubles per sample) is either the
// caller's workspace or this device's static scratch array.
static int mdn_outer_launch(const char* fn, const float* nodes, const float* x, const float* gate_logits,
                            const float* log_sigma, int B, int K, int N, const float* mu, const float* logvar, int L,
                            float* grad_nodes, float* grad_logits, float* grad_log_sigma_rows, float* grad_mu,
                            float* grad_logvar, float* resp, int x_symmetric, double* per_sample, float* loss_out,
                            hipStream_t st) {
    MV_REQUIRE(B >= 0 && L >= 0, E_ARG, "%s: bad size", fn);
    MV_REQUIRE(N >= 1 && N <= MO_NP - 1, E_SHAPE, "%s: N = %d outside 1..%d", fn, N, MO_NP - 1);
    MV_REQUIRE(K >= 1 && K <= MDN_MAXK, E_SHAPE, "%s: %d components (1..%d supported)", fn, K, MDN_MAXK);
    MV_REQUIRE(per_sample && loss_out, E_ARG, "%s: workspace/loss_out is null", fn);
    if (B > 0) {
        MV_REQUIRE(nodes && x && gate_logits && log_sigma && grad_nodes && grad_logits && grad_log_sigma_rows && resp,
                   E_ARG, "%s: null pointer", fn);
        MV_REQUIRE(L == 0 || (mu && logvar), E_ARG, "%s: mu/logvar is null", fn);
        MV_REQUIRE((grad_mu == nullptr) == (grad_logvar == nullptr), E_ARG, "%s: grad_mu and grad_logvar come together",
                   fn);
#define CALL(KK)                                                                                                     \
    if (x_symmetric)                                                                                                 \
        hipLaunchKernelGGL((mdn_outer_kernel<KK, true>),                                                             \
                           dim3(mdn_outer_grid(B)),                                                                  \
                           dim3(MDN_NT), 0, st, nodes, x, gate_logits, log_sigma, B, N, mu, logvar, L, grad_nodes,   \
                           grad_logits, grad_log_sigma_rows, grad_mu, grad_logvar, resp, per_sample);                \
    else                                                                                                             \
        hipLaunchKernelGGL((mdn_outer_kernel<KK, false>),                                                            \
                           dim3(mdn_outer_grid(B)),                                                                  \
                           dim3(MDN_NT), 0, st, nodes, x, gate_logits, log_sigma, B, N, mu, logvar, L, grad_nodes,   \
                           grad_logits, grad_log_sigma_rows, grad_mu, grad_logvar, resp, per_sample)
        MV_MDN_DISPATCH(K, CALL)
#undef CALL
        MV_LAUNCH_CHECK("mdn_outer_kernel");
    }
    hipLaunchKernelGGL(mdn_finalize_kernel, dim3(1), dim3(PAIR_NT), 0, st, (const double*)per_sample, B, loss_out);
    MV_LAUNCH_CHECK("mdn_finalize_kernel");
    return 0;
}

int mdnvae_mdn_outer_fwdbwd(const float* nodes, const float* x, const float* gate_logits, const float* log_sigma,
                            int B, int K, int N, const float* mu, const float* logvar, int L, float* grad_nodes,
                            float* grad_logits, float* grad_log_sigma_rows, float* grad_mu, float* grad_logvar,
                            float* resp, int x_symmetric, void* workspace, float* loss_out, void* stream) {
    return mdn_outer_launch("mdnvae_mdn_outer_fwdbwd", nodes, x, gate_logits, log_sigma, B, K, N, mu, logvar, L,
                            grad_nodes, grad_logits, grad_log_sigma_rows, grad_mu, grad_logvar, resp, x_symmetric,
                            (double*)workspace, loss_out, (hipStream_t)stream);
}

int mdnvae_mdn_outer_fwdbwd_static(const float* nodes, const float* x, const float* gate_logits,
                                   const float* log_sigma, int B, int K, int N, const float* mu, const float* logvar,
                                   int L, float* grad_nodes, float* grad_logits, float* grad_log_sigma_rows,
                                   float* grad_mu, float* grad_logvar, float* resp, int x_symmetric, float* loss_out,
                                   void* stream) {
    MV_REQUIRE(B <= MDN_STATIC_MAX_B, E_SHAPE,
               "mdnvae_mdn_outer_fwdbwd_static: B = %d exceeds the static scratch (at most %d samples per call); "
               "use mdnvae_mdn_outer_fwdbwd with a workspace",
               B, MDN_STATIC_MAX_B);
    double* per_sample = nullptr;
    if (int rc = static_scratch("mdnvae_mdn_outer_fwdbwd_static", &per_sample)) return rc;
    // every sample's pair is written by the sweep before the finalize launch (same stream) reads it,
    // so nothing of the previous call's contents has to be cleared
    return mdn_outer_launch("mdnvae_mdn_outer_fwdbwd_static", nodes, x, gate_logits, log_sigma, B, K, N, mu, logvar,
                            L, grad_nodes, grad_logits, grad_log_sigma_rows, grad_mu, grad_logvar, resp, x_symmetric,
                            per_sample, loss_out, (hipStream_t)stream);
}

int mdnvae_mdn_loss_bwd(const float* comp_mean, const float* x, const float* gate_logits, const float* log_sigma,
                        const float* resp, const float* sumsq, int B, int K, int D, const float* mu,
                        const float* logvar, int L, const float* grad_loss, float* grad_comp, float* grad_logits,
                        float* grad_log_sigma_rows, float* grad_mu, float* grad_logvar, void* workspace,
                        void* stream) {
    MV_REQUIRE(B >= 0 && D >= 1 && L >= 0, E_ARG, "mdnvae_mdn_loss_bwd: bad size");
    MV_REQUIRE(K >= 1 && K <= MDN_MAXK, E_SHAPE, "mdnvae_mdn_loss_bwd: %d components (1..%d supported)", K, MDN_MAXK);
    if (B == 0) return 0;
    MV_REQUIRE((double)B * K * D < 4294967296.0, E_SHAPE,
               "mdnvae_mdn_loss_bwd: B*K*D must stay below 2^32 elements per call");
    MV_REQUIRE(comp_mean && x && gate_logits && log_sigma && resp && sumsq && grad_comp && grad_logits &&
                   grad_log_sigma_rows && workspace, E_ARG, "mdnvae_mdn_loss_bwd: null pointer");
    MV_REQUIRE((grad_mu == nullptr) == (grad_logvar == nullptr), E_ARG, "mdnvae_mdn_loss_bwd: grad_mu/grad_logvar");
    MV_REQUIRE(!grad_mu || (mu && logvar), E_ARG, "mdnvae_mdn_loss_bwd: mu/logvar is null");
    hipStream_t st = (hipStream_t)stream;
    float* coef = (float*)workspace;                       // B*K floats
    const long small = (long)B * (K > L ? K : L);
    hipLaunchKernelGGL(mdn_bwd_small_kernel, dim3((unsigned)((small + 255) / 256 > 1024 ? 1024 : (small + 255) / 256)),
                       dim3(256), 0, st, gate_logits, log_sigma, resp, sumsq, B, K, D, mu, logvar, L, grad_loss, coef,
                       grad_logits, grad_log_sigma_rows, grad_mu, grad_logvar);
    MV_LAUNCH_CHECK("mdn_bwd_small_kernel");
    const unsigned total = (unsigned)((long)B * K * D);
    const unsigned tiles = ((total >> 2) + MDN_NT * MDN_UNROLL - 1) / (MDN_NT * MDN_UNROLL);
    const unsigned grid = tiles < 1 ? 1 : (tiles > MDN_BWD_BLOCKS_N ? MDN_BWD_BLOCKS_N : tiles);
    const unsigned magicD = (unsigned)((4294967296ull + D - 1) / D), magicK = (unsigned)((4294967296ull + K - 1) / K);
    hipLaunchKernelGGL(mdn_bwd_stream_kernel, dim3(grid), dim3(MDN_NT), 0, st, comp_mean, x, coef, total, (unsigned)D,
                       (unsigned)K, magicD, magicK, grad_comp);
    MV_LAUNCH_CHECK("mdn_bwd_stream_kernel");
    return 0;
}

}  // extern "C"
