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
#include <type_traits>

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
#define MV_LDS_PTR_(p) ((__attribute__((address_space(3))) void*)(p))
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

// out[0] = nll + KL, out[1] = nll, out[2] = KL (fixed summation order).  K more workgroups, when rows is given,
// add up the K columns of rows [B][K] (the per-sample log-sigma gradients) into colsum [K]: thread t takes
// samples t, t + PAIR_NT, ... (four in flight), the threads' sums meet in a fixed order.
__global__ __launch_bounds__(PAIR_NT) void mdn_finalize_kernel(const double* __restrict__ per_sample, int B,
                                                           float* __restrict__ out, const float* __restrict__ rows, int K,
                                                           float* __restrict__ colsum) {
    __shared__ double red[PAIR_NT / 64];
    if (blockIdx.x >= 1) {
        // workgroup 1 + k adds up column k (one workgroup for all K columns pulled the whole [B][K] array through
        // one CU with 20-byte strides: 17 us at B = 16384)
        __shared__ float redf[PAIR_NT / 64];
        const int k = blockIdx.x - 1;
        float acc = 0.0f;
        int b = threadIdx.x;
        for (; b + 3 * PAIR_NT < B; b += 4 * PAIR_NT) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = rows[(long)(b + u * PAIR_NT) * K + k];
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (; b < B; b += PAIR_NT) acc += rows[(long)b * K + k];
        const float t = block_sum<PAIR_NT, float>(acc, redf);
        if (threadIdx.x == 0) colsum[k] = t;
        return;
    }
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
// once per component from cache (default policy), comp / grad_comp are non-temporal.  Two forms that
// fetch x once were measured slower (4.3 against 5.2 TB/s at B = 16384, N = 87, K = 5): tiles of one
// sample's rows with all K component quads in flight behind one x quad, and this sweep dealt to the XCDs
// by sample so that a sample's K rows pass through one L2 - the compact single stream matters more than
// the re-reads, which the fabric counters show (1.27x the algorithmic bytes) but the memory-side cache serves.
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


// ---- symmetric targets, second form: the sample's matrix streamed through wave-private LDS ----
// For symmetric x and m = g g^T everything the likelihood and its gradient need follows from the
// K matrix-vector products y_k = x g_k and the sum of squares of x:
//   ss_k        = sum_ij (g_i g_j - x_ij)^2 = |g_k|^4 - 2 g_k.y_k + sum_ij x_ij^2
//   (r_k g_k)_c = g_kc |g_k|^2 - y_kc                   (row sums = column sums)
// so a row of x costs one packed multiply-add per component, a third of forming the residual.
// Every input of a sample arrives through ONE in-order stream of LDS-DMA operations per wave: the
// node vectors (1 KiB pieces, 16 bytes per lane, into a header area), the gate row and the latent
// rows (4 bytes per lane), then x[b] as a flat run of 1 KiB pieces that starts at the 128-byte
// boundary at or below its first element (whole cache lines per piece), into a ring of MS_RING pieces; MS_AHEAD steps are in
// flight and the stream runs on into the wave's next sample without a gap.  The only wait is the
// issuing wave's own counted vmcnt; no other load is issued inside the loop and the result stores
// are counted exactly, so nothing drains the stream.  N is a template parameter: the whole schedule
// of a sample (which piece goes to which ring slot, which rows are complete after which piece, every
// LDS offset) is fixed at compile time and the loop body has no control flow and next to no scalar
// arithmetic - with N at run time the kernel spent more time on that than on the stream.  Lane c
// owns columns c and c+64 as one packed pair.  The piece in ring slot 0 is also written to a mirror
// slot behind the last one, so a row that starts in the last slot reads on without wrapping.  After
// the sweep every per-component quantity is uniform over the wave (sums by DPP, no LDS), so the
// log-sum-exp over the components is plain arithmetic.
#ifndef MS_AHEAD_N
#define MS_AHEAD_N 6
#endif
#ifndef MS_RING_N
#define MS_RING_N 8
#endif
#ifndef MS_WAVES_N
#define MS_WAVES_N 4
#endif
#ifndef MS_NT_N
#define MS_NT_N 1
#endif
#ifndef MS_BURST_N
#define MS_BURST_N 1
#endif
#ifndef MS_ALIGN_N
#define MS_ALIGN_N 128
#endif
constexpr int MS_ALIGN = MS_ALIGN_N;                 // bytes a sample's run is aligned down to (16 .. 128)
constexpr int MS_SLACK = MS_ALIGN / 4 - 1;           // floats the sample's first element can lie into its run
constexpr int MS_BURST = MS_BURST_N;                 // steps issued back to back (a contiguous run of that many KiB)
constexpr int MS_AHEAD = MS_AHEAD_N;                 // stream steps in flight per wave, at least
constexpr int MS_RING = MS_RING_N;                   // ring slots (power of two, >= MS_AHEAD + 2)
constexpr int MS_WAVES = MS_WAVES_N;                 // waves per workgroup
constexpr int MS_PIECE = 256;                        // floats per piece (64 lanes x 16 bytes)
constexpr int MS_RINGF = MS_RING * MS_PIECE;
constexpr int MS_GS = 8;                             // floats per row of the broadcast table (K <= 8)
constexpr int MS_MAXL = 128;                         // latent width this sweep takes
static_assert((MS_RING & (MS_RING - 1)) == 0 && MS_RING >= MS_AHEAD + MS_BURST + 1, "ring too small for the steps in flight");
// steps of a sample (and the first MS_AHEAD of the next) issued once consumer step c has issued its share
__host__ __device__ constexpr int ms_issued_upto(int c, int steps) {
    const int e = c - c % MS_BURST + MS_AHEAD + MS_BURST;
    return e < steps + MS_AHEAD ? e : steps + MS_AHEAD;
}
__host__ __device__ constexpr int ms_hdr_pieces(int K) { return (K * MO_NP + MS_PIECE - 1) / MS_PIECE; }
// header: node vectors [K][MO_NP] as they lie in memory, gate row (64), mu (128), logvar (128)
__host__ __device__ constexpr int ms_hdr_floats(int K) { return ms_hdr_pieces(K) * MS_PIECE + 64 + 2 * MS_MAXL; }
// per wave: ring + mirror slot, header, broadcast table
__host__ __device__ constexpr int ms_wave_floats(int K) {
    return MS_RINGF + MS_PIECE + ms_hdr_floats(K) + MO_NP * MS_GS;
}
// pieces of x per sample: enough for any phase of its first element, rounded up to whole turns of
// the ring so that every sample's run starts in slot 0 (the surplus pieces re-read one granule)
__host__ __device__ constexpr int ms_x_pieces_used(int N) { return (MS_SLACK + N * N + MS_PIECE - 1) / MS_PIECE; }
__host__ __device__ constexpr int ms_x_pieces(int N) { return (ms_x_pieces_used(N) + MS_RING - 1) / MS_RING * MS_RING; }
// the piece after whose arrival row r is complete whatever the phase (worst case: MS_SLACK floats)
__host__ __device__ constexpr int ms_row_piece(int r, int N) { return (MS_SLACK + (r + 1) * N - 1) / MS_PIECE; }
// first row that is NOT complete after piece p
__host__ __device__ constexpr int ms_rows_after(int p, int N) {
    int r = 0;
    while (r < N && ms_row_piece(r, N) <= p) ++r;
    return r;
}

// One LDS-DMA operation: lane l copies BYTES (16 or 4) from base + off to lds_base + BYTES * l (base
// uniform, off per lane).  The wait for it is the issuing wave's own counted vmcnt (ms_wait); the
// compiler does not see these loads.
template <int BYTES>
__device__ __forceinline__ void ms_dma(const void* base, unsigned off, unsigned lds_base) {
    unsigned keep;
    if constexpr (BYTES == 16) {
#if MS_NT_N
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_base) : "memory");
#else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_base) : "memory");
#endif
    } else {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_base) : "memory");
    }
}
template <int N_LEFT>
__device__ __forceinline__ void ms_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_LEFT) : "memory");
}
// Result stores of the streamed sweep, one instruction each by construction: vmcnt counts stores and
// LDS-DMA operations together, in issue order, so the waits must know exactly how many stores lie
// between the pieces they wait for and the newest operation.
__device__ __forceinline__ void ms_store(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ms_store(double* p, double v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
// sum over the wave, the same value in every lane (DPP adds, then one lane read)
__device__ __forceinline__ float ms_sum(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_dpp(v)), 63));
}
template <int I, int E, typename F>
__device__ __forceinline__ void ms_static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        ms_static_for<I + 1, E>(f);
    }
}
// The uniform addresses of one sample's inputs.
struct MsSample {
    const char* xrun;      // 16-byte aligned start of the run of x
    unsigned xlast;        // offset in the run of the last granule that holds an element of the sample
    const char* nodes;     // its node vectors
    const float *gate, *mu, *lv;
};

template <int K, int N>
__global__ __launch_bounds__(MS_WAVES * 64) void mdn_outer_sym_kernel(
    const float* __restrict__ nodes, const float* __restrict__ x, const float* __restrict__ logits,
    const float* __restrict__ log_sigma, int B, const float* __restrict__ mu, const float* __restrict__ logvar, int L,
    float* __restrict__ gnodes, float* __restrict__ glogits, float* __restrict__ glsig_rows, float* __restrict__ gmu,
    float* __restrict__ glogvar, float* __restrict__ resp, double* __restrict__ per_sample) {
    constexpr int HP = ms_hdr_pieces(K), TPX = ms_x_pieces(N), STEPS = HP + 1 + TPX, D = N * N;
    static_assert(N <= MO_NP - 1 && N >= 64 && STEPS > MS_AHEAD, "shape outside the streamed sweep");
    __shared__ __attribute__((aligned(1024))) float lds[MS_WAVES * ms_wave_floats(K)];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = blockIdx.x * MS_WAVES + wid;
    const int nwaves = gridDim.x * MS_WAVES;
    if (wave >= B) return;
    float* ring = lds + wid * ms_wave_floats(K);                 // [MS_RING + 1 slots][MS_PIECE]
    float* hdr = ring + MS_RINGF + MS_PIECE;                       // node vectors | gate row | mu | logvar
    float* h_gate = hdr + HP * MS_PIECE;
    float* h_mu = h_gate + 64;
    float* h_lv = h_mu + MS_MAXL;
    float* gs = hdr + ms_hdr_floats(K);                            // [row][MS_GS]
    const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)MV_LDS_PTR_(ring));
    const unsigned hdr_base = ring_base + (MS_RINGF + MS_PIECE) * 4;
    const unsigned gate_base = hdr_base + HP * MS_PIECE * 4;
    constexpr bool HAS1 = N > 64;                                  // N >= 64: every lane owns column `lane`
    const bool c1 = lane + 64 < N;
    const unsigned lane16 = lane * 16;
    // lanes of the last header piece that lie past the sample's node vectors re-read its last granule
    const unsigned noff_last = min(lane16, (unsigned)(K * MO_NP * 4 - 16 - (HP - 1) * MS_PIECE * 4));
    const unsigned goff = (lane < K ? lane : K - 1) * 4;
    const unsigned loff0 = (lane < L ? lane : L - 1) * 4, loff1 = (lane + 64 < L ? lane + 64 : L - 1) * 4;

    float ls[K], is2[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ls[k] = log_sigma[k];
        is2[k] = expf(-2.0f * ls[k]);
    }
    ms_wait<0>();      // from here on the only vector-memory operations are the stream's and the result stores

    auto sample_at = [&](int b) {
        MsSample sm;
        const unsigned long a = reinterpret_cast<unsigned long>(x + (long)b * D);
        sm.xrun = reinterpret_cast<const char*>(a & ~(unsigned long)(MS_ALIGN - 1));
        sm.xlast = (unsigned)((((a & (unsigned long)(MS_ALIGN - 1)) + (unsigned long)D * 4ul) - 1ul) & ~15ul);
        sm.nodes = reinterpret_cast<const char*>(nodes + (long)b * K * MO_NP);
        sm.gate = logits + (long)b * K;
        sm.mu = mu + (long)b * L;
        sm.lv = logvar + (long)b * L;
        return sm;
    };
    // step j of a sample's part of the stream: HP header pieces, the small rows, TPX pieces of x
    auto issue = [&](const MsSample& sm, auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < HP - 1) {
            ms_dma<16>(sm.nodes + j * (MS_PIECE * 4), lane16, hdr_base + j * (MS_PIECE * 4));
        } else if constexpr (j == HP - 1) {
            ms_dma<16>(sm.nodes + j * (MS_PIECE * 4), noff_last, hdr_base + j * (MS_PIECE * 4));
        } else if constexpr (j == HP) {
            ms_dma<4>(sm.gate, goff, gate_base);
            ms_dma<4>(sm.mu, loff0, gate_base + 64 * 4);
            ms_dma<4>(sm.mu, loff1, gate_base + 128 * 4);
            ms_dma<4>(sm.lv, loff0, gate_base + (64 + MS_MAXL) * 4);
            ms_dma<4>(sm.lv, loff1, gate_base + (128 + MS_MAXL) * 4);
        } else {
            constexpr int p = j - HP - 1, slot = p & (MS_RING - 1);
            // lanes past the sample's last granule (and all lanes of a surplus piece) re-read it
            const unsigned off = min(lane16 + (unsigned)(p * MS_PIECE * 4), sm.xlast);
            // the mirror copy first: once the piece itself counts as landed, so does its mirror
            if constexpr (slot == 0) ms_dma<16>(sm.xrun, off, ring_base + MS_RINGF * 4);
            ms_dma<16>(sm.xrun, off, ring_base + slot * (MS_PIECE * 4));
        }
    };
    // After a sample's result stores (S of them, all younger than the MS_AHEAD steps then in flight)
    // the next MS_AHEAD waits leave those stores outstanding as well.
    constexpr int S_PLAIN = 2 * K + 5, S_LATENT = S_PLAIN + 4;
    const bool latent_grads = gmu != nullptr;

    // wave w takes samples w, w + nwaves, ...: at any time the waves work on one contiguous window of x
    // (one consecutive block of samples per wave was measured 4 % slower)
    const int b_begin = wave, b_end = B, b_step = nwaves;
    MsSample cur = sample_at(b_begin);
    ms_static_for<0, MS_AHEAD>([&](auto jc) { issue(cur, jc); });
    bool first = true;
    for (int b = b_begin; b < b_end; b += b_step) {
        // past its last sample the stream re-reads this sample's first steps: harmless, a few KB per wave
        const MsSample nxt = sample_at(b + b_step < b_end ? b + b_step : b);
        const int shift = (int)((reinterpret_cast<unsigned long>(x + (long)b * D) & (unsigned long)(MS_ALIGN - 1)) >> 2);
        // one more step goes out, then all but the MS_AHEAD youngest operations (and the stores
        // behind them) have completed: step c of this sample is in LDS
        auto step = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c % MS_BURST == 0) {
                ms_static_for<c + MS_AHEAD, ms_issued_upto(c, STEPS)>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    if constexpr (t < STEPS) issue(cur, tc);
                    else issue(nxt, std::integral_constant<int, t - STEPS>{});
                });
            }
            constexpr int W = ms_issued_upto(c, STEPS) - (c + 1);       // steps younger than step c
            if constexpr (c < MS_AHEAD) {
                if (first) ms_wait<W>();
                else if (latent_grads) ms_wait<W + S_LATENT>();
                else ms_wait<W + S_PLAIN>();
            } else {
                ms_wait<W>();
            }
        };
        // ---- header: node vectors -> registers (own columns) and the broadcast table gs[row][k] ----
        ms_static_for<0, HP + 1>(step);
        mo_v2 g[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            g[k].x = hdr[k * MO_NP + lane];
            g[k].y = (HAS1 && c1) ? hdr[k * MO_NP + lane + 64] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            gs[lane * MS_GS + k] = g[k].x;
            if (lane + 64 < MO_NP) gs[(lane + 64) * MS_GS + k] = g[k].y;
        }
        // latent part and gate row (all of it from the header); the latent gradients wait in
        // registers for the store burst at the end of the sample
        float kl = 0.0f, lat_m[2] = {0.0f, 0.0f}, lat_v[2] = {0.0f, 0.0f};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int i = m * 64 + lane;
            if (i < L) {
                const float mm = h_mu[i], lv = h_lv[i];
                const float e = __expf(lv);
                kl += -0.5f * (1.0f + lv - mm * mm - e);
                lat_m[m] = mm;
                lat_v[m] = 0.5f * (e - 1.0f);
            }
        }
        float logpi[K];
        {
            float gl[K], gmax = -INFINITY, gsum = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                gl[k] = h_gate[k];
                gmax = fmaxf(gmax, gl[k]);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) gsum += __expf(gl[k] - gmax);
            const float lg = __logf(gsum);
#pragma unroll
            for (int k = 0; k < K; ++k) logpi[k] = gl[k] - gmax - lg;
        }
        // ---- the sweep ----
        mo_v2 y[K], sx2 = mo_v2{0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < K; ++k) y[k] = mo_v2{0.0f, 0.0f};
        const float* xrow = ring + shift + lane;          // this lane's element of row 0
        ms_static_for<0, TPX>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            step(std::integral_constant<int, HP + 1 + p>{});
            constexpr int r_lo = p == 0 ? 0 : ms_rows_after(p - 1, N), r_hi = ms_rows_after(p, N);
            ms_static_for<r_lo, r_hi>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                constexpr int off = (r * N) & (MS_RINGF - 1);
                mo_v2 xv;
                xv.x = xrow[off];
                xv.y = HAS1 ? xrow[off + 64] : 0.0f;
                sx2 = __builtin_elementwise_fma(xv, xv, sx2);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float gi = gs[r * MS_GS + k];
                    y[k] = __builtin_elementwise_fma(xv, mo_v2{gi, gi}, y[k]);
                }
            });
        });
        // what the lanes without a second column gathered is whatever lay behind the row: drop it
        if (!c1) sx2.y = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (!c1) y[k].y = 0.0f;
        // ---- per-sample scalars: everything below is the same in every lane ----
        const float sx = ms_sum(sx2.x + sx2.y);
        const float klsum = ms_sum(kl);
        float gg[K], ss[K], u[K], t[K];
        float tmin = INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            gg[k] = ms_sum(fmaf(g[k].x, g[k].x, g[k].y * g[k].y));
            const float q = ms_sum(fmaf(g[k].x, y[k].x, g[k].y * y[k].y));
            ss[k] = fmaxf(fmaf(gg[k], gg[k], sx) - 2.0f * q, 0.0f);
            u[k] = ss[k] * is2[k];
            t[k] = 0.5f * u[k] + (float)D * ls[k];
            tmin = fminf(tmin, t[k]);
        }
        // exponents relative to the component with the smallest one (see mixture_exponent)
        float um = 0.0f, lsm = 0.0f;
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
            if (t[k] == tmin) {
                um = u[k];
                lsm = ls[k];
            }
        }
        float a[K], amax = -INFINITY, asum = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            a[k] = logpi[k] - (0.5f * (u[k] - um) + (float)D * (ls[k] - lsm));
            amax = fmaxf(amax, a[k]);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) asum += __expf(a[k] - amax);
        const float lse = amax + __logf(asum);
        const double offset = 0.5 * (double)um + (double)D * ((double)lsm + 0.5 * (double)LOG_2PI);
        float r_own = 0.0f, gl_own = 0.0f, gs_own = 0.0f;
        float* dst = gnodes + (long)b * K * MO_NP;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float r = __expf(a[k] - lse);
            if (lane == k) {
                r_own = r;
                gl_own = __expf(logpi[k]) - r;
                gs_own = r * ((float)D - ss[k] * is2[k]);
            }
            const float ck2 = 2.0f * r * is2[k];
            ms_store(dst + k * MO_NP + lane, ck2 * fmaf(g[k].x, gg[k], -y[k].x));
            if (lane + 64 < MO_NP) ms_store(dst + k * MO_NP + lane + 64, c1 ? ck2 * fmaf(g[k].y, gg[k], -y[k].y) : 0.0f);
        }
        if (lane < K) {
            ms_store(resp + (long)b * K + lane, r_own);
            ms_store(glogits + (long)b * K + lane, gl_own);
            ms_store(glsig_rows + (long)b * K + lane, gs_own);
        }
        if (lane == 0) {
            ms_store(per_sample + 2 * b, offset - (double)lse);
            ms_store(per_sample + 2 * b + 1, (double)klsum);
        }
        if (latent_grads) {
            // always four stores: lanes past the row's end repeat its last element
            const int last = L - 1;
            const float last_m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(last >= 64 ? lat_m[1] : lat_m[0]), last & 63));
            const float last_v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(last >= 64 ? lat_v[1] : lat_v[0]), last & 63));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bool live = m * 64 + lane < L;
                const long j = (long)b * L + (live ? m * 64 + lane : last);
                ms_store(gmu + j, live ? lat_m[m] : last_m);
                ms_store(glogvar + j, live ? lat_v[m] : last_v);
            }
        }
        first = false;
        cur = nxt;
    }
    // nothing of this wave's stream may still be in flight when its LDS goes to another workgroup
    ms_wait<0>();
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
#ifndef MDN_SYM_OLD
#define MDN_SYM_OLD 0
#endif
#ifndef MS_BLOCKS_PER_CU_N
#define MS_BLOCKS_PER_CU_N 2
#endif
// LDS-streamed symmetric sweep: every workgroup resident at once (its LDS admits MS_BLOCKS_PER_CU_N per
// CU), and as many waves as give each the same number of samples
static unsigned mdn_sym_grid(int B) {
    const long max_waves = 256L * MS_BLOCKS_PER_CU_N * MS_WAVES;
    const long per_wave = (B + max_waves - 1) / max_waves;
    const long waves = (B + per_wave - 1) / per_wave;
    return (unsigned)((waves + MS_WAVES - 1) / MS_WAVES);
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
    hipLaunchKernelGGL(mdn_finalize_kernel, dim3(1), dim3(PAIR_NT), 0, st, (const double*)workspace, B, loss_out,
                       (const float*)nullptr, 0, (float*)nullptr);
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
    MV_REQUIRE(per_sample, E_ARG, "%s: workspace is null", fn);
    if (B > 0) {
        MV_REQUIRE(nodes && x && gate_logits && log_sigma && grad_nodes && grad_logits && grad_log_sigma_rows && resp,
                   E_ARG, "%s: null pointer", fn);
        MV_REQUIRE(L == 0 || (mu && logvar), E_ARG, "%s: mu/logvar is null", fn);
        MV_REQUIRE((grad_mu == nullptr) == (grad_logvar == nullptr), E_ARG, "%s: grad_mu and grad_logvar come together",
                   fn);
        // The LDS-streamed sweep exists for the two connectome sizes of the scripts (87 x 87 raw,
        // 68 x 68 cortical crop), 16-byte aligned node vectors and a latent row of at most MS_MAXL;
        // other symmetric input takes the row-wise symmetric sweep.
        const bool streamed = !MDN_SYM_OLD && x_symmetric && (N == 87 || N == 68) && aligned16(nodes) && L >= 1 &&
                              L <= MS_MAXL;
#define MV_SYM_STREAM(KK, NN)                                                                                        \
    hipLaunchKernelGGL((mdn_outer_sym_kernel<KK, NN>), dim3(mdn_sym_grid(B)), dim3(MS_WAVES * 64), 0, st, nodes, x,  \
                       gate_logits, log_sigma, B, mu, logvar, L, grad_nodes, grad_logits, grad_log_sigma_rows,       \
                       grad_mu, grad_logvar, resp, per_sample)
#define CALL(KK)                                                                                                     \
    if (streamed && N == 87)                                                                                         \
        MV_SYM_STREAM(KK, 87);                                                                                       \
    else if (streamed)                                                                                               \
        MV_SYM_STREAM(KK, 68);                                                                                       \
    else if (x_symmetric)                                                                                            \
        hipLaunchKernelGGL((mdn_outer_kernel<KK, true>), dim3(mdn_outer_grid(B)), dim3(MDN_NT), 0, st, nodes, x,     \
                           gate_logits, log_sigma, B, N, mu, logvar, L, grad_nodes, grad_logits,                     \
                           grad_log_sigma_rows, grad_mu, grad_logvar, resp, per_sample);                             \
    else                                                                                                             \
        hipLaunchKernelGGL((mdn_outer_kernel<KK, false>),                                                            \
                           dim3(mdn_outer_grid(B)),                                                                  \
                           dim3(MDN_NT), 0, st, nodes, x, gate_logits, log_sigma, B, N, mu, logvar, L, grad_nodes,   \
                           grad_logits, grad_log_sigma_rows, grad_mu, grad_logvar, resp, per_sample)
        MV_MDN_DISPATCH(K, CALL)
#undef CALL
#undef MV_SYM_STREAM
        MV_LAUNCH_CHECK("mdn_outer_kernel");
    }
    if (loss_out) {
        hipLaunchKernelGGL(mdn_finalize_kernel, dim3(1), dim3(PAIR_NT), 0, st, (const double*)per_sample, B, loss_out,
                           (const float*)nullptr, 0, (float*)nullptr);
        MV_LAUNCH_CHECK("mdn_finalize_kernel");
    }
    return 0;
}

int mdnvae_mdn_finalize(const void* workspace, int B, float* loss_out, const float* grad_log_sigma_rows, int K,
                        float* grad_log_sigma, void* stream) {
    MV_REQUIRE(B >= 0 && workspace && loss_out, E_ARG, "mdnvae_mdn_finalize: bad argument");
    MV_REQUIRE((grad_log_sigma_rows == nullptr) == (grad_log_sigma == nullptr), E_ARG,
               "mdnvae_mdn_finalize: the log-sigma rows and their sum come together");
    MV_REQUIRE(!grad_log_sigma || (K >= 1 && K <= MDN_MAXK), E_SHAPE, "mdnvae_mdn_finalize: %d components (1..%d supported)", K,
               MDN_MAXK);
    hipLaunchKernelGGL(mdn_finalize_kernel, dim3(grad_log_sigma ? 1 + K : 1), dim3(PAIR_NT), 0, (hipStream_t)stream,
                       (const double*)workspace, B, loss_out, grad_log_sigma_rows, K, grad_log_sigma);
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
    MV_REQUIRE(loss_out, E_ARG, "mdnvae_mdn_outer_fwdbwd_static: loss_out is null");
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
