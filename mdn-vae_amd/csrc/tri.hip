// First dense layer on symmetric connectomes.  x.view(-1, N*N) @ W^T (function/MM_Gate.py:157-165,
// function_FC_SC/SC-FC.py:95-104) sums w[h,i,j] x[b,i,j] over all N*N entries; when x[b] is
// symmetric the (i,j) and (j,i) terms share their input, so the product over the N(N+1)/2 entries
// of the upper triangle with the folded weight w[h,i,j] + w[h,j,i] is the same number with half the
// matrix-core work, and the weight gradient of (i,j) and (j,i) is one value written twice.
// These kernels do the packing, folding and unfolding around the library GEMMs; the packing
// kernel also verifies the symmetry it relies on.
//   t(i,j) = i*N - i(i-1)/2 + (j - i), 0 <= i <= j < N   (row-major walk of the upper triangle)
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

typedef float tri_v4 __attribute__((ext_vector_type(4)));
typedef tri_v4 tri_v4u __attribute__((aligned(4)));

__device__ __forceinline__ int tri_index(int i, int j, int N) { return i * N - ((i * (i - 1)) >> 1) + (j - i); }

// One workgroup of 512 threads per matrix (a sample of x, or the weights of one output unit; 256 threads: 0.147 ms
// for 16384 matrices of 87 x 87, 512: 0.123, 1024: 0.141).  The matrix is
// staged in LDS (16-byte pieces in, 16-byte LDS writes) so that the transposed element comes from
// LDS (stride N dwords: conflict-free for odd N, eight-way for N = 68 - an odd padded stride was
// measured there and bought nothing: the pass is bound by its global traffic).
//   FOLD = false: out[t] = m[i][j], *asym |= (m[i][j] != m[j][i])
//   FOLD = true : out[t] = m[i][j] + m[j][i] (i < j), m[i][i]
//   amax != NULL: amax[blockIdx % 64] = max(itself, bit pattern of the largest |out| written) - the scale
//   of the two-part fp16 operand split that reads `out` next is taken from the largest of the 64 slots.
//   One atomic without return per workgroup; 64 addresses because atomics on one address serialise
//   (one slot for all 16384 workgroups of a batch cost 30 us of a 150 us pass).
#ifndef TRI_NT_N
#define TRI_NT_N 512
#endif
constexpr int TRI_NT = TRI_NT_N;
template <bool FOLD>
__global__ __launch_bounds__(TRI_NT) void tri_pack_kernel(const float* __restrict__ src, int N, int T, int ld,
                                                       float* __restrict__ out, int* __restrict__ asym,
                                                       unsigned* __restrict__ amax) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ unsigned wave_max[TRI_NT / 64];
    const int P = N * N;
    const float* m = src + (long)blockIdx.x * P;
    float* o = out + (long)blockIdx.x * ld;
    const int P4 = P >> 2;
    for (int q = threadIdx.x; q < P4; q += TRI_NT)
        reinterpret_cast<tri_v4*>(sm)[q] = __builtin_nontemporal_load(reinterpret_cast<const tri_v4u*>(m) + q);
    for (int p = (P4 << 2) + threadIdx.x; p < P; p += TRI_NT) sm[p] = m[p];
    __syncthreads();
    // Thread -> entry of the triangle without a square root: rows v and N - 1 - v of the triangle hold N - v and
    // v + 1 entries, N + 1 together, so the triangle is (N + 1) / 2 "double rows" of exactly N + 1 slots (for odd N
    // the middle row pairs with itself and half of its slots stay empty).  Slot s = (v, q): q < N - v is entry
    // (v, v + q), the rest entry (N - 1 - v, N - 1 - v + q - (N - v)); consecutive slots are consecutive entries of
    // the packed row within each of the two segments, so a wave's store is two or three contiguous runs.  (The
    // form that walked consecutive packed entries and found their row by a square root plus corrections took
    // three times the instructions per entry and 164 us where this takes 157 at 16384 matrices of 87 x 87.)
    const int W = N + 1, V = (N + 1) >> 1;
    const float inv_w = 1.0f / (float)W;                   // (s + 0.5) / W stays 0.5 / W away from every integer
    bool differs = false;
    unsigned mx = 0;
    for (int s = threadIdx.x; s < V * W; s += TRI_NT) {
        const int v = (int)(((float)s + 0.5f) * inv_w), q = s - v * W;
        const bool first = q < N - v;
        const int i = first ? v : N - 1 - v;
        const int j = first ? v + q : q - 1;               // N - 1 - v + (q - (N - v))
        if (!first && i == v) continue;                    // odd N: the middle row is covered by its first half
        const float a = sm[i * N + j], b = sm[j * N + i];
        float val;
        if (FOLD) val = (j == i) ? a : a + b;
        else { val = a; differs |= !(a == b); }
        o[tri_index(i, j, N)] = val;
        mx = max(mx, __float_as_uint(val) & 0x7fffffffu);
    }
    for (int t = T + threadIdx.x; t < ld; t += TRI_NT) o[t] = 0.0f;      // padding columns of the row
    if (!FOLD && differs) atomicOr(asym, 1);
    if (amax) {
        // wave maximum on the DPP path (lanes that have no source take 0, the identity here); total in lane 63
#define TRI_DPP_MAX(ctrl, row_mask) mx = max(mx, (unsigned)__builtin_amdgcn_update_dpp(0, (int)mx, ctrl, row_mask, 0xf, false))
        TRI_DPP_MAX(0x111, 0xf);
        TRI_DPP_MAX(0x112, 0xf);
        TRI_DPP_MAX(0x114, 0xf);
        TRI_DPP_MAX(0x118, 0xf);
        TRI_DPP_MAX(0x142, 0xa);
        TRI_DPP_MAX(0x143, 0xc);
#undef TRI_DPP_MAX
        // one atomic per workgroup: they serialise at some 6 ns each whatever the address (one per wave, 131 k
        // of them for a batch of 16384, took 0.7 ms)
        if ((threadIdx.x & 63) == 63) wave_max[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned mw = wave_max[0];
#pragma unroll
            for (int w = 1; w < TRI_NT / 64; ++w) mw = max(mw, wave_max[w]);
            atomicMax(amax + (blockIdx.x & (MDNVAE_ABSMAX_SLOTS - 1)), mw);
        }
    }
}

// gw[h][i*N+j] = g[t(min(i,j), max(i,j)) * stride_t + h * stride_h]: a 64 x 64 tile of (p, h) goes
// through LDS so that both the reads (contiguous along whichever of t / h has stride 1) and the
// writes (contiguous along p) move whole cache lines.
constexpr int UF_T = 64;
__global__ __launch_bounds__(256) void tri_unfold_kernel(const float* __restrict__ g, int H, int N, long stride_t,
                                                         long stride_h, int accumulate, float* __restrict__ gw) {
    __shared__ float tile[UF_T][UF_T + 1];          // [p][h]
    const int P = N * N;
    const int p0 = blockIdx.x * UF_T, h0 = blockIdx.y * UF_T;
    const int a = threadIdx.x & 63, b4 = threadIdx.x >> 6;
    if (stride_h == 1) {                            // g is [T][H]: lanes along h
        for (int r = b4; r < UF_T; r += 4) {
            const int p = p0 + r, h = h0 + a;
            if (p < P && h < H) {
                const int i = p / N, j = p - i * N;
                const int t = i <= j ? tri_index(i, j, N) : tri_index(j, i, N);
                tile[r][a] = g[(long)t * stride_t + h];
            }
        }
    } else {                                        // g is [H][T]: lanes along p
        const int p = p0 + a;
        int t = 0;
        if (p < P) {
            const int i = p / N, j = p - i * N;
            t = i <= j ? tri_index(i, j, N) : tri_index(j, i, N);
        }
        for (int r = b4; r < UF_T; r += 4) {
            const int h = h0 + r;
            if (p < P && h < H) tile[a][r] = g[(long)t * stride_t + (long)h * stride_h];
        }
    }
    __syncthreads();
    for (int r = b4; r < UF_T; r += 4) {
        const int h = h0 + r, p = p0 + a;
        if (h < H && p < P) {
            const long o = (long)h * P + p;
            gw[o] = accumulate ? gw[o] + tile[a][r] : tile[a][r];
        }
    }
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

static int tri_launch(bool fold, const float* src, long rows, int N, int ld, float* out, int* asym, unsigned* amax,
                      void* stream, const char* who) {
    MV_REQUIRE(rows >= 0 && N >= 1, E_ARG, "%s: bad shape", who);
    MV_REQUIRE(N <= 192, E_SHAPE, "%s: N = %d exceeds the LDS-resident matrix limit (192)", who, N);
    MV_REQUIRE(rows <= 0x7fffffffL, E_SHAPE, "%s: too many matrices for one launch", who);
    if (rows == 0) return 0;
    MV_REQUIRE(src && out && (fold || asym), E_ARG, "%s: null pointer", who);
    const int T = N * (N + 1) / 2;
    MV_REQUIRE(ld >= T && ld - T < 256, E_ARG, "%s: row stride %d for %d triangle entries", who, ld, T);
    const size_t lds = sizeof(float) * (size_t)N * N;
    if (fold) {
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tri_pack_kernel<true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);   // static LDS: the wave maxima
            MV_REQUIRE(e == hipSuccess, (int)e, "%s: cannot raise the LDS limit: %s", who, hipGetErrorString(e));
        }
        hipLaunchKernelGGL(tri_pack_kernel<true>, dim3((unsigned)rows), dim3(TRI_NT), lds, (hipStream_t)stream, src, N, T,
                           ld, out, asym, amax);
    } else {
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tri_pack_kernel<false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);   // static LDS: the wave maxima
            MV_REQUIRE(e == hipSuccess, (int)e, "%s: cannot raise the LDS limit: %s", who, hipGetErrorString(e));
        }
        hipLaunchKernelGGL(tri_pack_kernel<false>, dim3((unsigned)rows), dim3(TRI_NT), lds, (hipStream_t)stream, src, N, T,
                           ld, out, asym, amax);
    }
    MV_LAUNCH_CHECK("tri_pack_kernel");
    return 0;
}

int mdnvae_tri_pack(const float* x, long B, int N, int ld, float* out, int* asym, void* stream) {
    return tri_launch(false, x, B, N, ld, out, asym, nullptr, stream, "mdnvae_tri_pack");
}

int mdnvae_tri_fold(const float* w, long H, int N, int ld, float* out, void* stream) {
    return tri_launch(true, w, H, N, ld, out, nullptr, nullptr, stream, "mdnvae_tri_fold");
}

static int tri_clear_slot(unsigned* absmax, void* stream, const char* who) {
    MV_REQUIRE(absmax, E_ARG, "%s: the largest-magnitude slot is NULL", who);
    const hipError_t e = clear_words(absmax, MDNVAE_ABSMAX_SLOTS, (hipStream_t)stream);
    MV_REQUIRE(e == hipSuccess, (int)e, "%s: cannot clear the slot: %s", who, hipGetErrorString(e));
    return 0;
}

int mdnvae_tri_pack_absmax(const float* x, long B, int N, int ld, float* out, int* asym, int reset, unsigned* absmax,
                           void* stream) {
    MV_REQUIRE(absmax, E_ARG, "mdnvae_tri_pack_absmax: the largest-magnitude slot is NULL");
    if (reset) {
        const int rc = tri_clear_slot(absmax, stream, "mdnvae_tri_pack_absmax");
        if (rc) return rc;
    }
    return tri_launch(false, x, B, N, ld, out, asym, absmax, stream, "mdnvae_tri_pack_absmax");
}

int mdnvae_tri_fold_absmax(const float* w, long H, int N, int ld, float* out, int reset, unsigned* absmax, void* stream) {
    MV_REQUIRE(absmax, E_ARG, "mdnvae_tri_fold_absmax: the largest-magnitude slot is NULL");
    if (reset) {
        const int rc = tri_clear_slot(absmax, stream, "mdnvae_tri_fold_absmax");
        if (rc) return rc;
    }
    return tri_launch(true, w, H, N, ld, out, nullptr, absmax, stream, "mdnvae_tri_fold_absmax");
}

int mdnvae_tri_unfold(const float* g, long H, int N, long stride_t, long stride_h, float* grad_w, int accumulate,
                      void* stream) {
    MV_REQUIRE(H >= 0 && N >= 1, E_ARG, "mdnvae_tri_unfold: bad shape");
    if (H == 0) return 0;
    MV_REQUIRE(g && grad_w, E_ARG, "mdnvae_tri_unfold: null pointer");
    MV_REQUIRE(stride_t >= 1 && stride_h >= 1 && (stride_t == 1 || stride_h == 1), E_ARG,
               "mdnvae_tri_unfold: one of the two strides must be 1");
    const long P = (long)N * N;
    MV_REQUIRE(P <= 0x7fffffffL && H <= 0x7fffffffL, E_SHAPE, "mdnvae_tri_unfold: shape too large");
    const dim3 grid((unsigned)((P + UF_T - 1) / UF_T), (unsigned)((H + UF_T - 1) / UF_T));
    MV_REQUIRE(grid.y <= 65535, E_SHAPE, "mdnvae_tri_unfold: too many output units for one launch");
    hipLaunchKernelGGL(tri_unfold_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, (int)H, N, stride_t, stride_h,
                       accumulate, grad_w);
    MV_LAUNCH_CHECK("tri_unfold_kernel");
    return 0;
}

}  // extern "C"
