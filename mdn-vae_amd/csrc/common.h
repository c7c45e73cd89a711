// Shared host/device helpers for the connectome-VAE HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

namespace mdnvae {

// ---- error reporting -------------------------------------------------------
// Every extern "C" entry point returns 0 on success, a negative MDNVAE_E* code for a
// rejected argument, or a positive hipError_t when a launch fails.  The message of the
// last failure on the calling thread is kept for mdnvae_last_error().
constexpr int E_ARG = -1;      // bad pointer / size / enum
constexpr int E_SHAPE = -2;    // shape outside what the kernels are built for
constexpr int E_ALIGN = -3;    // pointer alignment

void set_error(const char* fmt, ...);

#define MV_REQUIRE(cond, code, ...)                \
    do {                                           \
        if (!(cond)) {                             \
            ::mdnvae::set_error(__VA_ARGS__);      \
            return (code);                         \
        }                                          \
    } while (0)

#define MV_LAUNCH_CHECK(name)                                                         \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            ::mdnvae::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return (int)e_;                                                           \
        }                                                                             \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers ----------------------------------------------------------
constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Wave sum on the DPP path (no LDS crossbar): inclusive scan within each row of 16 lanes, then
// the row totals are carried forward by the two row broadcasts; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define MV_DPP_ADD(ctrl, row_mask)                                                                       \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, row_mask, 0xf, false))
    MV_DPP_ADD(0x111, 0xf);   // row_shr:1
    MV_DPP_ADD(0x112, 0xf);   // row_shr:2
    MV_DPP_ADD(0x114, 0xf);   // row_shr:4
    MV_DPP_ADD(0x118, 0xf);   // row_shr:8
    MV_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    MV_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3
#undef MV_DPP_ADD
    return v;
}

// Sum over a workgroup of NT threads (multiple of 64); result valid in thread 0.
// `scratch` must hold NT/64 elements; fixed summation order (deterministic).
template <int NT, typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    T r = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) r += scratch[i];
    }
    __syncthreads();
    return r;
}

// Sums of the even and of the odd entries of `pairs` (n pairs of doubles: a per-sample or
// per-workgroup term and its KL part) by ONE workgroup of PAIR_NT threads, four pairs in flight per
// thread: the loop of a finalize kernel is a chain of dependent memory round trips, so its time
// is rounds x latency (16384 pairs: 18 us with 256 threads and one pair in flight, 3 us this way).
// Fixed order; the sums are valid in thread 0.  `red` holds PAIR_NT/64 doubles.
constexpr int PAIR_NT = 1024;
struct PairSum { double a, k; };
__device__ __forceinline__ PairSum block_pair_sum(const double* __restrict__ pairs, int n, double* red) {
    double a = 0.0, k = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * PAIR_NT < n; i += 4 * PAIR_NT) {
        double va[4], vk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            va[u] = pairs[2 * (long)(i + u * PAIR_NT)];
            vk[u] = pairs[2 * (long)(i + u * PAIR_NT) + 1];
        }
        a += (va[0] + va[1]) + (va[2] + va[3]);
        k += (vk[0] + vk[1]) + (vk[2] + vk[3]);
    }
    for (; i < n; i += PAIR_NT) {
        a += pairs[2 * (long)i];
        k += pairs[2 * (long)i + 1];
    }
    a = block_sum<PAIR_NT, double>(a, red);
    k = block_sum<PAIR_NT, double>(k, red);
    return PairSum{a, k};
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// Clearing a few words that kernels launched next update with atomics (largest-magnitude records, small
// accumulators).  A kernel, not hipMemsetAsync: inside a replayed HIP graph a memset node was seen to overlap the
// kernel node recorded after it (the record then missed the atomics that had already landed), while a kernel
// node always ran before the next one.
static __global__ void clear_words_kernel(unsigned* __restrict__ p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}
static inline hipError_t clear_words(void* p, int n_words, hipStream_t stream) {
    hipLaunchKernelGGL(clear_words_kernel, dim3(1), dim3(64), 0, stream, static_cast<unsigned*>(p), n_words);
    return hipGetLastError();
}

}  // namespace mdnvae
