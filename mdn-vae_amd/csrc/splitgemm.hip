// The two big dense layers of the FNN encoder towers, self.fc11 / self.fc12 of function/MM_Gate.py:157-158
// (Linear(N*N, 1024) + ReLU each, on the same input), and the weight gradients autograd derives for
// them, on the bf16 matrix cores at fp32 accuracy.
//
// An fp32 number is the exact sum of three bf16 numbers (8 significand bits each): x = x1 + x2 + x3.
// A product of two such sums, kept to the terms of relative size >= 2^-16,
//     x w ~= x1 w1 + (x1 w2 + x2 w1) + (x1 w3 + x2 w2 + x3 w1),
// leaves out only terms of 2^-24 and below, the size of one fp32 rounding.  Each of the six terms is a
// bf16 MFMA product accumulated in fp32, so a GEMM costs 6 bf16 products per fp32 one - 6/16 of the
// fp32 matrix rate - and both operands of every MFMA are reused three times from registers.
//
// Operands are split ahead of the GEMM into a tiled layout whose unit is exactly one MFMA operand
// fragment, [row tile of 32][k step of 16][part 3][k half 2][row 32][8 bf16] (1 KiB per fragment, the
// three parts of a (tile, step) contiguous): the GEMM stages fragments by LDS-DMA as whole 1 KiB
// pieces (fully coalesced, no address arithmetic per element) and reads them back with one
// conflict-free ds_read_b128 per lane.  Both operands are "k-contiguous" ([M][K] and [N][K]): the
// forward product takes x [B][T] and the folded weights [H][T]; the weight gradient takes x^T [T][B]
// and grad_pre^T [H][B], which the column-splitting kernel writes straight from the row-major tensors.
#include <cstdlib>
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

typedef __bf16 sg_frag __attribute__((ext_vector_type(8)));
typedef float sg_acc __attribute__((ext_vector_type(16)));
typedef float sg_v4 __attribute__((ext_vector_type(4)));

constexpr int SG_FRAG = 1024;                 // bytes of one fragment (32 rows x 16 k, bf16)
constexpr int SG_STEP = 3 * SG_FRAG;          // the three parts of one (row tile, k step)

// x = b[0] + b[1] + b[2] (each rounded to nearest; the remainders are exact in fp32)
__device__ __forceinline__ void sg_split(float x, __bf16 (&b)[3]) {
    b[0] = (__bf16)x;
    const float r1 = x - (float)b[0];
    b[1] = (__bf16)r1;
    const float r2 = r1 - (float)b[1];
    b[2] = (__bf16)r2;
}

__device__ __forceinline__ void sg_emit(const float (&v)[8], char* dst, int parts = 3) {
    sg_frag p[3];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 b[3];
        sg_split(v[e], b);
        p[0][e] = b[0];
        p[1][e] = b[1];
        p[2][e] = b[2];
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
        if (s < parts) *reinterpret_cast<sg_frag*>(dst + s * SG_FRAG) = p[s];
}

// ---- two fp16 parts instead of three bf16 parts ----
// x 2^e = h + l with h = fp16(x 2^e) and l = fp16(x 2^e - h): 22 significand bits in two parts, so
//     x w ~= h_x h_w + (h_x l_w + l_x h_w)
// is three fp16 MFMA products (each exact, accumulated in fp32) that leave out l_x l_w <= 2^-22 |x w| and
// carry each operand to 2^-22 of itself - four times an fp32 rounding - at half the matrix-core work of the
// six bf16 products.  fp16 spans 30 binades, so every tensor is first multiplied by the power of two that
// brings its largest magnitude into [2^14, 2^15) (exact; undone in the product's epilogue): an entry keeps
// its 22 bits as long as it is within 2^-17 of the tensor's largest one, below that its low part thins out
// and the absolute error stays under 2^-29 of the largest entry.
typedef _Float16 sg_hfrag __attribute__((ext_vector_type(8)));
constexpr float SG_UNIT_F16 = 16384.0f;       // what a row of ones holds in a scaled fp16 operand (exact at every scale)

// the power of two that brings a magnitude with bit pattern `bits` (the tensor's largest) into [2^14, 2^15)
__device__ __forceinline__ float sg_pow2_scale(unsigned bits) {
    const int e = (int)((bits >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.0f;                 // all zeros (or subnormal), inf / nan: as it stands
    const int se = 268 - e;                              // biased exponent of 2^(14 - (e - 127))
    return __uint_as_float((unsigned)(se < 253 ? se : 253) << 23);
}
__device__ __forceinline__ float sg_pow2_unscale(unsigned bits) { return 1.0f / sg_pow2_scale(bits); }
// A tensor's largest magnitude is kept as MDNVAE_ABSMAX_SLOTS partial maxima (the kernels that fill them
// spread their atomics over the slots); every wave that needs it takes one slot per lane.
static_assert(MDNVAE_ABSMAX_SLOTS == 64, "one slot per lane");
__device__ __forceinline__ unsigned sg_absmax(const unsigned* __restrict__ slots) {
    unsigned m = slots[threadIdx.x & 63];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    return m;
}

__device__ __forceinline__ void sg_emit_f16(const float (&v)[8], float s, char* dst) {
    sg_hfrag hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = v[e] * s;
        const _Float16 h = (_Float16)t;
        hi[e] = h;
        lo[e] = (_Float16)(t - (float)h);
    }
    *reinterpret_cast<sg_hfrag*>(dst) = hi;
    *reinterpret_cast<sg_hfrag*>(dst + SG_FRAG) = lo;
}

// amax[blockIdx % 64] = max(itself, largest |src[r][c]| this workgroup saw) as a bit pattern (magnitudes order like their bit patterns; a NaN
// orders above everything and so shows).  Flat over rows of `cols` floats with stride ld.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ src, long rows, long cols, long ld,
                                                     float floor_value, unsigned* __restrict__ amax) {
    __shared__ unsigned part[4];
    unsigned m = __float_as_uint(fabsf(floor_value));
    const long n = rows * cols;
    if (cols == ld && (cols & 3) == 0 && (reinterpret_cast<unsigned long>(src) & 15) == 0) {
        const sg_v4* s4 = reinterpret_cast<const sg_v4*>(src);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (long)gridDim.x * 256) {
            const sg_v4 v = __builtin_nontemporal_load(s4 + i);
            m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
            m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            const long r = i / cols;
            m = max(m, __float_as_uint(src[r * ld + (i - r * cols)]) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(amax + (blockIdx.x & (MDNVAE_ABSMAX_SLOTS - 1)), max(max(part[0], part[1]), max(part[2], part[3])));
}

// src [R][ld] fp32, K columns used -> tiled parts, operand row = src row, k = src column (ceil(K/16) k
// steps, zero beyond K).  One workgroup: 32 rows x 64 columns; a thread owns 8 consecutive columns of one
// row, which is one 16-byte chunk of each part.  Rows >= R are written as zeros.  VEC: K a multiple of
// 16, rows 16-byte aligned - two 16-byte loads per thread.  Otherwise the 32 x 64 block goes through LDS:
// consecutive lanes load consecutive floats, whatever the row stride and alignment.
// amax != NULL: two fp16 parts of src scaled by the power of two *amax gives, instead of three bf16 parts.
template <bool VEC>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, int R, int K, long ld,
                                                         int row_tile0, int KS, const unsigned* __restrict__ amax,
                                                         char* __restrict__ out) {
    __shared__ float tile[VEC ? 1 : 32][VEC ? 1 : 65];
    const unsigned scale_bits = amax ? sg_absmax(amax) : 0u;
    const int rt = blockIdx.x, r = threadIdx.x >> 3, c = threadIdx.x & 7;
    const int k0 = blockIdx.y * 64 + c * 8;
    const int row = rt * 32 + r;
    float v[8];
    if (VEC) {
        if (k0 >= K) return;
        if (row < R) {
            const sg_v4 a = *reinterpret_cast<const sg_v4*>(src + (long)row * ld + k0);
            const sg_v4 b = *reinterpret_cast<const sg_v4*>(src + (long)row * ld + k0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.0f;
        }
    } else {
        const int lc = threadIdx.x & 63, lr = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rr = rt * 32 + lr + 4 * i, kk = blockIdx.y * 64 + lc;
            tile[lr + 4 * i][lc] = (rr < R && kk < K) ? src[(long)rr * ld + kk] : 0.0f;
        }
        __syncthreads();
        if (k0 >= KS * 16) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[r][c * 8 + e];
    }
    const int ks = k0 >> 4, h = (k0 >> 3) & 1;
    char* dst = out + ((long)(row_tile0 + rt) * KS + ks) * SG_STEP + h * 512 + r * 16;
    if (amax) sg_emit_f16(v, sg_pow2_scale(scale_bits), dst);
    else sg_emit(v, dst);
}

// src [Kd][ld] fp32, R columns used -> tiled parts with operand row = src COLUMN and k = src row (the
// transposed operand of a weight-gradient product).  One workgroup: 64 src rows x 32 src columns
// through LDS; k beyond Kd is written as zeros (the tiled operand holds ceil(Kd/16) steps).
// gate (same shape and row stride as src, or NULL): an entry counts only where its gate is positive - the
// ReLU backward of a layer whose outputs are the gate, done while its output gradient is split.  ones_col:
// operand row R (one past the source's columns) is all ones over k < Kd, so that the product's row R is
// the column sum of the other operand (a bias gradient).  parts: how many of the three parts are written.
// With amax (two fp16 parts) that row holds 2^14 whatever the tensor's scale - ones times the scale could leave
// fp16's range - and the product is told which row it is (unit_row of split_gemm_kernel).
template <bool VEC>
__global__ __launch_bounds__(256) void split_cols_kernel(const float* __restrict__ src, const float* __restrict__ gate,
                                                         int Kd, int R, long ld, int ones_col, int parts, int row_tile0,
                                                         int KS, const unsigned* __restrict__ amax, char* __restrict__ out) {
    __shared__ float tile[64][33];
    const unsigned scale_bits = amax ? sg_absmax(amax) : 0u;
    const int rt = blockIdx.x, kb = blockIdx.y * 64;
    {
        const int rr = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int kk = kb + pass * 32 + rr, col = rt * 32 + c4;
            sg_v4 a = {0.0f, 0.0f, 0.0f, 0.0f};
            if (kk < Kd) {
                if (VEC && col + 3 < R) {
                    a = *reinterpret_cast<const sg_v4*>(src + (long)kk * ld + col);
                    if (gate) {
                        const sg_v4 y = *reinterpret_cast<const sg_v4*>(gate + (long)kk * ld + col);
                        a.x = y.x > 0.0f ? a.x : 0.0f;
                        a.y = y.y > 0.0f ? a.y : 0.0f;
                        a.z = y.z > 0.0f ? a.z : 0.0f;
                        a.w = y.w > 0.0f ? a.w : 0.0f;
                    }
                } else {
                    float* av = reinterpret_cast<float*>(&a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (col + e < R) {
                            const float v = src[(long)kk * ld + col + e];
                            av[e] = (gate && !(gate[(long)kk * ld + col + e] > 0.0f)) ? 0.0f : v;
                        } else if (ones_col && col + e == R) {
                            av[e] = amax ? SG_UNIT_F16 / sg_pow2_scale(scale_bits) : 1.0f;
                        }
                    }
                }
            }
            tile[pass * 32 + rr][c4] = a.x;
            tile[pass * 32 + rr][c4 + 1] = a.y;
            tile[pass * 32 + rr][c4 + 2] = a.z;
            tile[pass * 32 + rr][c4 + 3] = a.w;
        }
    }
    __syncthreads();
    const int r = threadIdx.x & 31, q = threadIdx.x >> 5;        // q = (k step within the 64, half)
    const int ks = (kb >> 4) + (q >> 1), h = q & 1;
    if (ks >= KS) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[q * 8 + e][r];
    char* dst = out + ((long)(row_tile0 + rt) * KS + ks) * SG_STEP + h * 512 + r * 16;
    if (amax) sg_emit_f16(v, sg_pow2_scale(scale_bits), dst);
    else sg_emit(v, dst, parts);
}

// Both operands a batch takes part in, from one read of it: src [Kd][ld] (one sample per row) becomes
//   the forward product's operand (operand row = sample, k = column, all ld columns; two fp16 parts of
//   src * 2^e(amax), as split_rows_kernel writes them) and
//   the weight-gradient product's operand (operand row = column c < R, plus the row of ones, k = sample; `parts`
//   bf16 parts, as split_cols_kernel writes them - or, parts == 0, two fp16 parts at the scale of the forward
//   operand with 2^14 in the unit row).
// One workgroup: 64 samples x 32 columns through LDS.  ld a multiple of 32 columns' worth of tiles (grid.x * 32 >= ld
// and >= R + ones_col), 16-byte aligned rows.
__global__ __launch_bounds__(256) void split_both_kernel(const float* __restrict__ src, int Kd, int R, long ld, int ones_col,
                                                         int parts, int KS_cols, int KS_rows,
                                                         const unsigned* __restrict__ amax, char* __restrict__ out_rows,
                                                         char* __restrict__ out_cols) {
    __shared__ float tile[64][33];
    const unsigned scale_bits = sg_absmax(amax);
    const int ct = blockIdx.x, kb = blockIdx.y * 64;
    {
        const int rr = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int kk = kb + pass * 32 + rr, col = ct * 32 + c4;
            sg_v4 a = {0.0f, 0.0f, 0.0f, 0.0f};
            if (kk < Kd && col < ld) a = *reinterpret_cast<const sg_v4*>(src + (long)kk * ld + col);
            tile[pass * 32 + rr][c4] = a.x;
            tile[pass * 32 + rr][c4 + 1] = a.y;
            tile[pass * 32 + rr][c4 + 2] = a.z;
            tile[pass * 32 + rr][c4 + 3] = a.w;
        }
    }
    __syncthreads();
    {   // transposed operand: thread = (column r of the tile, eight consecutive samples q)
        const int r = threadIdx.x & 31, q = threadIdx.x >> 5;
        const int ks = (kb >> 4) + (q >> 1), h = q & 1, c = ct * 32 + r;
        if (ks < KS_cols && ct * 32 < R + ones_col) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool in_k = kb + q * 8 + e < Kd;
                v[e] = c < R ? tile[q * 8 + e][r] : ((ones_col && c == R && in_k) ? 1.0f : 0.0f);
            }
            char* dst = out_cols + ((long)ct * KS_cols + ks) * SG_STEP + h * 512 + r * 16;
            if (parts == 0) {
                const float s = sg_pow2_scale(scale_bits);
                if (c == R) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= SG_UNIT_F16 / s;
                }
                sg_emit_f16(v, s, dst);
            } else {
                sg_emit(v, dst, parts);
            }
        }
    }
    {   // forward operand: thread = (sample sr of the block, eight consecutive columns pc)
        const int sr = threadIdx.x >> 2, pc = threadIdx.x & 3;
        const int sample = kb + sr, k0 = ct * 32 + pc * 8;
        if (sample < ((Kd + 31) & ~31) && k0 < KS_rows * 16) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[sr][pc * 8 + e];
            sg_emit_f16(v, sg_pow2_scale(scale_bits),
                        out_rows + ((long)(sample >> 5) * KS_rows + (k0 >> 4)) * SG_STEP + ((k0 >> 3) & 1) * 512 + (sample & 31) * 16);
        }
    }
}

// The transposed split operand of the gradient of a layer's outputs y = relu(...) when that gradient is itself
// a product: g [Kd][R] = gpre [Kd][J] . w [J][R] (+ dense [Kd][R] if given), the data gradient of the next
// dense layer (function/MM_Gate.py:159-160: fc111 / fc222 after fc11 / fc12).  operand[row_tile0 * 32 + c][r] =
// y[r][c] > 0 ? g[r][c] : 0 - the product, the ReLU gate and the split in one pass, g is never written.
// One workgroup: 64 rows r x 64 columns c.  The w slice [J][64] sits in LDS; a wave owns 16 rows and forms
// its 16 x 64 block on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) with its gpre rows in registers:
// lane (row, g) loads 16-byte pieces, so within a group of 16 j the lane's four values are j = 4 g + e, and
// the reduction runs in that order on both operands.  The block then crosses to the column-major emit
// through an LDS tile, where gate and addend are read as 128-byte row segments.
typedef float sg_f4 __attribute__((ext_vector_type(4)));
constexpr int DG_LDW = 68;        // row stride of the w slice: 16-byte rows, k groups on different banks
template <int J>
__global__ __launch_bounds__(256) void split_cols_dgrad_kernel(const float* __restrict__ gpre, const float* __restrict__ w,
                                                               const float* __restrict__ dense,
                                                               const float* __restrict__ gate, int Kd, int R, int parts,
                                                               int row_tile0, int KS, const unsigned* __restrict__ amax,
                                                               char* __restrict__ out) {
    static_assert(J % 16 == 0 && J <= 128, "the inner dimension runs in groups of 16; the w slice must fit static LDS");
    // amax != NULL: two fp16 parts of g * 2^e(*amax); *amax bounds |g| from above (dgrad_bound_kernel)
    const float f16_scale = amax ? sg_pow2_scale(sg_absmax(amax)) : 0.0f;
    __shared__ __attribute__((aligned(16))) float wsl[J * DG_LDW];
    __shared__ float tile[64][65];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.y * 64;
    // w[j][c0 .. c0 + 63] -> LDS, 16 bytes per thread and pass; it stays for all the row blocks this
    // workgroup walks (blockIdx.x, blockIdx.x + gridDim.x, ...)
    for (int i = t; i < J * 16; i += 256) {
        const int j = i >> 4, q = (i & 15) * 4;
        sg_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (c0 + q + 3 < R) v = *reinterpret_cast<const sg_f4*>(w + (long)j * R + c0 + q);
        else {
            float* vv = reinterpret_cast<float*>(&v);
            for (int e = 0; e < 4; ++e)
                if (c0 + q + e < R) vv[e] = w[(long)j * R + c0 + q + e];
        }
        *reinterpret_cast<sg_f4*>(wsl + j * DG_LDW + q) = v;
    }
    const int nrb = (Kd + 63) / 64;
    // a wave's 16 rows of gpre (rows past the end: the last row, dropped at the emit)
    sg_f4 a[J / 16];
    auto load_a = [&](int rb) {
        const int row = min(rb * 64 + wave * 16 + c16, Kd - 1);
#pragma unroll
        for (int i = 0; i < J / 16; ++i) a[i] = *reinterpret_cast<const sg_f4*>(gpre + (long)row * J + 16 * i + 4 * g);
    };
    load_a(blockIdx.x);
    const int r = t & 31, q = t >> 5;                    // emit: q = (k step within the 64 rows, half)
    for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const int r0 = rb * 64;
        // gate and addend of this block's emit: 128-byte row segments, issued before the product
        float y[2][8], d[2][8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c = c0 + half * 32 + r;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long at = (long)min(r0 + q * 8 + e, Kd - 1) * R + (c < R ? c : 0);
                y[half][e] = gate[at];
                d[half][e] = dense ? dense[at] : 0.0f;
            }
        }
        __syncthreads();                                 // the w slice is in; the previous block's emit is done
        sg_f4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = sg_f4{0.0f, 0.0f, 0.0f, 0.0f};
        // w values one group of 16 j (16 registers) ahead of the MFMAs that use them
        float bv[2][16];
        auto load_b = [&](int i, float (&bb)[16]) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) bb[e * 4 + ct] = wsl[(16 * i + 4 * g + e) * DG_LDW + ct * 16 + c16];
        };
        load_b(0, bv[0]);
#pragma unroll
        for (int i = 0; i < J / 16; ++i) {
            if (i + 1 < J / 16) load_b(i + 1, bv[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], bv[i & 1][e * 4 + ct], acc[ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (rb + (int)gridDim.x < nrb) load_a(rb + gridDim.x);       // the next block's rows, under the emit
        // accumulator register rr of column tile ct: row 4 g + rr of the wave's 16, column 16 ct + c16
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) tile[wave * 16 + 4 * g + rr][ct * 16 + c16] = acc[ct][rr];
        __syncthreads();
        const int ks = (r0 >> 4) + (q >> 1), h = q & 1;
        if (ks < KS) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = c0 + half * 32 + r;
                if (c0 + half * 32 < R) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gsum = tile[q * 8 + e][half * 32 + r] + d[half][e];
                        v[e] = (c < R && r0 + q * 8 + e < Kd && y[half][e] > 0.0f) ? gsum : 0.0f;
                    }
                    char* dst = out + ((long)(row_tile0 + blockIdx.y * 2 + half) * KS + ks) * SG_STEP + h * 512 + r * 16;
                    if (amax) sg_emit_f16(v, f16_scale, dst);
                    else sg_emit(v, dst, parts);
                }
            }
        }
    }
}

// The same pass with the product itself on the 16-bit matrix cores: gpre and the workgroup's w slice as two fp16 parts
// each (three v_mfma_f32_16x16x32_f16 per multiply-add, error <= 3 * 2^-22 of sum |g||w|, the level of the fp32
// product above, at under a quarter of its matrix-pipe time).  gpre is scaled by the power of two of its record
// (amax_g) and split in registers; the w slice by that of its own largest magnitude (found here: output columns
// belong to one workgroup, so every slice may carry its own scale), split once per workgroup into B fragments in
// LDS.  Output: two fp16 parts of the result scaled by the record `amax` (a bound of the result).
template <int J>
__global__ __launch_bounds__(256) void split_cols_dgrad_f16x2_kernel(const float* __restrict__ gpre, const float* __restrict__ w,
                                                                    const float* __restrict__ dense,
                                                                    const float* __restrict__ gate, int Kd, int R,
                                                                    int row_tile0, int KS, const unsigned* __restrict__ amax,
                                                                    const unsigned* __restrict__ amax_g, char* __restrict__ out) {
    static_assert(J % 32 == 0 && J <= 128, "k steps of 32");
    constexpr int NK = J / 32;
    const float f16_scale = sg_pow2_scale(sg_absmax(amax));
    const float sa = sg_pow2_scale(sg_absmax(amax_g));
    __shared__ __attribute__((aligned(16))) char bfr[2 * NK * 4 * 64 * 16];      // [part][k step][column tile][lane][8 halves]
    __shared__ float tile[64][65];
    __shared__ unsigned wmax[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.y * 64;
    // this thread's B fragments: (k step, column tile) pairs t / 64 + 4 i, lane t % 64: column c0 + 16 ct + c16, k = 32 ks + 8 g ..
    float wv[NK][8];
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int pair = wave + 4 * i, ks = pair >> 2, ct = pair & 3;
        const int c = c0 + ct * 16 + c16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = c < R ? w[(long)(32 * ks + 8 * g + j) * R + c] : 0.0f;
            wv[i][j] = v;
            m = max(m, __float_as_uint(v) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    const float sw = sg_pow2_scale(max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int pair = wave + 4 * i;
        sg_hfrag hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = wv[i][j] * sw;
            hi[j] = (_Float16)v;
            lo[j] = (_Float16)(v - (float)hi[j]);
        }
        *reinterpret_cast<sg_hfrag*>(bfr + (pair * 64 + lane) * 16) = hi;
        *reinterpret_cast<sg_hfrag*>(bfr + ((NK * 4 + pair) * 64 + lane) * 16) = lo;
    }
    const float unscale = 1.0f / (sa * sw);
    const int nrb = (Kd + 63) / 64;
    // a wave's 16 rows of gpre: lane (row c16, k group g) holds k = 32 ks + 8 g .. + 7 (rows past the end: the last row)
    sg_f4 a[NK][2];
    auto load_a = [&](int rb) {
        const int row = min(rb * 64 + wave * 16 + c16, Kd - 1);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            a[ks][0] = *reinterpret_cast<const sg_f4*>(gpre + (long)row * J + 32 * ks + 8 * g);
            a[ks][1] = *reinterpret_cast<const sg_f4*>(gpre + (long)row * J + 32 * ks + 8 * g + 4);
        }
    };
    load_a(blockIdx.x);
    const int r = t & 31, q = t >> 5;                    // emit: q = (k step within the 64 rows, half)
    for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const int r0 = rb * 64;
        float y[2][8], d[2][8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c = c0 + half * 32 + r;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long at = (long)min(r0 + q * 8 + e, Kd - 1) * R + (c < R ? c : 0);
                y[half][e] = gate[at];
                d[half][e] = dense ? dense[at] : 0.0f;
            }
        }
        __syncthreads();                                 // the B fragments are in; the previous block's emit is done
        sg_f4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = sg_f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            sg_hfrag ah, al;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = a[ks][j >> 2][j & 3] * sa;
                ah[j] = (_Float16)v;
                al[j] = (_Float16)(v - (float)ah[j]);
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const sg_hfrag bh = *reinterpret_cast<const sg_hfrag*>(bfr + ((ks * 4 + ct) * 64 + lane) * 16);
                const sg_hfrag bl = *reinterpret_cast<const sg_hfrag*>(bfr + ((NK * 4 + ks * 4 + ct) * 64 + lane) * 16);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[ct], 0, 0, 0);
            }
        }
        if (rb + (int)gridDim.x < nrb) load_a(rb + gridDim.x);       // the next block's rows, under the emit
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) tile[wave * 16 + 4 * g + rr][ct * 16 + c16] = acc[ct][rr] * unscale;
        __syncthreads();
        const int ks = (r0 >> 4) + (q >> 1), h = q & 1;
        if (ks < KS) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = c0 + half * 32 + r;
                if (c0 + half * 32 < R) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gsum = tile[q * 8 + e][half * 32 + r] + d[half][e];
                        v[e] = (c < R && r0 + q * 8 + e < Kd && y[half][e] > 0.0f) ? gsum : 0.0f;
                    }
                    char* dst = out + ((long)(row_tile0 + blockIdx.y * 2 + half) * KS + ks) * SG_STEP + h * 512 + r * 16;
                    sg_emit_f16(v, f16_scale, dst);
                }
            }
        }
    }
}


// An upper bound of |gpre . w + dense| over all entries, from the largest magnitudes of gpre and dense and the
// largest column sum of |w| [J][R]: amax_out = max(itself, |gpre|max * max_c sum_j |w[j][c]| + |dense|max).  A bound
// a few binades above the true maximum costs the fp16 split nothing that shows (its absolute error is 2^-29 of
// the scale's magnitude), and it needs no pass over the product.  One workgroup per 32 columns: thread (c, q)
// sums rows q, q + 8, ... of its column.
__global__ __launch_bounds__(256) void dgrad_bound_kernel(const unsigned* __restrict__ amax_gpre, const float* __restrict__ w,
                                                          int J, int R, const unsigned* __restrict__ amax_dense,
                                                          unsigned* __restrict__ amax_out) {
    __shared__ float part[8][32];
    const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.0f;
    if (c < R)
        for (int j = q; j < J; j += 8) s += fabsf(w[(long)j * R + c]);
    part[q][cl] = s;
    const float gmax = __uint_as_float(sg_absmax(amax_gpre));
    const float dmax = amax_dense ? __uint_as_float(sg_absmax(amax_dense)) : 0.0f;
    __syncthreads();
    if (threadIdx.x < 32) {
        float col = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) col += part[i][cl];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) col = fmaxf(col, __shfl_xor(col, o));
        if (threadIdx.x == 0) {
            // rounded up a little: the sums above are themselves rounded
            const float bound = (gmax * col) * 1.0001f + dmax;
            atomicMax(amax_out + (blockIdx.x & (MDNVAE_ABSMAX_SLOTS - 1)), __float_as_uint(bound));
        }
    }
}

// ---- the product ----
// C [M][N] = sum_k A[m][k] B[n][k] over tiled, split operands.  Workgroup of 8 waves (2 x 4), output
// tile (2 * WM * 32) x 256, wave tile (WM * 32) x 64; one k step (16) per barrier: the six MFMAs per
// fragment pair make a step 48 (WM = 4) or 24 (WM = 2) MFMAs per wave, long enough to cover the
// staging of the next step, which is issued first (LDS-DMA into the other buffer).
#define SG_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

enum { SG_EPI_BIAS_RELU = 0, SG_EPI_PLAIN = 1, SG_EPI_BIAS = 2 };

// TERMS = 6: all products down to 2^-24 of the result.  TERMS = 3: x1 w1 + x1 w2 + x2 w1 only (what is left
// out is 2^-16 of |x w| per term at most), the third part of neither operand is staged.
// F16 (with TERMS = 3): the operands hold two fp16 parts of power-of-two scaled tensors (sg_emit_f16) - h h + h l +
// l h on the fp16 MFMA, the scales undone in the epilogue from the two tensors' largest magnitudes.
// DEEP (with TERMS = 3): one k step per stage in a ring of four stage buffers, three stages in flight.  A
// three-term k step is 24 MFMAs per wave (0.35 us): one stage ahead, the next operands (L2 -> LDS, a
// microsecond and more under load) arrive after the MFMAs that hide them have run out, and every barrier waits
// for them.  The wait before the barrier is the wave's own counted vmcnt (two younger stages stay in flight).
template <int WM, int EPI, int TERMS, bool F16 = false, bool DEEP = false>
__global__ __launch_bounds__(512) void split_gemm_kernel(const char* __restrict__ A, const char* __restrict__ Bop,
                                                         const unsigned* __restrict__ amax_a,
                                                         const unsigned* __restrict__ amax_b,
                                                         int KS, int ntm, int ntn, int m_valid,
                                                         const float* __restrict__ bias0, const float* __restrict__ bias1,
                                                         float* __restrict__ out0, float* __restrict__ out1, int n_split,
                                                         long ld_out, int n_valid, int accumulate, int unit_row) {
    // unit_row (F16 only, -1 = none): the row of A that holds SG_UNIT_F16 in place of scaled ones; its products
    // are brought back with that factor instead of A's scale, so the output row is the column sum of B
    static_assert(!F16 || TERMS == 3, "two fp16 parts make a three-term product");
    static_assert(!DEEP || TERMS == 3, "the four-stage ring is sized for two-part operands");
    constexpr int TA = 2 * WM, TB = 8;                   // fragment rows-tiles of A / B per workgroup
    constexpr int NPART = TERMS == 6 ? 3 : 2;            // parts of a fragment that take part
    constexpr int KB = (TERMS == 6 || DEEP) ? 1 : 2;     // k steps per barrier
    constexpr int FSTEP = NPART * SG_FRAG;               // one (row tile, k step) in LDS
    constexpr int KSTAGE = (TA + TB) * FSTEP;            // one k step of all row tiles
    constexpr int STAGE = KB * KSTAGE;
    constexpr int PIECES = (TA + TB) * NPART * KB;
    constexpr int PPW = (PIECES + 7) / 8;                // pieces per wave (the last may be missing)
#ifndef SG_NBUF
#define SG_NBUF 4
#endif
    constexpr int NBUF = DEEP ? SG_NBUF : 2;
    static_assert(!DEEP || PIECES % 8 == 0, "counted waits need the same number of pieces in every wave");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    // tiles of one XCD (blocks g, g+8, ...) are consecutive: they share A row tiles through its L2
    // accumulate == 2: the k range is split over two workgroups per tile, both add their share atomically
    const int ksplit = accumulate == 2 ? 2 : 1;
    const int ntiles = ntm * ntn * ksplit;
    int tl = blockIdx.x;
    if ((ntiles & 7) == 0) tl = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
    const int tm = tl / (ntn * ksplit), tns = tl - tm * ntn * ksplit;
    const int tn = tns / ksplit, sk = tns - tn * ksplit;
    const int ks_per = ((KS + ksplit - 1) / ksplit + KB - 1) / KB * KB;
    const int ks_begin = sk * ks_per, ks_end = min(KS, ks_begin + ks_per);

    // this wave's staging pieces: piece q = wid + 8 i is part (q % NPART) of fragment row-tile
    // (q / NPART) % (TA + TB) of k step q / (NPART (TA + TB)) of the stage
    const char* src[PPW];
    int dst[PPW], kof[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wid + 8 * i;
        const int kk = q / (NPART * (TA + TB)), qq = q - kk * NPART * (TA + TB);
        const int t = qq / NPART, part = qq - NPART * t;
        const bool isA = t < TA;
        const long row_tile = isA ? (long)tm * TA + t : (long)tn * TB + (t - TA);
        src[i] = (isA ? A : Bop) + row_tile * KS * SG_STEP + (long)kk * SG_STEP + part * SG_FRAG + lane * 16;
        dst[i] = kk * KSTAGE + t * FSTEP + part * SG_FRAG;
        kof[i] = kk;
    }
    auto stage = [&](int buf, int ks) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if ((PIECES % 8 == 0 || wid + 8 * i < PIECES) && (KB == 1 || ks + kof[i] < ks_end))
                __builtin_amdgcn_global_load_lds(SG_GLOBAL_PTR(src[i] + (long)ks * SG_STEP),
                                                 SG_LDS_PTR(smem + buf * STAGE + dst[i]), 16, 0, 0);
        }
    };

    // output columns of this wave, and their biases (loaded now, long landed by the epilogue)
    const int n_tile0 = tn * 256 + wn * 64;
    const bool second = n_tile0 >= n_split;
    float* out = second ? out1 : out0;
    const int n_base = n_tile0 - (second ? n_split : 0);
    float bv[2] = {0.0f, 0.0f};
    if (EPI != SG_EPI_PLAIN) {
        const float* bias = second ? bias1 : bias0;
        bv[0] = n_base + (lane & 31) < n_valid ? bias[n_base + (lane & 31)] : 0.0f;
        bv[1] = n_base + 32 + (lane & 31) < n_valid ? bias[n_base + 32 + (lane & 31)] : 0.0f;
    }

    float unscale_a = 1.0f, unscale_b = 1.0f;
    if (F16) {
        unscale_a = sg_pow2_unscale(sg_absmax(amax_a));
        unscale_b = sg_pow2_unscale(sg_absmax(amax_b));
    }
    auto mfma = [](const sg_frag& a, const sg_frag& b, const sg_acc& c) {
        if constexpr (F16)
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sg_hfrag, a), __builtin_bit_cast(sg_hfrag, b), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    };

    sg_acc acc[WM][2];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // one k step of MFMAs from stage buffer `buf`, k step `kk` of the stage
    auto compute = [&](int buf, int kk) {
        const char* sa = smem + buf * STAGE + kk * KSTAGE + (wm * WM) * FSTEP + lane * 16;
        const char* sb = smem + buf * STAGE + kk * KSTAGE + (TA + wn * 2) * FSTEP + lane * 16;
        // B fragments and the first A row tile, then A row tile i + 1 is fetched while the MFMAs of
        // row tile i run: only the first block of a step waits on LDS
        sg_frag a[2][NPART], b[2][NPART];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < NPART; ++s) b[j][s] = *reinterpret_cast<const sg_frag*>(sb + j * FSTEP + s * SG_FRAG);
#pragma unroll
        for (int s = 0; s < NPART; ++s) a[0][s] = *reinterpret_cast<const sg_frag*>(sa + s * SG_FRAG);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            if (i + 1 < WM) {
#pragma unroll
                for (int s = 0; s < NPART; ++s)
                    a[(i + 1) & 1][s] = *reinterpret_cast<const sg_frag*>(sa + (i + 1) * FSTEP + s * SG_FRAG);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // smallest terms first
                if (TERMS == 6) {
                    acc[i][j] = mfma(a[i & 1][0], b[j][NPART - 1], acc[i][j]);
                    acc[i][j] = mfma(a[i & 1][1], b[j][1], acc[i][j]);
                    acc[i][j] = mfma(a[i & 1][NPART - 1], b[j][0], acc[i][j]);
                }
                acc[i][j] = mfma(a[i & 1][0], b[j][1], acc[i][j]);
                acc[i][j] = mfma(a[i & 1][1], b[j][0], acc[i][j]);
                acc[i][j] = mfma(a[i & 1][0], b[j][0], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (DEEP) {
        // stage s of this tile's k range lives in buffer s % 4; past the end the last stage is fetched again
        // (into a buffer nobody reads any more), so that every wave always has the same number of operations
        // in flight and one counted wait fits every iteration
        const int nks = ks_end - ks_begin;
        if (nks > 0) {
#pragma unroll
            for (int s = 0; s < NBUF - 1; ++s) stage(s, ks_begin + min(s, nks - 1));
        }
        for (int it = 0; it < nks; ++it) {
            // own pieces of stage `it` landed (two younger stages may be in flight); after the barrier
            // everybody's have, and everybody has finished with the buffer of stage it - 1
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NBUF - 2) * PPW) : "memory");
            stage((it + NBUF - 1) % NBUF, ks_begin + min(it + NBUF - 1, nks - 1));
            compute(it % NBUF, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // nothing in flight when the LDS goes back
    } else {
        stage(0, ks_begin);
        __syncthreads();
        int buf = 0;
        for (int ks = ks_begin; ks < ks_end; ks += KB, buf ^= 1) {
            if (ks + KB < ks_end) stage(buf ^ 1, ks + KB);
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                if (KB > 1 && ks + kk >= ks_end) break;
                compute(buf, kk);
            }
            __syncthreads();
        }
    }

    // epilogue: accumulator register e of a 32 x 32 tile is row (e & 3) + 8 (e >> 2) + 4 (lane >> 5),
    // column lane & 31.  A tile that lies wholly inside the valid rows stores without per-row tests.
    // n_valid counts columns of the selected output; accumulate adds to what the output holds
    const bool whole = (tm * TA + wm * WM + WM) * 32 <= m_valid && n_base + 64 <= n_valid && !accumulate;
    if (accumulate == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n_base + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int m0 = (tm * TA + wm * WM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + (e & 3) + 8 * (e >> 2);
                    if (m < m_valid && n < n_valid)
                        unsafeAtomicAdd(out + n + (long)m * ld_out,
                                        F16 ? acc[i][j][e] * (m == unit_row ? 1.0f / SG_UNIT_F16 : unscale_a) * unscale_b
                                            : acc[i][j][e]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n_base + j * 32 + (lane & 31);
        float* col = out + n;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int m0 = (tm * TA + wm * WM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (e & 3) + 8 * (e >> 2);
                float v = acc[i][j][e];
                if (F16) v = v * (m == unit_row ? 1.0f / SG_UNIT_F16 : unscale_a) * unscale_b;
                if (EPI == SG_EPI_BIAS_RELU) v = fmaxf(v + bv[j], 0.0f);
                if (EPI == SG_EPI_BIAS) v += bv[j];
                if (whole) col[(long)m * ld_out] = v;
                else if (m < m_valid && n < n_valid) col[(long)m * ld_out] = accumulate ? col[(long)m * ld_out] + v : v;
            }
        }
    }
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

size_t mdnvae_split_operand_bytes(long rows, long k) {
    const long rt = (rows + 31) / 32, ks = (k + 15) / 16;
    return (size_t)(rt > 0 ? rt : 0) * (size_t)(ks > 0 ? ks : 0) * SG_STEP;
}

static int split_rows_impl(const float* src, long R, long K, long ld, long row_tile0, const unsigned* amax, void* out,
                           void* stream) {
    MV_REQUIRE(R >= 0 && K >= 0 && ld >= K && row_tile0 >= 0, E_ARG, "mdnvae_split_rows: bad shape");
    if (R == 0 || K == 0) return 0;
    MV_REQUIRE(src && out, E_ARG, "mdnvae_split_rows: null pointer");
    MV_REQUIRE(aligned16(out), E_ALIGN, "mdnvae_split_rows: the operand must be 16-byte aligned");
    const long rt = (R + 31) / 32, KS = (K + 15) / 16, kb = (KS * 16 + 63) / 64;
    MV_REQUIRE(rt <= 0x7fffffffL && kb <= 65535, E_SHAPE, "mdnvae_split_rows: shape too large for one launch");
    const bool vec = K % 16 == 0 && ld % 4 == 0 && aligned16(src);
    if (vec)
        hipLaunchKernelGGL(split_rows_kernel<true>, dim3((unsigned)rt, (unsigned)kb), dim3(256), 0, (hipStream_t)stream, src,
                           (int)R, (int)K, ld, (int)row_tile0, (int)KS, amax, (char*)out);
    else
        hipLaunchKernelGGL(split_rows_kernel<false>, dim3((unsigned)rt, (unsigned)kb), dim3(256), 0, (hipStream_t)stream, src,
                           (int)R, (int)K, ld, (int)row_tile0, (int)KS, amax, (char*)out);
    MV_LAUNCH_CHECK("split_rows_kernel");
    return 0;
}

int mdnvae_split_rows(const float* src, long R, long K, long ld, long row_tile0, void* out, void* stream) {
    return split_rows_impl(src, R, K, ld, row_tile0, nullptr, out, stream);
}

int mdnvae_split_rows_f16(const float* src, long R, long K, long ld, long row_tile0, const unsigned* absmax, void* out,
                          void* stream) {
    MV_REQUIRE(absmax, E_ARG, "mdnvae_split_rows_f16: the largest-magnitude slot is NULL");
    return split_rows_impl(src, R, K, ld, row_tile0, absmax, out, stream);
}

int mdnvae_absmax(const float* src, long rows, long cols, long ld, float floor_value, int reset, unsigned* absmax,
                  void* stream) {
    MV_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols, E_ARG, "mdnvae_absmax: bad shape");
    MV_REQUIRE(absmax && (src || rows * cols == 0), E_ARG, "mdnvae_absmax: null pointer");
    if (reset) {
        const hipError_t e = clear_words(absmax, MDNVAE_ABSMAX_SLOTS, (hipStream_t)stream);
        MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_absmax: cannot clear the slot: %s", hipGetErrorString(e));
    }
    const long n = rows * cols;
    if (n == 0 && floor_value == 0.0f) return 0;
    const long blocks = n / 4 / 256 / 4 + 1;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, src,
                       rows, cols, ld, floor_value, absmax);
    MV_LAUNCH_CHECK("absmax_kernel");
    return 0;
}

static int split_cols_impl(const float* src, const float* gate, long Kd, long R, long ld, int ones_row, int parts,
                           long row_tile0, const unsigned* amax, void* out, void* stream) {
    MV_REQUIRE(R >= 0 && Kd >= 0 && ld >= R && row_tile0 >= 0, E_ARG, "mdnvae_split_cols: bad shape");
    MV_REQUIRE(parts >= 1 && parts <= 3, E_ARG, "mdnvae_split_cols: parts must be 1, 2 or 3");
    if (R == 0 || Kd == 0) return 0;
    MV_REQUIRE(src && out, E_ARG, "mdnvae_split_cols: null pointer");
    MV_REQUIRE(aligned16(out), E_ALIGN, "mdnvae_split_cols: the operand must be 16-byte aligned");
    const long rt = (R + (ones_row ? 1 : 0) + 31) / 32, kb = (Kd + 63) / 64, KS = (Kd + 15) / 16;
    MV_REQUIRE(rt <= 0x7fffffffL && kb <= 65535, E_SHAPE, "mdnvae_split_cols: shape too large for one launch");
    const bool vec = ld % 4 == 0 && aligned16(src) && (!gate || aligned16(gate));
    if (vec)
        hipLaunchKernelGGL(split_cols_kernel<true>, dim3((unsigned)rt, (unsigned)kb), dim3(256), 0, (hipStream_t)stream, src,
                           gate, (int)Kd, (int)R, ld, ones_row ? 1 : 0, parts, (int)row_tile0, (int)KS, amax, (char*)out);
    else
        hipLaunchKernelGGL(split_cols_kernel<false>, dim3((unsigned)rt, (unsigned)kb), dim3(256), 0, (hipStream_t)stream, src,
                           gate, (int)Kd, (int)R, ld, ones_row ? 1 : 0, parts, (int)row_tile0, (int)KS, amax, (char*)out);
    MV_LAUNCH_CHECK("split_cols_kernel");
    return 0;
}

int mdnvae_split_cols_gated(const float* src, const float* gate, long Kd, long R, long ld, int ones_row, int parts,
                            long row_tile0, void* out, void* stream) {
    return split_cols_impl(src, gate, Kd, R, ld, ones_row, parts, row_tile0, nullptr, out, stream);
}

int mdnvae_split_cols_f16(const float* src, const float* gate, long Kd, long R, long ld, int ones_row, long row_tile0,
                          const unsigned* absmax, void* out, void* stream) {
    MV_REQUIRE(absmax, E_ARG, "mdnvae_split_cols_f16: the largest-magnitude slot is NULL");
    return split_cols_impl(src, gate, Kd, R, ld, ones_row, 2, row_tile0, absmax, out, stream);
}

int mdnvae_split_rows_f16_cols(const float* src, long Kd, long R, long ld, int ones_row, int parts, const unsigned* absmax,
                               void* out_rows, void* out_cols, void* stream) {
    MV_REQUIRE(Kd >= 0 && R >= 0 && ld >= R, E_ARG, "mdnvae_split_rows_f16_cols: bad shape");
    MV_REQUIRE(parts >= 0 && parts <= 3, E_ARG, "mdnvae_split_rows_f16_cols: parts must be 0 (fp16), 1, 2 or 3");
    if (Kd == 0 || ld == 0) return 0;
    MV_REQUIRE(src && absmax && out_rows && out_cols, E_ARG, "mdnvae_split_rows_f16_cols: null pointer");
    MV_REQUIRE(ld % 16 == 0 && aligned16(src) && aligned16(out_rows) && aligned16(out_cols), E_ALIGN,
               "mdnvae_split_rows_f16_cols: a row stride that is a multiple of 16 floats and 16-byte aligned buffers");
    const long ctiles = ((ld > R + (ones_row ? 1 : 0) ? ld : R + (ones_row ? 1 : 0)) + 31) / 32, kb = (Kd + 63) / 64;
    MV_REQUIRE(ctiles <= 0x7fffffffL && kb <= 65535, E_SHAPE, "mdnvae_split_rows_f16_cols: shape too large for one launch");
    hipLaunchKernelGGL(split_both_kernel, dim3((unsigned)ctiles, (unsigned)kb), dim3(256), 0, (hipStream_t)stream, src, (int)Kd,
                       (int)R, ld, ones_row ? 1 : 0, parts, (int)((Kd + 15) / 16), (int)(ld / 16), absmax, (char*)out_rows,
                       (char*)out_cols);
    MV_LAUNCH_CHECK("split_both_kernel");
    return 0;
}

int mdnvae_split_cols(const float* src, long Kd, long R, long ld, long row_tile0, void* out, void* stream) {
    return mdnvae_split_cols_gated(src, nullptr, Kd, R, ld, 0, 3, row_tile0, out, stream);
}

static int split_cols_dgrad_impl(const float* gpre, const float* w, const float* dense, const float* gate, long Kd, long R,
                                 long J, int parts, long row_tile0, const unsigned* amax, void* out, void* stream,
                                 const unsigned* amax_g = nullptr) {
    MV_REQUIRE(R >= 0 && Kd >= 0 && row_tile0 >= 0, E_ARG, "mdnvae_split_cols_dgrad: bad shape");
    MV_REQUIRE(J == 64 || J == 128, E_SHAPE, "mdnvae_split_cols_dgrad: the inner dimension must be 64 or 128 (got %ld)", J);
    MV_REQUIRE(parts >= 1 && parts <= 3, E_ARG, "mdnvae_split_cols_dgrad: parts must be 1, 2 or 3");
    if (R == 0 || Kd == 0) return 0;
    MV_REQUIRE(gpre && w && gate && out, E_ARG, "mdnvae_split_cols_dgrad: null pointer");
    MV_REQUIRE(aligned16(out) && aligned16(gpre) && aligned16(w) && R % 4 == 0, E_ALIGN,
               "mdnvae_split_cols_dgrad: 16-byte aligned operands and a column count that is a multiple of 4");
    const long nrb = (Kd + 63) / 64, cb = (R + 63) / 64, KS = (Kd + 15) / 16;
    MV_REQUIRE(cb <= 65535, E_SHAPE, "mdnvae_split_cols_dgrad: shape too large for one launch");
    // three workgroups per CU, each keeping its slice of w for all the row blocks it walks
    const long rb = nrb < (768 + cb - 1) / cb ? nrb : (768 + cb - 1) / cb;
    if (amax_g) {
        // both factors as fp16 parts: the product on the 16-bit matrix cores
        if (J == 64)
            hipLaunchKernelGGL(split_cols_dgrad_f16x2_kernel<64>, dim3((unsigned)rb, (unsigned)cb), dim3(256), 0, (hipStream_t)stream,
                               gpre, w, dense, gate, (int)Kd, (int)R, (int)row_tile0, (int)KS, amax, amax_g, (char*)out);
        else
            hipLaunchKernelGGL(split_cols_dgrad_f16x2_kernel<128>, dim3((unsigned)rb, (unsigned)cb), dim3(256), 0, (hipStream_t)stream,
                               gpre, w, dense, gate, (int)Kd, (int)R, (int)row_tile0, (int)KS, amax, amax_g, (char*)out);
        MV_LAUNCH_CHECK("split_cols_dgrad_f16x2_kernel");
        return 0;
    }
#define DG_LAUNCH(JJ)                                                                                                 \
    hipLaunchKernelGGL(split_cols_dgrad_kernel<JJ>, dim3((unsigned)rb, (unsigned)cb), dim3(256), 0, (hipStream_t)stream, \
                       gpre, w, dense, gate, (int)Kd, (int)R, parts, (int)row_tile0, (int)KS, amax, (char*)out)
    if (J == 64) DG_LAUNCH(64);
    else DG_LAUNCH(128);
#undef DG_LAUNCH
    MV_LAUNCH_CHECK("split_cols_dgrad_kernel");
    return 0;
}

int mdnvae_split_cols_dgrad(const float* gpre, const float* w, const float* dense, const float* gate, long Kd, long R,
                            long J, int parts, long row_tile0, void* out, void* stream) {
    return split_cols_dgrad_impl(gpre, w, dense, gate, Kd, R, J, parts, row_tile0, nullptr, out, stream);
}

int mdnvae_split_cols_dgrad_f16(const float* gpre, const float* w, const float* dense, const float* gate, long Kd, long R,
                                long J, long row_tile0, const unsigned* absmax, const unsigned* absmax_gpre, void* out,
                                void* stream) {
    MV_REQUIRE(absmax, E_ARG, "mdnvae_split_cols_dgrad_f16: the largest-magnitude slot is NULL");
    return split_cols_dgrad_impl(gpre, w, dense, gate, Kd, R, J, 2, row_tile0, absmax, out, stream, absmax_gpre);
}

int mdnvae_dgrad_bound(const unsigned* absmax_gpre, const float* w, long J, long R, const unsigned* absmax_dense, int reset,
                       unsigned* absmax, void* stream) {
    MV_REQUIRE(J >= 1 && R >= 1 && R <= 0x7fffffffL && J <= 0x7fffffffL, E_ARG, "mdnvae_dgrad_bound: bad shape");
    MV_REQUIRE(absmax_gpre && w && absmax, E_ARG, "mdnvae_dgrad_bound: null pointer");
    if (reset) {
        const hipError_t e = clear_words(absmax, MDNVAE_ABSMAX_SLOTS, (hipStream_t)stream);
        MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_dgrad_bound: cannot clear the slot: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(dgrad_bound_kernel, dim3((unsigned)((R + 31) / 32)), dim3(256), 0, (hipStream_t)stream, absmax_gpre, w,
                       (int)J, (int)R, absmax_dense, absmax);
    MV_LAUNCH_CHECK("dgrad_bound_kernel");
    return 0;
}

static int split_gemm_impl(const void* a, const void* b, const unsigned* amax_a, const unsigned* amax_b, long M, long N,
                           long K, long m_valid, long n_valid, const float* bias0, const float* bias1, int relu,
                           int accumulate, float* out0, float* out1, long n_split, long ld_out, int tile_m, int terms,
                           long unit_row, void* stream) {
    const bool f16 = amax_a != nullptr;
    MV_REQUIRE(unit_row >= -1 && unit_row < M, E_ARG, "mdnvae_split_gemm_f16: unit_row is -1 or a row of a");
    constexpr bool deep = true;       // the four-stage ring; the two-buffer loop stays instantiable for comparison
    MV_REQUIRE(terms == 6 || terms == 3, E_ARG, "mdnvae_split_gemm: terms must be 6 (every product down to 2^-24) or 3");
    MV_REQUIRE(M >= 1 && N >= 1 && K >= 16 && m_valid >= 0 && m_valid <= M, E_ARG, "mdnvae_split_gemm: bad shape");
    MV_REQUIRE(n_valid >= 0 && n_valid <= n_split, E_ARG, "mdnvae_split_gemm: n_valid counts columns of one output (at most n_split)");
    MV_REQUIRE(accumulate >= 0 && accumulate <= 2, E_ARG, "mdnvae_split_gemm: accumulate is 0, 1 or 2");
    MV_REQUIRE(!(relu && accumulate), E_ARG, "mdnvae_split_gemm: the bias + ReLU epilogue does not accumulate");
    MV_REQUIRE(tile_m == 256 || tile_m == 128, E_ARG, "mdnvae_split_gemm: tile_m must be 128 or 256");
    MV_REQUIRE(M % tile_m == 0 && N % 256 == 0 && K % 16 == 0, E_SHAPE,
               "mdnvae_split_gemm: M must be a multiple of tile_m, N of 256, K of 16 (pad the split operands)");
    MV_REQUIRE(n_split >= 256 && n_split % 256 == 0 && n_split <= N, E_SHAPE,
               "mdnvae_split_gemm: n_split must be a multiple of 256 within N");
    MV_REQUIRE(a && b && out0 && (n_split == N || out1), E_ARG, "mdnvae_split_gemm: null pointer");
    MV_REQUIRE(!relu || (bias0 && (n_split == N || bias1)), E_ARG, "mdnvae_split_gemm: the bias + ReLU epilogue needs biases");
    MV_REQUIRE(!bias0 || n_split == N || bias1, E_ARG, "mdnvae_split_gemm: two outputs with a bias need both biases");
    MV_REQUIRE(!(bias0 && accumulate), E_ARG, "mdnvae_split_gemm: a bias epilogue does not accumulate");
    const bool bias_only = bias0 && !relu;
    MV_REQUIRE(aligned16(a) && aligned16(b), E_ALIGN, "mdnvae_split_gemm: operands must be 16-byte aligned");
    const long ntm = M / tile_m, ntn = N / 256;
    MV_REQUIRE(ntm * ntn <= 0x7fffffffL, E_SHAPE, "mdnvae_split_gemm: too many tiles");
    const int KS = (int)(K / 16);
    hipStream_t st = (hipStream_t)stream;
#define SG_LAUNCH(WM, EPI)                                                                                          \
    do {                                                                                                            \
        if (f16 && deep) SG_LAUNCH_T(WM, EPI, 3, true, true);                                                       \
        else if (f16) SG_LAUNCH_T(WM, EPI, 3, true, false);                                                         \
        else if (terms == 6) SG_LAUNCH_T(WM, EPI, 6, false, false);                                                 \
        else if (deep) SG_LAUNCH_T(WM, EPI, 3, false, true);                                                        \
        else SG_LAUNCH_T(WM, EPI, 3, false, false);                                                                 \
    } while (0)
#define SG_LAUNCH_T(WM, EPI, TERMS, F16, DEEP)                                                                               \
    do {                                                                                                            \
        const size_t lds = (size_t)(2 * WM + 8) * (TERMS == 6 ? 6 : (DEEP ? 2 * SG_NBUF : 8)) * SG_FRAG;                                   \
        /* per launch: the attribute belongs to the current device's copy of the kernel */                         \
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<WM, EPI, TERMS, F16, DEEP>), \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);           \
        MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_split_gemm: cannot raise the LDS limit: %s",                    \
                   hipGetErrorString(e));                                                                           \
        hipLaunchKernelGGL((split_gemm_kernel<WM, EPI, TERMS, F16, DEEP>), dim3((unsigned)(ntm * ntn * (accumulate == 2 ? 2 : 1))), dim3(512), lds, st, \
                           (const char*)a, (const char*)b, amax_a, amax_b, KS, (int)ntm, (int)ntn, (int)m_valid, bias0, bias1,      \
                           out0, out1 ? out1 : out0, (int)n_split, ld_out, (int)n_valid, accumulate, (int)unit_row); \
    } while (0)
    if (tile_m == 256) {
        if (relu) SG_LAUNCH(4, SG_EPI_BIAS_RELU);
        else if (bias_only) SG_LAUNCH(4, SG_EPI_BIAS);
        else SG_LAUNCH(4, SG_EPI_PLAIN);
    } else {
        if (relu) SG_LAUNCH(2, SG_EPI_BIAS_RELU);
        else if (bias_only) SG_LAUNCH(2, SG_EPI_BIAS);
        else SG_LAUNCH(2, SG_EPI_PLAIN);
    }
#undef SG_LAUNCH
#undef SG_LAUNCH_T
    MV_LAUNCH_CHECK("split_gemm_kernel");
    return 0;
}

int mdnvae_split_gemm(const void* a, const void* b, long M, long N, long K, long m_valid, long n_valid,
                      const float* bias0, const float* bias1, int relu, int accumulate, float* out0, float* out1,
                      long n_split, long ld_out, int tile_m, int terms, void* stream) {
    return split_gemm_impl(a, b, nullptr, nullptr, M, N, K, m_valid, n_valid, bias0, bias1, relu, accumulate, out0, out1,
                           n_split, ld_out, tile_m, terms, -1, stream);
}

int mdnvae_split_gemm_f16(const void* a, const void* b, const unsigned* absmax_a, const unsigned* absmax_b, long M, long N,
                          long K, long m_valid, long n_valid, const float* bias0, const float* bias1, int relu,
                          int accumulate, float* out0, float* out1, long n_split, long ld_out, int tile_m, long unit_row,
                          void* stream) {
    MV_REQUIRE(absmax_a && absmax_b, E_ARG, "mdnvae_split_gemm_f16: a largest-magnitude slot is NULL");
    return split_gemm_impl(a, b, absmax_a, absmax_b, M, N, K, m_valid, n_valid, bias0, bias1, relu, accumulate, out0, out1,
                           n_split, ld_out, tile_m, 3, unit_row, stream);
}

}  // extern "C"
