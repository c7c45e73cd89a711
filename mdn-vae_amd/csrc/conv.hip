// Direct convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains).
//
// Replaces the ATen conv2d / convolution_backward calls of the encoder stack
//   function/VAE_CNN_NN.py:110-134 (stem 7x7 reflect-3, two stride-2 3x3, ten 3x3 reflect-1 convs)
// Activations stay NCHW fp32.  One workgroup owns one sample, a tile of PT consecutive output
// pixels and a tile of 96 output channels; per chunk of CK input channels it stages
//   * the input rows that tile needs, with padding (zero / reflection) and, optionally, the
//     instance norm + ReLU of the producer layer already applied, as slab[c][row][col] in LDS,
//   * the weights of the chunk as wl[tap][c][n] in LDS,
// and runs a tap loop of 32x32x2 MFMAs: A = 32 output pixels x 2 input channels read from the
// slab at a per-lane base + a per-tap scalar offset, B = 2 input channels x 32 output channels.
// Operands of tap t+1 are read from LDS while the MFMAs of tap t run.
//
// The data gradient is the same kernel over the output gradient: stride 1 = correlation with
// the flipped weights over a zero-padded plane; stride 2 = four launches, one per parity class of
// the input pixel, each using only the 1, 2, 2 or 4 taps that reach that class (no zero-inserted
// work).  The weight gradient is its own kernel: D[n][c] per tap accumulates over pixels
// (A = grad_out[n][pixel], B = slab[c][pixel + tap]); sample groups write partial sums that a
// second kernel adds in a fixed order.
#include "common.h"
#include "../../include/mdnvae_hip.h"
#include <algorithm>
#include <cstdlib>

namespace mdnvae {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int MAX_TAPS = 9;

struct ConvGeom {
    int B, Cin, H, W;        // input tensor [B,Cin,H,W]
    int Cout, OHf, OWf;      // output tensor [B,Cout,OHf,OWf]
    int SH, SW;              // grid of output pixels computed by this launch
    int osy, osx, oy0, ox0;  // grid pixel (a,b) is output pixel (a*osy+oy0, b*osx+ox0)
    int S;                   // slab step per grid pixel
    int pad, reflect;        // padding of the input seen through the slab
    int KHe;                 // slab rows needed beyond the first row of a grid row
    int IWp;                 // slab row length
    int slab_stride;         // floats per channel in the slab
    int CoutP;               // padded output-channel count of the packed weights
    int ntaps;
    int tap_off[MAX_TAPS];   // slab offset of tap t
    int tap_w[MAX_TAPS];     // index of tap t in the packed weights
    int relu_in;             // ReLU after the (optional) normalisation on load
    unsigned m_iwp, m_sw;    // magic multipliers: x / d == umulhi(x, m)  (x*d < 2^32)
};

__host__ __device__ inline unsigned magic_of(unsigned d) { return (unsigned)(0x100000000ull / d) + 1u; }
__device__ __forceinline__ int fdiv(int x, unsigned magic) { return (int)__umulhi((unsigned)x, magic); }

// value of the padded (and optionally normalised) input at slab coordinates (pr, pc)
__device__ __forceinline__ float virt_input(const ConvGeom& g, const float* __restrict__ plane, int pr, int pc,
                                            float mean, float rstd, bool norm) {
    int vy = pr - g.pad, vx = pc - g.pad;
    bool ok = true;
    if (g.reflect) {
        vy = vy < 0 ? -vy : (vy >= g.H ? 2 * (g.H - 1) - vy : vy);
        vx = vx < 0 ? -vx : (vx >= g.W ? 2 * (g.W - 1) - vx : vx);
    } else {
        ok = vy >= 0 && vy < g.H && vx >= 0 && vx < g.W;
    }
    float v = 0.0f;
    if (ok) {
        v = plane[vy * g.W + vx];
        if (norm) v = (v - mean) * rstd;
        if (g.relu_in) v = fmaxf(v, 0.0f);
    }
    return v;
}

// Stage `nch` channel slabs (rows pr0.., all IWp columns) of sample b, channels c_first.. into LDS.
// Loads are issued E at a time per thread before any of them is consumed.
template <int NT, int E>
__device__ __forceinline__ void stage_slab(const ConvGeom& g, const float* __restrict__ x,
                                           const float* __restrict__ stats, int b, int c_first, int nch, int c_limit,
                                           int pr0, int slab_elems, unsigned m_slab, float* __restrict__ slab,
                                           int tid) {
    const int total = nch * slab_elems;
    const long in_plane = (long)g.H * g.W;
    for (int base = 0; base < total; base += NT * E) {
        float v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = base + tid + e * NT;
            v[e] = 0.0f;
            if (idx < total) {
                const int c = fdiv(idx, m_slab), rem = idx - c * slab_elems;
                if (c_first + c < c_limit) {
                    const int r = fdiv(rem, g.m_iwp), col = rem - r * g.IWp;
                    const long pl = (long)b * g.Cin + c_first + c;
                    float mean = 0.0f, rstd = 1.0f;
                    if (stats) {
                        mean = stats[2 * pl];
                        rstd = stats[2 * pl + 1];
                    }
                    v[e] = virt_input(g, x + pl * in_plane, pr0 + r, col, mean, rstd, stats != nullptr);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = base + tid + e * NT;
            if (idx < total) {
                const int c = fdiv(idx, m_slab), rem = idx - c * slab_elems;
                slab[c * g.slab_stride + rem] = v[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ---------------------------------------------------------------------------------------
// The MFMA row dimension runs over (sample, grid pixel) pairs flattened over the whole batch,
// q = sample*P + pixel, so a tile of PT rows may end one sample and begin the next: the slab of a
// channel then has two segments (rows of the first sample's tail, rows of the next sample's
// head) and no MFMA row is wasted on per-sample padding.
//
// Staging is LDS-DMA (global_load_lds), double buffered: while the MFMA loop of chunk k runs,
// the raw input values of chunk k+1 are gathered straight into the other slab buffer (4 bytes
// per lane, source address from a per-workgroup offset table), its packed weights are copied
// 16 bytes per lane, and the {mean, rstd} pairs it needs are fetched too; no staging value
// passes through a register and there is one barrier per chunk.  Padding positions are never
// written by the DMA: they are filled once with 0, or with NaN when the producer's instance
// norm + ReLU is applied on the fly -- max((NaN-mean)*rstd, 0) is 0, which is what a padded
// activation must be.  The normalisation itself is applied to the A operand after its LDS
// read (3 VALU ops per operand, hidden under the 64-cycle MFMAs).
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
#define MV_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MV_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// TAPS > 0: tap count known at compile time (loop unrolled); PLAIN: the input is used as it stands
// (no normalisation / ReLU on load), which leaves the tap loop free of branches
template <int CK, int SI, int TAPS, bool PLAIN>
__global__ __launch_bounds__(256, (SI > 8 ? 2 : 3)) void conv_mfma_kernel(ConvGeom g, const float* __restrict__ x,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ wp,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WM = 2, WN = 3, WAVES = 4;
    constexpr int NT = WAVES * 64;
    constexpr int PT = WAVES * WM * 32;
    constexpr int NTILE = WN * 32;
    constexpr int NQ = NTILE / 4;
    constexpr int CP = CK / 2;
    constexpr int WBUF = MAX_TAPS * CK * NTILE;          // floats per weight buffer
    const int sbuf = CK * g.slab_stride;                  // floats per slab buffer
    float* wl0 = smem;                                    // [2][ntaps][CK][NTILE]
    float* slab0 = smem + 2 * WBUF;                       // [2][CK][slab_elems]
    float* st0 = slab0 + 2 * sbuf;                        // [2][2 samples][CK][2]
    static_assert(CK == WAVES, "wave w stages input channel c0 + w of every chunk");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int P = g.SH * g.SW;
    const long Q = (long)g.B * P;
    const long q0 = (long)blockIdx.x * PT;
    const int n0 = blockIdx.y * NTILE;
    const long qlast = min(Q - 1, q0 + PT - 1);
    const int s0 = (int)(q0 / P);
    const int pa0 = (int)(q0 - (long)s0 * P);
    const bool two = qlast >= (long)(s0 + 1) * P;
    // segment 0: sample s0, grid pixels pa0 .. pe0;  segment 1: sample s0+1, grid pixels 0 .. pe1
    const int pe0 = two ? P - 1 : (int)(qlast - (long)s0 * P);
    const int a00 = fdiv(pa0, g.m_sw);
    const int seg0 = ((fdiv(pe0, g.m_sw) - a00) * g.S + g.KHe) * g.IWp;
    const int pe1 = two ? (int)(qlast - (long)(s0 + 1) * P) : 0;
    const int seg1 = two ? (fdiv(pe1, g.m_sw) * g.S + g.KHe) * g.IWp : 0;
    const int slab_elems = seg0 + seg1;                   // <= slab_stride; also the channel stride
    const int in_plane = g.H * g.W;
    const bool norm = !PLAIN && stats != nullptr;

    // Wave w stages channel c0 + w of every chunk; lane l owns slab positions l, l+64, ...  Their
    // source offsets (relative to channel 0 of sample s0; -1 = padding, -2 = past the slab) are the
    // same for every chunk, so they are computed once and kept in registers.
    int soff[SI];
#pragma unroll
    for (int i = 0; i < SI; ++i) {
        const int rem = i * 64 + lane;
        soff[i] = -2;
        if (rem < slab_elems) {
            const int sg = rem >= seg0;
            const int rr = sg ? rem - seg0 : rem;
            const int r = fdiv(rr, g.m_iwp), col = rr - r * g.IWp;
            int vy = (sg ? 0 : a00 * g.S) + r - g.pad, vx = col - g.pad;
            bool ok = true;
            if (g.reflect) {
                vy = vy < 0 ? -vy : (vy >= g.H ? 2 * (g.H - 1) - vy : vy);
                vx = vx < 0 ? -vx : (vx >= g.W ? 2 * (g.W - 1) - vx : vx);
            } else {
                ok = vy >= 0 && vy < g.H && vx >= 0 && vx < g.W;
            }
            soff[i] = ok ? sg * g.Cin * in_plane + vy * g.W + vx : -1;
        }
    }
    {   // padding positions of both buffers, once (the DMA never writes them)
        const float fill = norm ? __builtin_nanf("") : 0.0f;
#pragma unroll
        for (int i = 0; i < SI; ++i)
            if (soff[i] == -1) {
                slab0[wave * slab_elems + i * 64 + lane] = fill;
                slab0[sbuf + wave * slab_elems + i * 64 + lane] = fill;
            }
    }

    int lane_base[WM], lane_seg[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const long q = min(Q - 1, q0 + (wave * WM + m) * 32 + l32);
        const int sg = q >= (long)(s0 + 1) * P;
        const int pp = (int)(q - (long)(s0 + sg) * P);
        const int a = fdiv(pp, g.m_sw), bb = pp - a * g.SW;
        lane_base[m] = (sg ? seg0 + a * g.S * g.IWp : (a - a00) * g.S * g.IWp) + bb * g.S + h * slab_elems;
        lane_seg[m] = sg;
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[m][n][j] = 0.0f;

    const int wtotal = g.ntaps * CK * NQ;
    const float* xs0 = x + (long)s0 * g.Cin * in_plane;
    const int s1 = min(s0 + 1, g.B - 1);

    auto issue = [&](int c0, int buf) {
        float* sdst = slab0 + buf * sbuf + wave * slab_elems;
        const float* src = xs0 + (long)(c0 + wave) * in_plane;
#pragma unroll
        for (int i = 0; i < SI; ++i)
            if (soff[i] >= 0)
                __builtin_amdgcn_global_load_lds(MV_GLOBAL_PTR(src + soff[i]), MV_LDS_PTR(sdst + i * 64), 4, 0, 0);
        float* wdst = wl0 + buf * WBUF;
        for (int base = 0; base < wtotal; base += NT) {
            const int i4 = base + tid;
            if (i4 < wtotal) {
                const int row = i4 / NQ, q = i4 - row * NQ;
                const int t = row / CK, c = row - t * CK;
                __builtin_amdgcn_global_load_lds(
                    MV_GLOBAL_PTR(wp + ((long)(g.tap_w[t] * g.Cin + c0 + c)) * g.CoutP + n0 + q * 4),
                    MV_LDS_PTR(wdst + (base + wave * 64) * 4), 16, 0, 0);
            }
        }
        if (norm && tid < 4 * CK) {      // {mean, rstd} of the chunk's channels for both samples
            const int sg = tid / (2 * CK), k = tid - sg * 2 * CK;
            __builtin_amdgcn_global_load_lds(MV_GLOBAL_PTR(stats + 2 * ((long)(sg ? s1 : s0) * g.Cin + c0) + k),
                                             MV_LDS_PTR(st0 + buf * 4 * CK), 4, 0, 0);
        }
    };

    issue(0, 0);
    int buf = 0;
    for (int c0 = 0; c0 < g.Cin; c0 += CK, buf ^= 1) {
        __syncthreads();     // chunk c0 has landed (the barrier drains the DMA) and the other buffer is free
        if (c0 + CK < g.Cin) issue(c0 + CK, buf ^ 1);
        const float* slab = slab0 + buf * sbuf;
        const float* wl = wl0 + buf * WBUF;
        float mean[WM][CP], rstd[WM][CP];
        if (norm) {
            const float* st = st0 + buf * 4 * CK;
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int cp = 0; cp < CP; ++cp) {
                    mean[m][cp] = st[(lane_seg[m] * CK + 2 * cp + h) * 2];
                    rstd[m][cp] = st[(lane_seg[m] * CK + 2 * cp + h) * 2 + 1];
                }
        }
#pragma unroll
        for (int t = 0; t < (TAPS > 0 ? TAPS : g.ntaps); ++t) {
            const float* sl = slab + g.tap_off[t];
            const float* wr = wl + (t * CK + h) * NTILE + l32;
            float A[CP][WM], Bv[CP][WN];
#pragma unroll
            for (int cp = 0; cp < CP; ++cp) {
#pragma unroll
                for (int m = 0; m < WM; ++m) A[cp][m] = sl[cp * 2 * slab_elems + lane_base[m]];
#pragma unroll
                for (int n = 0; n < WN; ++n) Bv[cp][n] = wr[cp * 2 * NTILE + n * 32];
            }
            if (PLAIN) {
            } else if (norm) {
#pragma unroll
                for (int cp = 0; cp < CP; ++cp)
#pragma unroll
                    for (int m = 0; m < WM; ++m) {
                        float v = (A[cp][m] - mean[m][cp]) * rstd[m][cp];
                        if (g.relu_in) v = fmaxf(v, 0.0f);
                        else v = (v != v) ? 0.0f : v;
                        A[cp][m] = v;
                    }
            } else if (g.relu_in) {
#pragma unroll
                for (int cp = 0; cp < CP; ++cp)
#pragma unroll
                    for (int m = 0; m < WM; ++m) A[cp][m] = fmaxf(A[cp][m], 0.0f);
            }
#pragma unroll
            for (int cp = 0; cp < CP; ++cp)
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cp][m], Bv[cp][n], acc[m][n], 0, 0, 0);
        }
    }
    // epilogue: accumulator registers 4g..4g+3 of a 32x32 tile hold rows 8g+4h .. 8g+4h+3 (four
    // consecutive q) of column l32
    const long oplane_sz = (long)g.OHf * g.OWf;
    const bool linear = g.osx == 1 && g.osy == 1 && g.SW == g.OWf;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int ch = n0 + n * 32 + l32;
        if (ch < g.Cout) {
            const float bv = bias ? bias[ch] : 0.0f;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const long q = q0 + (wave * WM + m) * 32 + 8 * gq + 4 * h;
                    if (q >= Q) continue;
                    const int sg = q >= (long)(s0 + 1) * P;
                    const int pp = (int)(q - (long)(s0 + sg) * P);
                    float* oplane = out + ((long)(s0 + sg) * g.Cout + ch) * oplane_sz;
                    if (linear && pp + 3 < P) {
                        f32x4_a4 o;
                        o.x = acc[m][n][4 * gq] + bv;
                        o.y = acc[m][n][4 * gq + 1] + bv;
                        o.z = acc[m][n][4 * gq + 2] + bv;
                        o.w = acc[m][n][4 * gq + 3] + bv;
                        *reinterpret_cast<f32x4_a4*>(oplane + pp) = o;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const long qk = q + k;
                            if (qk < Q) {
                                const int sk = qk >= (long)(s0 + 1) * P;
                                const int pk = (int)(qk - (long)(s0 + sk) * P);
                                const int a = fdiv(pk, g.m_sw), bb = pk - a * g.SW;
                                float* op = out + ((long)(s0 + sk) * g.Cout + ch) * oplane_sz;
                                op[(a * g.osy + g.oy0) * g.OWf + bb * g.osx + g.ox0] = acc[m][n][4 * gq + k] + bv;
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// forward / data-gradient kernel on the bf16 matrix cores
// ---------------------------------------------------------------------------------------
// PARTS = 1: both operands rounded to bf16, fp32 accumulation (one v_mfma_f32_32x32x16_bf16 per tile pair).
// PARTS = 3: every fp32 operand is the exact sum of three bf16 numbers, x = x1 + x2 + x3; the product keeps
// the six part products of relative size >= 2^-16 (x1 w1, x1 w2, x2 w1, x1 w3, x2 w2, x3 w1), accumulated
// smallest first - what is left out is below one fp32 rounding, so the result has fp32 accuracy at 6/16 of
// the cost of the fp32 matrix instruction.  PARTS = 2 keeps x1 w1 + x1 w2 + x2 w1 (2^-16 relative per
// product).
//
// Same row tiling as conv_mfma_kernel (flattened (sample, pixel) rows), with the K dimension of one MFMA
// running over 16 input channels.  Per chunk of 16 channels a thread stages whole slab positions: it
// loads the 16 channel values of a position (consecutive lanes = consecutive positions, so every load is
// coalesced along the plane), applies the producer's instance norm + ReLU if asked, splits / rounds to
// bf16 and writes 16-byte pieces, slab[part][k half][position][8 channels] - which is exactly the A operand
// fragment of a lane, read back with one ds_read_b128 per (32-row tile, tap, part).  Weights arrive
// pre-packed as whole B fragments ([part][tap][chunk][32-channel tile][lane][8 bf16], 1 KiB each) by
// 16-byte LDS-DMA.  LDS is single buffered: a second resident workgroup covers the staging of the first.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BF_CK = 16;

template <int PARTS>
__device__ __forceinline__ void bf_split(float x, __bf16 (&b)[PARTS]) {
    b[0] = (__bf16)x;
    if (PARTS > 1) {
        const float r1 = x - (float)b[0];
        b[PARTS > 1 ? 1 : 0] = (__bf16)r1;
        if (PARTS > 2) b[PARTS > 2 ? 2 : 0] = (__bf16)(r1 - (float)b[1]);
    }
}

template <int PARTS, int WM, int WN, int IT, int TAPS, bool PLAIN>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(ConvGeom g, const float* __restrict__ x,
                                                            const float* __restrict__ stats,
                                                            const char* __restrict__ wp,
                                                            const float* __restrict__ bias, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    constexpr int WAVES = 4;
    constexpr int NT = WAVES * 64;
    constexpr int PT = WAVES * WM * 32;
    constexpr int NTILE = WN * 32;
    constexpr int NTERM = PARTS == 1 ? 1 : PARTS == 2 ? 3 : 6;
    // part pairs (a, b) of the products kept, smallest first; the last NTERM entries are used:
    // PARTS = 3: (2,0) (1,1) (0,2) (1,0) (0,1) (0,0);  PARTS = 2: (1,0) (0,1) (0,0);  PARTS = 1: (0,0)
    constexpr int TA[6] = {2, 1, 0, PARTS >= 2 ? 1 : 0, 0, 0};
    constexpr int TB[6] = {0, 1, 2, 0, PARTS >= 2 ? 1 : 0, 0};
    const int wpart = MAX_TAPS * WN * 1024;               // bytes of one part of the weight tile
    char* wl = smem_b;                                    // [PARTS][ntaps][WN][1 KiB]
    char* slab = smem_b + PARTS * wpart;                  // [PARTS][2 k halves][slab_stride][16 B]
    const int half = g.slab_stride * 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int P = g.SH * g.SW;
    const long Q = (long)g.B * P;
    const long q0 = (long)blockIdx.x * PT;
    const int n0 = blockIdx.y * NTILE;
    const long qlast = min(Q - 1, q0 + PT - 1);
    const int s0 = (int)(q0 / P);
    const int pa0 = (int)(q0 - (long)s0 * P);
    const bool two = qlast >= (long)(s0 + 1) * P;
    const int pe0 = two ? P - 1 : (int)(qlast - (long)s0 * P);
    const int a00 = fdiv(pa0, g.m_sw);
    const int seg0 = ((fdiv(pe0, g.m_sw) - a00) * g.S + g.KHe) * g.IWp;
    const int pe1 = two ? (int)(qlast - (long)(s0 + 1) * P) : 0;
    const int seg1 = two ? (fdiv(pe1, g.m_sw) * g.S + g.KHe) * g.IWp : 0;
    const int slab_elems = seg0 + seg1;                   // <= slab_stride
    const int in_plane = g.H * g.W;
    const bool norm = !PLAIN && stats != nullptr;
    const int seg_off = g.Cin * in_plane;                 // source offset of sample s0 + 1 relative to s0

    // source offset of slab position it*256 + tid relative to channel 0 of sample s0
    // (-1 = padding, -2 = past the slab); the same for every chunk
    int soff[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int rem = i * NT + tid;
        soff[i] = -2;
        if (rem < slab_elems) {
            const int sg = rem >= seg0;
            const int rr = sg ? rem - seg0 : rem;
            const int r = fdiv(rr, g.m_iwp), col = rr - r * g.IWp;
            int vy = (sg ? 0 : a00 * g.S) + r - g.pad, vx = col - g.pad;
            bool ok = true;
            if (g.reflect) {
                vy = vy < 0 ? -vy : (vy >= g.H ? 2 * (g.H - 1) - vy : vy);
                vx = vx < 0 ? -vx : (vx >= g.W ? 2 * (g.W - 1) - vx : vx);
            } else {
                ok = vy >= 0 && vy < g.H && vx >= 0 && vx < g.W;
            }
            soff[i] = ok ? sg * seg_off + vy * g.W + vx : -1;
        }
    }
    int lane_pos[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const long q = min(Q - 1, q0 + (wave * WM + m) * 32 + l32);
        const int sg = q >= (long)(s0 + 1) * P;
        const int pp = (int)(q - (long)(s0 + sg) * P);
        const int a = fdiv(pp, g.m_sw), bb = pp - a * g.SW;
        lane_pos[m] = (sg ? seg0 + a * g.S * g.IWp : (a - a00) * g.S * g.IWp) + bb * g.S;
    }
    // the leading products and the small cross terms accumulate apart and meet once at the end: added
    // one by one to a sum thousands of times their size, the cross terms would lose most of their bits
    f32x16 acc[WM][WN], lo[PARTS > 1 ? WM : 1][PARTS > 1 ? WN : 1];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc[m][n][j] = 0.0f;
                if (PARTS > 1) lo[m][n][j] = 0.0f;
            }

    const int ntaps = TAPS > 0 ? TAPS : g.ntaps;
    const int wcount = ntaps * WN * 64;                   // 16-byte pieces of one part of one chunk's weights
    const int nchunk = g.Cin / BF_CK, ntile_all = g.CoutP / 32;
    const long wpart_g = 9L * nchunk * ntile_all * 1024;  // bytes of one part of the packed weights
    const float* xs0 = x + (long)s0 * g.Cin * in_plane;
    const int s1 = min(s0 + 1, g.B - 1);

    // raw values of the first PF slab positions of a chunk: loaded while the previous chunk's MFMAs run
    constexpr int PF = IT < 2 ? IT : 2;
    float pre[PF][BF_CK];
    auto load_item = [&](int i, int c0, float (&v)[BF_CK]) {
        // every load is issued whatever the position is (padding reads position 0 and drops the value):
        // a load under a condition becomes a branch and a wait of its own
        const int so = soff[i];
        const float* src = xs0 + (long)c0 * in_plane + (so >= 0 ? so : 0);
#pragma unroll
        for (int c = 0; c < BF_CK; ++c) v[c] = src[(long)c * in_plane];
    };
    auto store_item = [&](int i, int c0, float (&v)[BF_CK]) {
        const int so = soff[i];
        if (so == -2) return;
        if (!PLAIN && norm) {
            const float* stp = stats + 2 * ((long)(so >= seg_off ? s1 : s0) * g.Cin + c0);
#pragma unroll
            for (int c = 0; c < BF_CK; ++c) v[c] = (v[c] - stp[2 * c]) * stp[2 * c + 1];
        }
#pragma unroll
        for (int c = 0; c < BF_CK; ++c) {
            if (!PLAIN && g.relu_in) v[c] = fmaxf(v[c], 0.0f);
            v[c] = so >= 0 ? v[c] : 0.0f;
        }
        bf16x8 lo[PARTS], hi[PARTS];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            __bf16 bl[PARTS], bh[PARTS];
            bf_split<PARTS>(v[c], bl);
            bf_split<PARTS>(v[8 + c], bh);
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) {
                lo[pt][c] = bl[pt];
                hi[pt][c] = bh[pt];
            }
        }
        const int pos = i * NT + tid;
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) {
            *reinterpret_cast<bf16x8*>(slab + (2 * pt) * half + pos * 16) = lo[pt];
            *reinterpret_cast<bf16x8*>(slab + (2 * pt + 1) * half + pos * 16) = hi[pt];
        }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) load_item(u, 0, pre[u]);

    for (int ck = 0; ck < nchunk; ++ck) {
        const int c0 = ck * BF_CK;
        __syncthreads();                                  // the previous chunk's operand reads are done
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt)
            for (int base = 0; base < wcount; base += NT) {
                const int i = base + tid;
                if (i < wcount) {
                    const int t = i / (WN * 64), r = i - t * (WN * 64), j = r >> 6;
                    const long frag = ((long)g.tap_w[t] * nchunk + ck) * ntile_all + (n0 >> 5) + j;
                    __builtin_amdgcn_global_load_lds(MV_GLOBAL_PTR(wp + pt * wpart_g + frag * 1024 + (r & 63) * 16),
                                                     MV_LDS_PTR(wl + pt * wpart + (base + wave * 64) * 16), 16, 0, 0);
                }
            }
#pragma unroll
        for (int u = 0; u < PF; ++u) store_item(u, c0, pre[u]);
#pragma unroll
        for (int i0 = PF; i0 < IT; i0 += 2) {             // large slabs: the rest is loaded here, two positions at a time
            float v[2][BF_CK];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (i0 + u < IT) load_item(i0 + u, c0, v[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (i0 + u < IT) store_item(i0 + u, c0, v[u]);
        }
        __syncthreads();                                  // weights (the barrier drains the DMA) and slab are in place
        if (ck + 1 < nchunk) {
#pragma unroll
            for (int u = 0; u < PF; ++u) load_item(u, c0 + BF_CK, pre[u]);
        }
        const char* sl = slab + h * half;
#pragma unroll
        for (int t = 0; t < (TAPS > 0 ? TAPS : g.ntaps); ++t) {
            bf16x8 A[PARTS][WM], Bv[PARTS][WN];
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) {
#pragma unroll
                for (int m = 0; m < WM; ++m)
                    A[pt][m] = *reinterpret_cast<const bf16x8*>(sl + 2 * pt * half + (lane_pos[m] + g.tap_off[t]) * 16);
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    Bv[pt][n] = *reinterpret_cast<const bf16x8*>(wl + pt * wpart + (t * WN + n) * 1024 + lane * 16);
            }
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n)
                {
#pragma unroll
                    for (int e = 6 - NTERM; e < 5; ++e)
                        lo[PARTS > 1 ? m : 0][PARTS > 1 ? n : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            A[TA[e]][m], Bv[TB[e]][n], lo[PARTS > 1 ? m : 0][PARTS > 1 ? n : 0], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][m], Bv[0][n], acc[m][n], 0, 0, 0);
                }
        }
    }
    if (PARTS > 1) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) acc[m][n] += lo[PARTS > 1 ? m : 0][PARTS > 1 ? n : 0];
    }
    // epilogue: as conv_mfma_kernel (the accumulator layout of the two MFMA shapes is the same)
    const long oplane_sz = (long)g.OHf * g.OWf;
    const bool linear = g.osx == 1 && g.osy == 1 && g.SW == g.OWf;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int ch = n0 + n * 32 + l32;
        if (ch < g.Cout) {
            const float bv = bias ? bias[ch] : 0.0f;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const long q = q0 + (wave * WM + m) * 32 + 8 * gq + 4 * h;
                    if (q >= Q) continue;
                    const int sg = q >= (long)(s0 + 1) * P;
                    const int pp = (int)(q - (long)(s0 + sg) * P);
                    float* oplane = out + ((long)(s0 + sg) * g.Cout + ch) * oplane_sz;
                    if (linear && pp + 3 < P) {
                        f32x4_a4 o;
                        o.x = acc[m][n][4 * gq] + bv;
                        o.y = acc[m][n][4 * gq + 1] + bv;
                        o.z = acc[m][n][4 * gq + 2] + bv;
                        o.w = acc[m][n][4 * gq + 3] + bv;
                        *reinterpret_cast<f32x4_a4*>(oplane + pp) = o;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const long qk = q + k;
                            if (qk < Q) {
                                const int sk = qk >= (long)(s0 + 1) * P;
                                const int pk = (int)(qk - (long)(s0 + sk) * P);
                                const int a = fdiv(pk, g.m_sw), bb = pk - a * g.SW;
                                float* op = out + ((long)(s0 + sk) * g.Cout + ch) * oplane_sz;
                                op[(a * g.osy + g.oy0) * g.OWf + bb * g.osx + g.ox0] = acc[m][n][4 * gq + k] + bv;
                            }
                        }
                    }
                }
            }
        }
    }
}

// w [Cout,Cin,3,3] fp32 -> B fragments of the bf16 kernel: fwd[part][tap][Cin/16][CoutP/32][lane][8] with
// lane = (k half, output channel in the tile), entry j = input channel 16*chunk + 8*half + j; bwd the
// same with the roles of the channel counts exchanged and the taps flipped (the data gradient).
__global__ void conv_pack_bf16_kernel(const float* __restrict__ w, int Cout, int Cin, int CoutP, int CinP, int parts,
                                      __bf16* __restrict__ fwd, __bf16* __restrict__ bwd) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nf = 9L * Cin * CoutP, nb = 9L * Cout * CinP;
    float vf = 0.0f, vb = 0.0f;
    if (idx < nf) {
        const int j = (int)(idx & 7), ln = (int)((idx >> 3) & 63);
        long r = idx >> 9;
        const int nt = (int)(r % (CoutP / 32)); r /= (CoutP / 32);
        const int ck = (int)(r % (Cin / 16)), tap = (int)(r / (Cin / 16));
        const int n = nt * 32 + (ln & 31), c = ck * 16 + 8 * (ln >> 5) + j;
        vf = (n < Cout) ? w[((long)n * Cin + c) * 9 + tap] : 0.0f;
    }
    if (bwd && idx < nb) {
        const int j = (int)(idx & 7), ln = (int)((idx >> 3) & 63);
        long r = idx >> 9;
        const int ct = (int)(r % (CinP / 32)); r /= (CinP / 32);
        const int ck = (int)(r % (Cout / 16)), tap = (int)(r / (Cout / 16));
        const int c = ct * 32 + (ln & 31), n = ck * 16 + 8 * (ln >> 5) + j;
        vb = (c < Cin) ? w[((long)n * Cin + c) * 9 + (8 - tap)] : 0.0f;
    }
    for (int pt = 0; pt < parts; ++pt) {
        const __bf16 bf = (__bf16)vf, bb = (__bf16)vb;
        if (idx < nf) fwd[pt * nf + idx] = bf;
        if (bwd && idx < nb) bwd[pt * nb + idx] = bb;
        vf -= (float)bf;
        vb -= (float)bb;
    }
}

// Stem: one input channel, so the MFMA K dimension runs over the taps (49 -> 50, the last with
// zero weight).  wp is [50][CoutP].
template <int WAVES, int WM, int WN>
__global__ __launch_bounds__(WAVES * 64) void conv_stem_kernel(ConvGeom g, int KS, const float* __restrict__ x,
                                                               const float* __restrict__ wp,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ out, float* __restrict__ tile_stats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = WAVES * 64;
    constexpr int PT = WAVES * WM * 32;
    constexpr int NTILE = WN * 32;
    const int KK = KS * KS;
    const int KKP = (KK + 1) & ~1;
    float* slab = smem;
    float* wl = smem + g.slab_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int P = g.SH * g.SW;
    const int p0 = blockIdx.x * PT, n0 = blockIdx.y * NTILE, b = blockIdx.z;
    const int a0 = fdiv(p0, g.m_sw);
    const int plast = min(P - 1, p0 + PT - 1);
    const int nrows = (fdiv(plast, g.m_sw) - a0) * g.S + g.KHe;
    const int pr0 = a0 * g.S;
    const int slab_elems = nrows * g.IWp;
    int lane_base[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int p = min(P - 1, p0 + (wave * WM + m) * 32 + l32);
        const int a = fdiv(p, g.m_sw), bb = p - a * g.SW;
        lane_base[m] = (a - a0) * g.S * g.IWp + bb * g.S;
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[m][n][j] = 0.0f;

    stage_slab<NT, 8>(g, x, nullptr, b, 0, 1, 1, pr0, slab_elems, magic_of(slab_elems), slab, tid);
    for (int i = tid; i < KKP * NTILE; i += NT) {
        const int tap = i / NTILE, n = i - tap * NTILE;
        wl[i] = wp[(long)tap * g.CoutP + n0 + n];
    }
    __syncthreads();
#pragma unroll 5
    for (int step = 0; step < KKP / 2; ++step) {
        const int tap = 2 * step + h;
        const int ta = min(tap, KK - 1);            // address of the padded tap stays inside the slab
        const int ky = ta / KS, kx = ta - ky * KS;
        const int toff = ky * g.IWp + kx;
        float a[WM], bb[WN];
#pragma unroll
        for (int m = 0; m < WM; ++m) a[m] = slab[lane_base[m] + toff];
#pragma unroll
        for (int n = 0; n < WN; ++n) bb[n] = wl[tap * NTILE + n * 32 + l32];
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[n], acc[m][n], 0, 0, 0);
    }
    // Write-out.  A lane's accumulators are 16 positions of ONE channel, and the output is channel-major planes: stored
    // from the registers, every instruction would touch 64 planes with 4 bytes each.  The tile crosses through LDS,
    // [channel][position], one 32-channel column at a time, and leaves as 1 KiB runs of a plane (16 bytes per lane; the
    // planes start at 4-byte phases when H W is odd, which the hardware's unaligned 16-byte stores take).
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f4a __attribute__((ext_vector_type(4)));
    constexpr int ROWP = PT + 4;
    float* ep = smem;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        __syncthreads();                                   // the slab / the previous column has been read
        const int chl = n0 + n * 32 + l32;
        const float bv = (bias && chl < g.Cout) ? bias[chl] : 0.0f;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f4a*>(ep + l32 * ROWP + (wave * WM + m) * 32 + 8 * q + 4 * h) =
                    f4a{acc[m][n][4 * q] + bv, acc[m][n][4 * q + 1] + bv, acc[m][n][4 * q + 2] + bv, acc[m][n][4 * q + 3] + bv};
        __syncthreads();
        static_assert(PT / 4 == 64, "a wave reads back one channel's row of the tile per pass");
        for (int item = tid; item < 32 * (PT / 4); item += NT) {
            const int c = item / (PT / 4), q4 = item - c * (PT / 4);
            const int ch = n0 + n * 32 + c, p = p0 + 4 * q4;
            if (ch >= g.Cout) continue;                        // (a whole wave: c is the same for its 64 lanes)
            const f4a v = *reinterpret_cast<const f4a*>(ep + c * ROWP + 4 * q4);
            float* o = out + ((long)b * g.Cout + ch) * P + p;
            if (p + 3 < P) {
                *reinterpret_cast<f4u*>(o) = v;
            } else {
                for (int i = 0; i < 4 && p + i < P; ++i) o[i] = v[i];
            }
            if (tile_stats) {
                // this tile's share of the plane's instance-norm statistics: mean and sum of squared deviations of its
                // own pixels (combined over the tiles of a plane by stem_stats_finalize_kernel)
                const int cnt = min(PT, P - p0);
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) sum += (p + i < P) ? v[i] : 0.0f;
                const float mean = wave_sum_all(sum) / (float)cnt;
                float m2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = v[i] - mean;
                    m2 += (p + i < P) ? d * d : 0.0f;
                }
                m2 = wave_sum_all(m2);
                if (lane == 0) {
                    float* ts = tile_stats + (((long)b * g.Cout + ch) * gridDim.x + blockIdx.x) * 2;
                    ts[0] = mean;
                    ts[1] = m2;
                }
            }
        }
    }
}

// stats[plane] = {mean, 1 / sqrt(var + eps)} from the per-tile {mean, sum of squared deviations} of conv_stem_kernel:
// tiles are combined pairwise in double (mean and M2 of a union from those of its parts), in tile order.
__global__ __launch_bounds__(256) void stem_stats_finalize_kernel(const float* __restrict__ tile_stats, long nplanes, int ntiles,
                                                                  int P, int PT, float eps, float* __restrict__ stats) {
    const long pl = (long)blockIdx.x * 256 + threadIdx.x;
    if (pl >= nplanes) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int t = 0; t < ntiles; ++t) {
        const double nb = (double)min(PT, P - t * PT), mb = tile_stats[(pl * ntiles + t) * 2], sb = tile_stats[(pl * ntiles + t) * 2 + 1];
        const double tot = n + nb, delta = mb - mean;
        mean += delta * nb / tot;
        m2 += sb + delta * delta * n * nb / tot;
        n = tot;
    }
    stats[2 * pl] = (float)mean;
    stats[2 * pl + 1] = (float)(1.0 / sqrt(m2 / n + (double)eps));
}

// ---------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------
// w [Cout,Cin,KS,KS] -> fwd[tap][cin][CoutP]  and  bwd[flipped tap][cout][CinP]
__global__ void conv_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int KK, int CoutP, int CinP,
                                 float* __restrict__ fwd, float* __restrict__ bwd) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nf = (long)KK * Cin * CoutP;
    const long nb = (long)KK * Cout * CinP;
    if (idx < nf) {
        const int n = (int)(idx % CoutP);
        const long t = idx / CoutP;
        const int c = (int)(t % Cin), tap = (int)(t / Cin);
        fwd[idx] = (n < Cout) ? w[((long)n * Cin + c) * KK + tap] : 0.0f;
    }
    if (bwd && idx < nb) {
        const int c = (int)(idx % CinP);
        const long t = idx / CinP;
        const int n = (int)(t % Cout), tap = (int)(t / Cout);
        bwd[idx] = (c < Cin) ? w[((long)n * Cin + c) * KK + (KK - 1 - tap)] : 0.0f;
    }
}
// stem: w [Cout,1,KS,KS] -> [KKP][CoutP] (tap-major, zero padded)
__global__ void conv_pack_stem_kernel(const float* __restrict__ w, int Cout, int KK, int KKP, int CoutP,
                                      float* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KKP * CoutP) return;
    const int tap = idx / CoutP, n = idx - tap * CoutP;
    dst[idx] = (tap < KK && n < Cout) ? w[n * KK + tap] : 0.0f;
}

// ---------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------
// Workgroup tile: 64 output channels x 64 input channels x 9 taps; 4 waves as 2 (n) x 2 (c).
// K dimension = output pixels of the samples of this group, in chunks of ptw pixels (whole grid
// rows).  Both operand tiles are staged by LDS-DMA into double buffers while the previous
// chunk's MFMAs run; the source-offset table of chunk k+2 is computed during chunk k.  The
// producer's instance norm + ReLU is applied to the B operand after its LDS read.
// partial [G][9][Cout][Cin]
constexpr int WG_NT = 256;
template <bool NORM>
__global__ __launch_bounds__(WG_NT, 2) void conv_wgrad_kernel(ConvGeom g, const float* __restrict__ x,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ gout, int ptw, int cslab,
                                                              int samples_per_group, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KKT = 9;
    const int GS = ptw + 1;                          // odd row stride of the grad_out tile
    const int gbuf = 64 * GS, sbuf = cslab * g.slab_stride;
    float* gl0 = smem;                               // [2][64][GS]
    float* slab0 = smem + 2 * gbuf;                  // [2][cslab][slab_stride]
    int* tbl0 = reinterpret_cast<int*>(slab0 + 2 * sbuf);   // [2][slab_stride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int wn = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int P = g.SH * g.SW;
    const int b_begin = blockIdx.z * samples_per_group;
    const int b_end = min(g.B, b_begin + samples_per_group);
    const int npc = (P + ptw - 1) / ptw;             // chunks per plane
    const int nchunks = (b_end - b_begin) * npc;
    const unsigned m_gs = magic_of(GS), m_ss = magic_of(g.slab_stride);
    const int in_plane = g.H * g.W;
    constexpr bool norm = NORM;
    const float fill = norm ? __builtin_nanf("") : 0.0f;

    f32x16 acc[KKT];
#pragma unroll
    for (int t = 0; t < KKT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.0f;
    // lanes of a channel row that does not exist read row 0 (their products are discarded)
    const int crow = (c0 + wc * 32 + l32 < g.Cin) ? wc * 32 + l32 : 0;

    // geometry of chunk j of a plane (rows are whole grid rows, or the plane is one chunk)
    auto chunk_rows = [&](int j, int& a0, int& nrows) {
        const int p0 = j * ptw;
        a0 = fdiv(p0, g.m_sw);
        nrows = (fdiv(min(P - 1, p0 + ptw - 1), g.m_sw) - a0) * g.S + g.KHe;
    };
    auto make_tbl = [&](int ci) {                    // source offsets of chunk ci into tbl[ci & 1]
        if (ci >= nchunks) return;
        int a0, nrows;
        chunk_rows(ci % npc, a0, nrows);
        int* tbl = tbl0 + (ci & 1) * g.slab_stride;
        for (int rem = tid; rem < nrows * g.IWp; rem += WG_NT) {
            const int r = fdiv(rem, g.m_iwp), col = rem - r * g.IWp;
            int vy = a0 * g.S + r - g.pad, vx = col - g.pad;
            bool ok = true;
            if (g.reflect) {
                vy = vy < 0 ? -vy : (vy >= g.H ? 2 * (g.H - 1) - vy : vy);
                vx = vx < 0 ? -vx : (vx >= g.W ? 2 * (g.W - 1) - vx : vx);
            } else {
                ok = vy >= 0 && vy < g.H && vx >= 0 && vx < g.W;
            }
            tbl[rem] = ok ? vy * g.W + vx : -1;
        }
    };
    auto issue = [&](int ci) {                       // DMA of chunk ci into buffer ci & 1
        if (ci >= nchunks) return;
        const int b = b_begin + ci / npc, j = ci % npc, p0 = j * ptw;
        int a0, nrows;
        chunk_rows(j, a0, nrows);
        const int slab_elems = nrows * g.IWp;
        float* gdst = gl0 + (ci & 1) * gbuf;
        for (int base = 0; base < gbuf; base += WG_NT) {
            const int idx = base + tid;
            if (idx < gbuf) {
                const int n = fdiv(idx, m_gs), pp = idx - n * GS;
                if (pp < ptw && n0 + n < g.Cout && p0 + pp < P)
                    __builtin_amdgcn_global_load_lds(MV_GLOBAL_PTR(gout + ((long)b * g.Cout + n0 + n) * P + p0 + pp),
                                                     MV_LDS_PTR(gdst + base + wave * 64), 4, 0, 0);
            }
        }
        float* sdst = slab0 + (ci & 1) * sbuf;
        const int* tbl = tbl0 + (ci & 1) * g.slab_stride;
        const float* xb = x + ((long)b * g.Cin + c0) * in_plane;
        for (int base = 0; base < sbuf; base += WG_NT) {
            const int idx = base + tid;
            if (idx < sbuf) {
                const int c = fdiv(idx, m_ss), rem = idx - c * g.slab_stride;
                if (rem < slab_elems && c0 + c < g.Cin) {
                    const int off = tbl[rem];
                    if (off >= 0)
                        __builtin_amdgcn_global_load_lds(MV_GLOBAL_PTR(xb + (long)c * in_plane + off),
                                                         MV_LDS_PTR(sdst + base + wave * 64), 4, 0, 0);
                    else
                        sdst[idx] = fill;
                }
            }
        }
    };

    make_tbl(0);
    make_tbl(1);
    // {mean, rstd} of this lane's channel for the chunk about to be computed.  They are plain
    // global loads, so they are issued after a chunk's MFMA loop and complete at the next
    // barrier together with the DMA (a wait on them inside the loop would drain the DMA early).
    float mean_n = 0.0f, rstd_n = 1.0f;
    auto fetch_stats = [&](int ci) {
        if (norm && ci < nchunks) {
            const long pl = (long)(b_begin + ci / npc) * g.Cin + c0 + crow;
            mean_n = stats[2 * pl];
            rstd_n = stats[2 * pl + 1];
        }
    };
    fetch_stats(0);
    __syncthreads();
    issue(0);
    for (int ci = 0; ci < nchunks; ++ci) {
        __syncthreads();            // chunk ci has landed; buffers and table of ci+1 are free / ready
        issue(ci + 1);
        const int j = ci % npc, p0 = j * ptw;
        int a0, nrows;
        chunk_rows(j, a0, nrows);
        const float mean = mean_n, rstd = rstd_n;
        const float* ga = gl0 + (ci & 1) * gbuf + (wn * 32 + l32) * GS + h;
        const float* sb = slab0 + (ci & 1) * sbuf + crow * g.slab_stride;
        const int count = min(ptw, P - p0);
        const int npairs = (count + 1) >> 1;
        // grid pixel (p0 + 2k + h) walks the chunk two pixels per step; its slab offset moves by
        // 2*S inside a grid row and jumps by row_jump when the column wraps
        const int pix = min(p0 + h, P - 1);
        const int arow = fdiv(pix, g.m_sw);
        int bb = pix - arow * g.SW;
        int off = (arow - a0) * g.S * g.IWp + bb * g.S;
        const int row_jump = g.S * g.IWp - g.SW * g.S;
        const int off_max = (nrows - g.KHe) * g.IWp + (g.SW - 1) * g.S;   // last valid grid pixel of the chunk
        const int t1 = g.IWp, t2 = 2 * g.IWp;
        float av0, bv0[KKT], av1, bv1[KKT];
        // load_k only issues LDS reads (nothing consumes them), so the wait for step k+1's operands
        // lands after step k's MFMAs have been issued
        auto load_k = [&](int k, float& av, float (&bv)[KKT]) {
            av = ga[2 * k];
            const float* s0p = sb + min(off, off_max);
            bv[0] = s0p[0]; bv[1] = s0p[1]; bv[2] = s0p[2];
            bv[3] = s0p[t1]; bv[4] = s0p[t1 + 1]; bv[5] = s0p[t1 + 2];
            bv[6] = s0p[t2]; bv[7] = s0p[t2 + 1]; bv[8] = s0p[t2 + 2];
            bb += 2;
            off += 2 * g.S;
            if (bb >= g.SW) {
                bb -= g.SW;
                off += row_jump;
            }
        };
        auto mma_k = [&](int k, float av, float (&bv)[KKT]) {
            av = (2 * k + h < count) ? av : 0.0f;          // past the end of the plane: zero gradient
            if (norm) {
#pragma unroll
                for (int t = 0; t < KKT; ++t) {
                    float v = (bv[t] - mean) * rstd;
                    if (g.relu_in) v = fmaxf(v, 0.0f);
                    else v = (v != v) ? 0.0f : v;
                    bv[t] = v;
                }
            } else if (g.relu_in) {
#pragma unroll
                for (int t = 0; t < KKT; ++t) bv[t] = fmaxf(bv[t], 0.0f);
            }
#pragma unroll
            for (int t = 0; t < KKT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
        };
        load_k(0, av0, bv0);
        for (int k = 0; k < npairs; k += 2) {
            if (k + 1 < npairs) load_k(k + 1, av1, bv1);
            mma_k(k, av0, bv0);
            if (k + 1 < npairs) {
                if (k + 2 < npairs) load_k(k + 2, av0, bv0);
                mma_k(k + 1, av1, bv1);
            }
        }
        make_tbl(ci + 2);           // its buffer (ci & 1) was last read by issue(ci), before this chunk's barrier
        fetch_stats(ci + 1);
    }
    // partial[z][tap][n][c]
    float* dst = partial + (long)blockIdx.z * KKT * g.Cout * g.Cin;
    const int c = c0 + wc * 32 + l32;
    if (c < g.Cin) {
#pragma unroll
        for (int t = 0; t < KKT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = n0 + wn * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                if (n < g.Cout) dst[((long)t * g.Cout + n) * g.Cin + c] = acc[t][j];
            }
    }
}

// Weight gradient of the 3x3 / stride 1 / reflect-pad-1 convolutions (the ten residual-block convs,
// function/VAE_CNN_NN.py:86-93) with "flat" staging: the input rows a chunk needs are one contiguous
// run of the plane, and so is the chunk of every grad_out row, so both tiles are copied by 16-byte
// LDS-DMA from 16-byte aligned source addresses (a quarter of the DMA instructions of the 4-byte
// gather).  Nothing is padded in LDS: the reflected row / column of a border pixel is resolved in the
// B operand's address (three row offsets + three column offsets per K step).  A plane whose size is
// not a multiple of 4 floats starts at a 4-byte phase that differs per channel; the phase is added to
// the lane's base address.
#ifndef WGRAD_FLAT_LDS
#define WGRAD_FLAT_LDS (80 * 1024)
#endif
template <bool NORM>
__global__ __launch_bounds__(WG_NT, 2) void conv_wgrad_flat_kernel(int B, int Cin, int Cout, int H, int W,
                                                                   const float* __restrict__ x,
                                                                   const float* __restrict__ stats, int relu_in,
                                                                   const float* __restrict__ gout, int nr, int GS,
                                                                   int SS, int samples_per_group,
                                                                   float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KKT = 9;
    const int gbuf = 64 * GS, sbuf = 64 * SS;        // GS, SS are multiples of 4
    float* gl0 = smem;                               // [2][64][GS]
    float* slab0 = smem + 2 * gbuf;                  // [2][64][SS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int wn = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int P = H * W;
    const int b_begin = blockIdx.z * samples_per_group;
    const int b_end = min(B, b_begin + samples_per_group);
    const int npc = (H + nr - 1) / nr;               // chunks per plane
    const int nchunks = (b_end - b_begin) * npc;
    const unsigned m_g4 = magic_of(GS >> 2), m_s4 = magic_of(SS >> 2), m_w = magic_of(W);

    f32x16 acc[KKT];
#pragma unroll
    for (int t = 0; t < KKT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.0f;
    // lanes of a channel / output row that does not exist read row 0 (their products are discarded)
    const int crow = (c0 + wc * 32 + l32 < Cin) ? wc * 32 + l32 : 0;
    const int nrow = (n0 + wn * 32 + l32 < Cout) ? wn * 32 + l32 : 0;

    auto issue = [&](int ci) {                       // DMA of chunk ci into buffer ci & 1
        if (ci >= nchunks) return;
        const int b = b_begin + ci / npc, j = ci % npc;
        const int a0 = j * nr, nrows = min(nr, H - a0);
        const int p0 = a0 * W, count = nrows * W;
        float* gdst = gl0 + (ci & 1) * gbuf;
        for (int base = 0; base < (gbuf >> 2); base += WG_NT) {
            const int i4 = base + tid;
            if (i4 < (gbuf >> 2)) {
                const int n = fdiv(i4, m_g4), q = i4 - n * (GS >> 2);
                const long src = ((long)b * Cout + n0 + n) * P + p0;          // first float of the row's chunk
                const int ph = (int)(src & 3);
                if (n0 + n < Cout && 4 * q < ph + count)
                    __builtin_amdgcn_global_load_lds(MV_GLOBAL_PTR(gout + (src - ph) + 4 * q),
                                                     MV_LDS_PTR(gdst + (base + wave * 64) * 4), 16, 0, 0);
            }
        }
        const int r_lo = max(a0 - 1, 0), r_hi = min(a0 + nrows, H - 1);
        const int span = (r_hi - r_lo + 1) * W;
        float* sdst = slab0 + (ci & 1) * sbuf;
        for (int base = 0; base < (sbuf >> 2); base += WG_NT) {
            const int i4 = base + tid;
            if (i4 < (sbuf >> 2)) {
                const int c = fdiv(i4, m_s4), q = i4 - c * (SS >> 2);
                const long src = ((long)b * Cin + c0 + c) * P + r_lo * W;
                const int ph = (int)(src & 3);
                if (c0 + c < Cin && 4 * q < ph + span)
                    __builtin_amdgcn_global_load_lds(MV_GLOBAL_PTR(x + (src - ph) + 4 * q),
                                                     MV_LDS_PTR(sdst + (base + wave * 64) * 4), 16, 0, 0);
            }
        }
    };

    issue(0);
    for (int ci = 0; ci < nchunks; ++ci) {
        __syncthreads();            // chunk ci has landed (the barrier drains the DMA); the other buffers are free
        issue(ci + 1);
        const int b = b_begin + ci / npc, j = ci % npc;
        const int a0 = j * nr, nrows = min(nr, H - a0);
        const int p0 = a0 * W, count = nrows * W;
        const int r_lo = max(a0 - 1, 0);
        const int phA = (int)((((long)b * Cout + n0 + nrow) * P + p0) & 3);
        const int phB = (int)((((long)b * Cin + c0 + crow) * P + r_lo * W) & 3);
        const float* ga = gl0 + (ci & 1) * gbuf + nrow * GS + phA + h;
        const float* sb = slab0 + (ci & 1) * sbuf + crow * SS + phB - r_lo * W;   // sb[iy*W + ix]
        const int npairs = (count + 1) >> 1;
        float mean = 0.0f, rstd = 1.0f;                  // the producer's instance norm (+ ReLU) of this lane's channel
        if (NORM && stats) {
            mean = stats[2 * ((long)b * Cin + c0 + crow)];
            rstd = stats[2 * ((long)b * Cin + c0 + crow) + 1];
        }
        // grid pixel p0 + 2k + h walks the chunk two pixels per step
        int oy = a0 + fdiv(min(h, count - 1), m_w), ox = min(h, count - 1) - (oy - a0) * W;
        float av0, bv0[KKT], av1, bv1[KKT];
        // load_k only issues LDS reads (nothing consumes them), so the wait for step k+1's operands
        // lands after step k's MFMAs have been issued
        auto load_k = [&](int k, float& av, float (&bv)[KKT]) {
            av = ga[2 * k];
            const int oyc = min(oy, a0 + nrows - 1);                 // a step past the chunk's end re-reads its last row
            const int y0 = (oyc == 0 ? 1 : oyc - 1) * W, y1 = oyc * W, y2 = (oyc == H - 1 ? H - 2 : oyc + 1) * W;
            const int x0 = (ox == 0 ? 1 : ox - 1), x2 = (ox == W - 1 ? W - 2 : ox + 1);
            bv[0] = sb[y0 + x0]; bv[1] = sb[y0 + ox]; bv[2] = sb[y0 + x2];
            bv[3] = sb[y1 + x0]; bv[4] = sb[y1 + ox]; bv[5] = sb[y1 + x2];
            bv[6] = sb[y2 + x0]; bv[7] = sb[y2 + ox]; bv[8] = sb[y2 + x2];
            ox += 2;
            if (ox >= W) {
                ox -= W;
                ++oy;
            }
        };
        auto mma_k = [&](int k, float av, float (&bv)[KKT]) {
            av = (2 * k + h < count) ? av : 0.0f;          // past the end of the chunk: zero gradient
            if (NORM) {
#pragma unroll
                for (int t = 0; t < KKT; ++t) {
                    const float v = (bv[t] - mean) * rstd;
                    bv[t] = relu_in ? fmaxf(v, 0.0f) : v;
                }
            }
            if (2 * k + h >= count) {
#pragma unroll
                for (int t = 0; t < KKT; ++t) bv[t] = 0.0f;    // (whatever was read there may be NaN)
            }
#pragma unroll
            for (int t = 0; t < KKT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
        };
        load_k(0, av0, bv0);
        for (int k = 0; k < npairs; k += 2) {
            if (k + 1 < npairs) load_k(k + 1, av1, bv1);
            mma_k(k, av0, bv0);
            if (k + 1 < npairs) {
                if (k + 2 < npairs) load_k(k + 2, av0, bv0);
                mma_k(k + 1, av1, bv1);
            }
        }
    }
    // partial[z][tap][n][c]
    float* dst = partial + (long)blockIdx.z * KKT * Cout * Cin;
    const int c = c0 + wc * 32 + l32;
    if (c < Cin) {
#pragma unroll
        for (int t = 0; t < KKT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = n0 + wn * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                if (n < Cout) dst[((long)t * Cout + n) * Cin + c] = acc[t][j];
            }
    }
}

// ---------------------------------------------------------------------------------------
// weight gradient on the bf16 matrix cores
// ---------------------------------------------------------------------------------------
// D[n][c] per tap accumulates over output pixels, 16 per v_mfma_f32_32x32x16_bf16: a lane's 8 k values
// are 8 consecutive pixels of one output row.  A = grad_out[n][pixels] is that run as it lies in memory.
// B = input[c][pixel*S + tap] is the same run shifted by the tap column dx.  Input rows are staged as S
// planes per row (column phase = column mod S, so a stride-2 layer keeps even and odd columns apart):
// tap dx reads plane dx mod S shifted by dx / S elements, and a shifted fragment is built in registers
// from the aligned 16-byte piece and the first dword of the next one (v_alignbit for an odd shift, plain
// register renaming for a shift of two), so every LDS read is aligned and the input is staged once.
// Rows are padded to a multiple of 8 pixels with zeros in grad_out.  Both tiles are split / rounded to
// PARTS bf16 parts while staged (through registers: padding, reflection and the producer's instance
// norm + ReLU are applied there); the part products kept are those of conv_bf16_kernel.
// Channel strides in LDS are an odd number of 16-byte units, which keeps the fragment reads conflict-free.
// Workgroup tile: 64 output x 64 input channels x 9 taps; 4 waves as 2 (n) x 2 (c); chunk = nr output
// rows of one sample; LDS is single buffered, a second resident workgroup covers the staging.
// partial [G][9][Cout][Cin], summed by conv_wgrad_reduce_kernel.
struct WgradBf16 {
    int B, Cin, Cout, H, W, OH, OW, pad, reflect, relu_in;
    int nr;        // output rows per chunk
    int G8;        // 8-pixel groups per output row (OW padded)
    int rows;      // input rows per chunk = (nr - 1) * S + 3
    int gunits;    // 16-byte units per grad_out channel in LDS (odd)
    int sunits;    // 16-byte units per input channel in LDS (odd): rows x ru
    int ru;        // 16-byte units per staged input row: S planes x (G8 + 1), plus one where that keeps row strides odd
    int spg;       // samples per z group
};
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the fragment that starts `shift` bf16 elements into (p0, next dword)
template <int SHIFT>
__device__ __forceinline__ bf16x8 frag_shift(u32x4 p0, unsigned nxt) {
    u32x4 o;
    if (SHIFT == 0) o = p0;
    else if (SHIFT == 1) {
        o.x = __builtin_amdgcn_alignbit(p0.y, p0.x, 16);
        o.y = __builtin_amdgcn_alignbit(p0.z, p0.y, 16);
        o.z = __builtin_amdgcn_alignbit(p0.w, p0.z, 16);
        o.w = __builtin_amdgcn_alignbit(nxt, p0.w, 16);
    } else {
        o.x = p0.y; o.y = p0.z; o.z = p0.w; o.w = nxt;
    }
    return __builtin_bit_cast(bf16x8, o);
}

// RW > 0: the stride-1 reflect-pad-1 layers whose planes are RW x RW (the residual blocks: 22 at N = 87, 17 at
// N = 68) stage whole rows per thread - RW loads (8 bytes each when RW is even), and the padded row of 32
// positions (reflected border columns, zeros behind) is assembled at compile-time register indices, so staging
// costs a handful of instructions per element instead of an index computation per element.
template <int S, int PARTS, int RW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(WgradBf16 p, const float* __restrict__ x,
                                                                  const float* __restrict__ stats,
                                                                  const float* __restrict__ gout,
                                                                  float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    constexpr int NTERM = PARTS == 1 ? 1 : PARTS == 2 ? 3 : 6;
    constexpr int TA[6] = {2, 1, 0, PARTS >= 2 ? 1 : 0, 0, 0};
    constexpr int TB[6] = {0, 1, 2, 0, PARTS >= 2 ? 1 : 0, 0};
    const int gpart = 64 * p.gunits * 16, spart = 64 * p.sunits * 16;
    char* gl = smem_w;                                   // [PARTS][64 n][gunits][16 B]
    char* slab = smem_w + PARTS * gpart;                 // [PARTS][64 c][rows][S planes][G8 + 1][16 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int wn = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int b_begin = blockIdx.z * p.spg, b_end = min(p.B, b_begin + p.spg);
    const int npc = (p.OH + p.nr - 1) / p.nr;
    const int nchunks = (b_end - b_begin) * npc;
    const long in_plane = (long)p.H * p.W, out_plane = (long)p.OH * p.OW;
    const int RU = p.ru;                                 // units per staged input row
    const unsigned m_g8 = magic_of(p.G8), m_gch = magic_of(p.nr * p.G8), m_sch = magic_of(p.rows * RU), m_ru = magic_of(RU),
                   m_g9 = magic_of(p.G8 + 1), m_rows = magic_of(p.rows), m_nr = magic_of(p.nr);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.0f;

    const int gitems = 64 * p.nr * p.G8, sitems = 64 * p.rows * RU;
    for (int ci = 0; ci < nchunks; ++ci) {
        const int b = b_begin + ci / npc, a0 = (ci % npc) * p.nr;
        __syncthreads();                                 // the previous chunk's operand reads are done
        if constexpr (RW > 0) {
            static_assert(S == 1 && RW >= 8 && RW <= 24, "row staging: stride 1, planes of 8..24 columns");
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            auto load_row = [&](const float* src, float (&v)[RW]) {
                if (RW % 2 == 0) {                       // rows of an even plane start 8-byte aligned
#pragma unroll
                    for (int j = 0; j < RW / 2; ++j) {
                        const f32x2 t = *reinterpret_cast<const f32x2*>(src + 2 * j);
                        v[2 * j] = t.x;
                        v[2 * j + 1] = t.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < RW; ++j) v[j] = src[j];
                }
            };
            auto store_units = [&](const float (&pr)[32], int nunits, char* dst, int part_stride) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u >= nunits) break;
                    bf16x8 o[PARTS];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        __bf16 bp[PARTS];
                        bf_split<PARTS>(pr[8 * u + j], bp);
#pragma unroll
                        for (int pt = 0; pt < PARTS; ++pt) o[pt][j] = bp[pt];
                    }
#pragma unroll
                    for (int pt = 0; pt < PARTS; ++pt) *reinterpret_cast<bf16x8*>(dst + pt * part_stride + u * 16) = o[pt];
                }
            };
            for (int it = tid; it < 64 * p.nr; it += 256) {          // grad_out rows: (n, output row)
                const int n = p.nr == 1 ? it : fdiv(it, m_nr), r = it - n * p.nr;      // (the magic of 1 does not exist)
                const bool ok = n0 + n < p.Cout && a0 + r < p.OH;
                float v[RW], pr[32];
                load_row(gout + ((long)b * p.Cout + (ok ? n0 + n : 0)) * out_plane + (long)(ok ? a0 + r : 0) * RW, v);
#pragma unroll
                for (int j = 0; j < 32; ++j) pr[j] = (j < RW && ok) ? v[j < RW ? j : 0] : 0.0f;
                store_units(pr, p.G8, gl + (n * p.gunits + r * p.G8) * 16, gpart);
            }
            for (int it = tid; it < 64 * p.rows; it += 256) {        // input rows: (c, input row), reflected
                const int c = fdiv(it, m_rows), r = it - c * p.rows;
                int vy = a0 + r - 1;
                vy = vy < 0 ? -vy : (vy >= RW ? 2 * (RW - 1) - vy : vy);
                const bool rok = c0 + c < p.Cin && vy >= 0;           // rows far past the plane pair with zero grad_out only
                const long pl = (long)b * p.Cin + (rok ? c0 + c : 0);
                float v[RW], pr[32];
                load_row(x + pl * in_plane + (long)(rok ? vy : 0) * RW, v);
                if (stats) {
                    const float mean = stats[2 * pl], rstd = stats[2 * pl + 1];
#pragma unroll
                    for (int j = 0; j < RW; ++j) v[j] = (v[j] - mean) * rstd;
                }
                if (p.relu_in) {
#pragma unroll
                    for (int j = 0; j < RW; ++j) v[j] = fmaxf(v[j], 0.0f);
                }
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    // padded column j holds x[reflect(j - 1)]; past column RW + 1 nothing
                    const int sx = j == 0 ? 1 : (j <= RW ? j - 1 : (j == RW + 1 ? RW - 2 : -1));
                    pr[j] = (sx >= 0 && rok) ? v[sx >= 0 ? sx : 0] : 0.0f;
                }
                store_units(pr, p.G8 + 1, slab + (c * p.sunits + r * RU) * 16, spart);
            }
        } else {
        // Staging runs U items at a time: their 8 loads each are all issued (unconditionally, clamped into the
        // plane, the value dropped where there is no pixel) before the first is converted.
        constexpr int U = 4;
        for (int it0 = tid; it0 < gitems; it0 += 256 * U) {     // grad_out: (n, row, group)
            float v[U][8];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = min(it0 + u * 256, gitems - 1);
                const int n = fdiv(it, m_gch), rg = it - n * p.nr * p.G8;
                const int r = fdiv(rg, m_g8), g8 = rg - r * p.G8;
                const bool ok = n0 + n < p.Cout && a0 + r < p.OH;
                const float* src = gout + ((long)b * p.Cout + (ok ? n0 + n : 0)) * out_plane + (long)(ok ? a0 + r : 0) * p.OW;
                dst[u] = it0 + u * 256 < gitems ? (n * p.gunits + rg) * 16 : -1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ow = 8 * g8 + j;
                    const float t = src[min(ow, p.OW - 1)];
                    v[u][j] = (ok && ow < p.OW) ? t : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (dst[u] < 0) continue;
                bf16x8 o[PARTS];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 bp[PARTS];
                    bf_split<PARTS>(v[u][j], bp);
#pragma unroll
                    for (int pt = 0; pt < PARTS; ++pt) o[pt][j] = bp[pt];
                }
#pragma unroll
                for (int pt = 0; pt < PARTS; ++pt) *reinterpret_cast<bf16x8*>(gl + pt * gpart + dst[u]) = o[pt];
            }
        }
        for (int it0 = tid; it0 < sitems; it0 += 256 * U) {     // input: (c, input row, plane, unit)
            float v[U][8];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = min(it0 + u * 256, sitems - 1);
                const int c = fdiv(it, m_sch), ru = it - c * p.rows * RU;
                const int r = fdiv(ru, m_ru), uu = ru - r * RU;
                const int ph = fdiv(uu, m_g9), gu = uu - ph * (p.G8 + 1);
                int vy = a0 * S + r - p.pad;
                bool rok = c0 + c < p.Cin;
                if (p.reflect) vy = vy < 0 ? -vy : (vy >= p.H ? 2 * (p.H - 1) - vy : vy);
                rok = rok && vy >= 0 && vy < p.H;         // zero padding, and rows past the plane (a short last chunk)
                const long pl = (long)b * p.Cin + (rok ? c0 + c : 0);
                const float* src = x + pl * in_plane + (long)(rok ? vy : 0) * p.W;
                float mean = 0.0f, rstd = 1.0f;
                if (stats) {
                    mean = stats[2 * pl];
                    rstd = stats[2 * pl + 1];
                }
                dst[u] = it0 + u * 256 < sitems ? (c * p.sunits + ru) * 16 : -1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int vx = (8 * gu + j) * S + ph - p.pad;
                    if (p.reflect) vx = vx < 0 ? -vx : (vx >= p.W ? 2 * (p.W - 1) - vx : vx);
                    const bool ok = rok && vx >= 0 && vx < p.W;
                    float t = src[min(max(vx, 0), p.W - 1)];
                    if (stats) t = (t - mean) * rstd;
                    if (p.relu_in) t = fmaxf(t, 0.0f);
                    v[u][j] = ok ? t : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (dst[u] < 0) continue;
                bf16x8 o[PARTS];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 bp[PARTS];
                    bf_split<PARTS>(v[u][j], bp);
#pragma unroll
                    for (int pt = 0; pt < PARTS; ++pt) o[pt][j] = bp[pt];
                }
#pragma unroll
                for (int pt = 0; pt < PARTS; ++pt) *reinterpret_cast<bf16x8*>(slab + pt * spart + dst[u]) = o[pt];
            }
        }
        }
        __syncthreads();
        const char* ga = gl + (wn * 32 + l32) * p.gunits * 16;
        const char* sb = slab + (wc * 32 + l32) * p.sunits * 16;
        const int ngroups = p.nr * p.G8, nsteps = (ngroups + 1) >> 1;
        int r = fdiv(h, m_g8), g8 = h - r * p.G8;         // this lane's group 2*ks + h
        for (int ks = 0; ks < nsteps; ++ks) {
            // an odd number of groups: the last step's upper half has none - zero A, and B from group 0
            const bool live = 2 * ks + h < ngroups;
            if (!live) r = g8 = 0;
            bf16x8 A[PARTS];
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) {
                u32x4 a = *reinterpret_cast<const u32x4*>(ga + pt * gpart + (live ? 2 * ks + h : 0) * 16);
                if (!live) a = u32x4{0u, 0u, 0u, 0u};
                A[pt] = __builtin_bit_cast(bf16x8, a);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const char* row = sb + ((r * S + dy) * RU + g8) * 16;
                bf16x8 Bv[3][PARTS];
#pragma unroll
                for (int pt = 0; pt < PARTS; ++pt) {
                    const char* rp = row + pt * spart;
                    const u32x4 e0 = *reinterpret_cast<const u32x4*>(rp);
                    const unsigned e1 = *reinterpret_cast<const unsigned*>(rp + 16);
                    if (S == 1) {
                        Bv[0][pt] = frag_shift<0>(e0, e1);
                        Bv[1][pt] = frag_shift<1>(e0, e1);
                        Bv[2][pt] = frag_shift<2>(e0, e1);
                    } else {
                        Bv[0][pt] = frag_shift<0>(e0, e1);
                        Bv[1][pt] = *reinterpret_cast<const bf16x8*>(rp + (p.G8 + 1) * 16);
                        Bv[2][pt] = frag_shift<1>(e0, e1);
                    }
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int e = 6 - NTERM; e < 6; ++e)
                        acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[TA[e]], Bv[dx][TB[e]], acc[dy * 3 + dx], 0, 0, 0);
            }
            g8 += 2;
            if (g8 >= p.G8) { g8 -= p.G8; ++r; }
            if (g8 >= p.G8) { g8 -= p.G8; ++r; }
        }
    }
    float* dst = partial + (long)blockIdx.z * 9 * p.Cout * p.Cin;
    const int c = c0 + wc * 32 + l32;
    if (c < p.Cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = n0 + wn * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                if (n < p.Cout) dst[((long)t * p.Cout + n) * p.Cin + c] = acc[t][j];
            }
    }
}

// dW[n][c][tap] = sum_z partial[z][tap][n][c]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int G, int KK,
                                                                int Cout, int Cin, float* __restrict__ dw) {
    const long total = (long)KK * Cout * Cin;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int z = 0;
    for (; z + 3 < G; z += 4) {
        s0 += partial[(long)z * total + idx];
        s1 += partial[(long)(z + 1) * total + idx];
        s2 += partial[(long)(z + 2) * total + idx];
        s3 += partial[(long)(z + 3) * total + idx];
    }
    for (; z < G; ++z) s0 += partial[(long)z * total + idx];
    const int c = (int)(idx % Cin);
    const long t = idx / Cin;
    const int n = (int)(t % Cout), tap = (int)(t / Cout);
    dw[((long)n * Cin + c) * KK + tap] = (s0 + s1) + (s2 + s3);
}

// Stem weight gradient (one input channel): D[n][tap] accumulates over pixels.
// 4 waves split the pixels of each chunk; their accumulators are added through LDS at the end.
// partial [G][Cout][KK]
// BF16: both operands rounded to bf16 as they leave LDS, v_mfma_f32_32x32x16_bf16 over 16 consecutive pixels of the chunk
// (an eighth of the matrix instructions of the fp32 form, whose K is 2 pixels).
// G16 (with BF16): the output gradient arrives as bf16 planes `gpitch` entries apart (a multiple of 8): a thread fetches 16
// aligned bytes = 8 pixels of a channel, the tile sits in LDS as bf16 [64][ptw + 8] and an A fragment is one 16-byte read.
template <bool BF16, bool G16>
__global__ __launch_bounds__(256) void conv_stem_wgrad_kernel(ConvGeom g, int KS, const float* __restrict__ x,
                                                              const void* __restrict__ gout_any, int gpitch, int ptw,
                                                              int samples_per_group, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(BF16 || !G16, "bf16 gradient planes feed the bf16 product");
    const float* gout = static_cast<const float*>(gout_any);
    const __bf16* gout16 = static_cast<const __bf16*>(gout_any);
    const int KK = KS * KS;
    const int gl_stride = G16 ? (ptw + 8) / 2 : ptw + 1;       // in floats
    float* gl = smem;                          // [64][ptw+1] fp32, or [64][ptw+8] bf16
    float* slab = smem + 64 * gl_stride;       // [slab_stride] (+ one zero word at the end)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int P = g.SH * g.SW;
    const int b_begin = blockIdx.z * samples_per_group;
    const int b_end = min(g.B, b_begin + samples_per_group);
    // per-lane tap offsets for the two tap tiles (taps >= KK read the zero word)
    int toff[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int tap = jt * 32 + l32;
        const int ky = tap / KS, kx = tap - ky * KS;
        toff[jt] = (tap < KK) ? ky * g.IWp + kx : -1;
    }
    const int zero_at = g.slab_stride - 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][jt][j] = 0.0f;

    // The output-gradient tile of a chunk (Cout planes x ptw pixels) is fetched into registers while the previous chunk's
    // MFMAs run, and goes to LDS at the top of its own chunk; channels >= Cout and pixels behind the plane stay zero.
    constexpr int GE = G16 ? 1 : 64;                        // tile rows; thread t holds pixel t of every row (ptw = 256)
    float gv[GE];
    uint4 gq[G16 ? 8 : 1];                                  // G16: pass e holds pixels 8 (t & 31) .. + 7 of row 8 e + (t >> 5)
    const int nper = (P + ptw - 1) / ptw;
    const int nchunks = (b_end - b_begin) * nper;
    auto fetch = [&](int ch) {
        const int b = b_begin + ch / nper, p0 = (ch % nper) * ptw;
        if constexpr (G16) {
            const int pp = p0 + 8 * (tid & 31);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int n = 8 * e + (tid >> 5);
                const bool ok = n < g.Cout && pp < P;
                uint4 t = *reinterpret_cast<const uint4*>(gout16 + ((long)b * g.Cout + (ok ? n : 0)) * gpitch + (ok ? pp : 0));
                // pixels behind the plane (the pitch's slack holds anything) and rows behind Cout are zero
                const int left = ok ? P - pp : 0;               // valid pixels of this piece
                unsigned* w = reinterpret_cast<unsigned*>(&t);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (2 * j >= left) w[j] = 0u;
                    else if (2 * j + 1 >= left) w[j] &= 0xffffu;
                }
                gq[e] = t;
            }
        } else {
#pragma unroll
            for (int e = 0; e < GE; ++e) {
                const int n = e, pp = tid;                      // ptw = 256 = the threads of the workgroup
                const bool ok = n < g.Cout && p0 + pp < P;
                const float t = gout[((long)b * g.Cout + (ok ? n : 0)) * P + (ok ? p0 + pp : 0)];
                gv[e] = ok ? t : 0.0f;
            }
        }
    };
    if (nchunks > 0) fetch(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int b = b_begin + ch / nper, p0 = (ch % nper) * ptw;
        const int a0 = fdiv(p0, g.m_sw);
        const int plast = min(P - 1, p0 + ptw - 1);
        const int nrows = (fdiv(plast, g.m_sw) - a0) * g.S + g.KHe;
        const int pr0 = a0 * g.S;
        const int slab_elems = nrows * g.IWp;
        __syncthreads();
        if constexpr (G16) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<uint4*>(gl + (8 * e + (tid >> 5)) * gl_stride + 4 * (tid & 31)) = gq[e];
        } else {
#pragma unroll
            for (int e = 0; e < GE; ++e) {
                gl[e * gl_stride + tid] = gv[e];
            }
        }
        stage_slab<256, 8>(g, x, nullptr, b, 0, 1, 1, pr0, slab_elems, magic_of(slab_elems), slab, tid);
        if (tid == 0) slab[zero_at] = 0.0f;
        __syncthreads();
        if (ch + 1 < nchunks) fetch(ch + 1);
        if constexpr (BF16) {
            // wave w takes the 16-pixel steps w, w+4, ...; lane half h holds pixels 8h .. 8h+7 of a step: of channel
            // (32 i + l32) for A, shifted by tap (32 jt + l32) for B
            const int nsteps = (min(ptw, P - p0) + 15) >> 4;
            for (int ks = wave; ks < nsteps; ks += 4) {
                const int pb = 16 * ks + 8 * h;
                bf16x8 A[2], Bv[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if constexpr (G16) {
                        A[i] = *reinterpret_cast<const bf16x8*>(gl + (i * 32 + l32) * gl_stride + pb / 2);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) A[i][j] = (__bf16)gl[(i * 32 + l32) * gl_stride + pb + j];
                    }
                }
                int pix = min(P - 1, p0 + pb);
                int a = fdiv(pix, g.m_sw), bb = pix - a * g.SW;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int off = (a - a0) * g.S * g.IWp + bb * g.S;
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt) Bv[jt][j] = (__bf16)slab[toff[jt] >= 0 ? off + toff[jt] : zero_at];
                    if (p0 + pb + j + 1 < P) {                  // (pixels behind the plane: A holds zeros there)
                        if (++bb == g.SW) { bb = 0; ++a; }
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt)
                        acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], Bv[jt], acc[i][jt], 0, 0, 0);
            }
            continue;
        }
        // wave w takes pixel pairs w, w+4, w+8, ...
        const int npairs = (min(ptw, P - p0) + 1) >> 1;
#pragma unroll 2
        for (int k = wave; k < npairs; k += 4) {
            const int pix = min(P - 1, p0 + 2 * k + h);
            const int a = fdiv(pix, g.m_sw), bb = pix - a * g.SW;
            const int off = (a - a0) * g.S * g.IWp + bb * g.S;
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = gl[(i * 32 + l32) * gl_stride + 2 * k + h];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) bv[jt] = slab[toff[jt] >= 0 ? off + toff[jt] : zero_at];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
                    acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[jt], acc[i][jt], 0, 0, 0);
        }
    }
    // add the four waves' tiles: red[wave][n 64][tap 64]
    __syncthreads();
    float* red = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = i * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                red[(wave * 64 + n) * 65 + jt * 32 + l32] = acc[i][jt][j];
            }
    __syncthreads();
    float* dst = partial + (long)blockIdx.z * g.Cout * KK;
    for (int idx = tid; idx < g.Cout * KK; idx += 256) {
        const int n = idx / KK, tap = idx - n * KK;
        dst[idx] = (red[(0 * 64 + n) * 65 + tap] + red[(1 * 64 + n) * 65 + tap]) +
                   (red[(2 * 64 + n) * 65 + tap] + red[(3 * 64 + n) * 65 + tap]);
    }
}

// out[i] = sum_z partial[z][i]: a workgroup takes 16 entries, thread t the slices z = t / 16, + 16, ... of entry t % 16
// (four running sums), and the 16 partial sums of an entry are added in a fixed order.  grid = ceil(n / 16).
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int G, long n,
                                                           float* __restrict__ out) {
    __shared__ float red[16][17];
    const int e = threadIdx.x & 15, z0 = threadIdx.x >> 4;
    const long idx = (long)blockIdx.x * 16 + e;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (idx < n) {
        int z = z0;
        for (; z + 48 < G; z += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += partial[(long)(z + 16 * u) * n + idx];
        }
        for (int u = 0; z < G; z += 16, ++u) s[u & 3] += partial[(long)z * n + idx];
    }
    red[z0][e] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (threadIdx.x < 16 && idx < n) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        out[idx] = t;
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// grid rows touched by `pt` consecutive grid pixels that start at a multiple of pt
static int rows_touched(int pt, int SW, int SH) {
    int r = (pt % SW == 0) ? pt / SW : (pt + SW - 2) / SW + 1;
    return r < SH ? r : SH;
}

static void set_lds_limit(const void* fn) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

static void finish_geom(ConvGeom& g, int KWe) {
    g.IWp = (g.SW - 1) * g.S + KWe;
    g.m_iwp = magic_of(g.IWp);
    g.m_sw = magic_of(g.SW);
}

template <int CK, int SI, int TAPS, bool PLAIN>
static void launch_conv_t(const ConvGeom& g, size_t lds, const float* x, const float* stats, const float* wp,
                          const float* bias, float* out, hipStream_t st) {
    constexpr int PT = 256, NTILE = 96;
    set_lds_limit(reinterpret_cast<const void*>(conv_mfma_kernel<CK, SI, TAPS, PLAIN>));
    const long Q = (long)g.B * g.SH * g.SW;
    dim3 grid((unsigned)((Q + PT - 1) / PT), g.CoutP / NTILE, 1);
    hipLaunchKernelGGL((conv_mfma_kernel<CK, SI, TAPS, PLAIN>), grid, dim3(256), lds, st, g, x, stats, wp, bias, out);
}

// One grid over all (sample, grid pixel) rows of the launch.
static int launch_conv(ConvGeom g, const float* x, const float* stats, const float* wp, const float* bias, float* out,
                       hipStream_t st) {
    const int P = g.SH * g.SW;
    constexpr int PT = 256;
    MV_REQUIRE(P >= PT, E_SHAPE, "mdnvae_conv2d: plane of %d pixels is smaller than the %d-pixel tile", P, PT);
    // slab rows of a 256-row tile: one segment starting anywhere in a plane, or the tail of one
    // sample plus the head of the next (each segment carries its own KHe halo)
    const int one = ((PT + g.SW - 2) / g.SW) * g.S + g.KHe;
    const int two = (PT / g.SW) * g.S + 2 * g.KHe;
    const int rows = one > two ? one : two;
    g.slab_stride = rows * g.IWp;
    MV_REQUIRE(g.slab_stride <= 24 * 64, E_SHAPE, "mdnvae_conv2d: slab of %d floats per channel is too large",
               g.slab_stride);
    // 4 input channels per chunk keeps three workgroups resident per CU
    const size_t lds = ((size_t)2 * MAX_TAPS * 4 * 96 + (size_t)2 * 4 * g.slab_stride + 2 * 4 * 4) * 4;
    MV_REQUIRE(lds <= 160 * 1024, E_SHAPE, "mdnvae_conv2d: tile needs %zu bytes of LDS", lds);
    const bool plain = stats == nullptr && !g.relu_in;
    if (g.slab_stride <= 7 * 64) {
        if (g.ntaps == 9 && plain) launch_conv_t<4, 7, 9, true>(g, lds, x, stats, wp, bias, out, st);
        else if (g.ntaps == 9) launch_conv_t<4, 7, 9, false>(g, lds, x, stats, wp, bias, out, st);
        // the parity classes of a stride-2 data gradient have 1, 2, 2 and 4 taps
        else if (g.ntaps == 4 && plain) launch_conv_t<4, 7, 4, true>(g, lds, x, stats, wp, bias, out, st);
        else if (g.ntaps == 2 && plain) launch_conv_t<4, 7, 2, true>(g, lds, x, stats, wp, bias, out, st);
        else if (g.ntaps == 1 && plain) launch_conv_t<4, 7, 1, true>(g, lds, x, stats, wp, bias, out, st);
        else if (plain) launch_conv_t<4, 7, 0, true>(g, lds, x, stats, wp, bias, out, st);
        else launch_conv_t<4, 7, 0, false>(g, lds, x, stats, wp, bias, out, st);
    } else if (plain && g.ntaps == 9) {
        launch_conv_t<4, 24, 9, true>(g, lds, x, stats, wp, bias, out, st);
    } else if (plain) {
        launch_conv_t<4, 24, 0, true>(g, lds, x, stats, wp, bias, out, st);
    } else {
        launch_conv_t<4, 24, 0, false>(g, lds, x, stats, wp, bias, out, st);
    }
    MV_LAUNCH_CHECK("conv_mfma_kernel");
    return 0;
}

template <int PARTS, int WM, int WN, int IT, int TAPS, bool PLAIN>
static void launch_conv_bf16_t(const ConvGeom& g, size_t lds, const float* x, const float* stats, const char* wp,
                               const float* bias, float* out, hipStream_t st) {
    constexpr int PT = 4 * WM * 32, NTILE = WN * 32;
    set_lds_limit(reinterpret_cast<const void*>(conv_bf16_kernel<PARTS, WM, WN, IT, TAPS, PLAIN>));
    const long Q = (long)g.B * g.SH * g.SW;
    dim3 grid((unsigned)((Q + PT - 1) / PT), (g.Cout + NTILE - 1) / NTILE, 1);
    hipLaunchKernelGGL((conv_bf16_kernel<PARTS, WM, WN, IT, TAPS, PLAIN>), grid, dim3(256), lds, st, g, x, stats, wp, bias,
                       out);
}

// tile shapes per number of parts: (rows, output channels) per workgroup chosen so that two workgroups fit a CU
template <int PARTS, int WM, int WN>
static int launch_conv_bf16_p(ConvGeom g, const float* x, const float* stats, const char* wp, const float* bias,
                              float* out, hipStream_t st) {
    const int P = g.SH * g.SW;
    constexpr int PT = 4 * WM * 32;
    MV_REQUIRE(P >= PT, E_SHAPE, "mdnvae_conv2d_bf16: plane of %d pixels is smaller than the %d-pixel tile", P, PT);
    const int one = ((PT + g.SW - 2) / g.SW) * g.S + g.KHe;
    const int two = (PT / g.SW) * g.S + 2 * g.KHe;
    const int rows = one > two ? one : two;
    g.slab_stride = rows * g.IWp;
    MV_REQUIRE(g.slab_stride <= 6 * 256, E_SHAPE, "mdnvae_conv2d_bf16: slab of %d positions is too large", g.slab_stride);
    const size_t lds = (size_t)PARTS * ((size_t)MAX_TAPS * WN * 1024 + (size_t)2 * g.slab_stride * 16);
    MV_REQUIRE(lds <= 160 * 1024, E_SHAPE, "mdnvae_conv2d_bf16: tile needs %zu bytes of LDS", lds);
    const bool plain = stats == nullptr && !g.relu_in;
    if (g.slab_stride <= 2 * 256) {
        if (g.ntaps == 9 && plain) launch_conv_bf16_t<PARTS, WM, WN, 2, 9, true>(g, lds, x, stats, wp, bias, out, st);
        else if (g.ntaps == 9) launch_conv_bf16_t<PARTS, WM, WN, 2, 9, false>(g, lds, x, stats, wp, bias, out, st);
        else if (plain) launch_conv_bf16_t<PARTS, WM, WN, 2, 0, true>(g, lds, x, stats, wp, bias, out, st);
        else launch_conv_bf16_t<PARTS, WM, WN, 2, 0, false>(g, lds, x, stats, wp, bias, out, st);
    } else {
        if (g.ntaps == 9 && plain) launch_conv_bf16_t<PARTS, WM, WN, 6, 9, true>(g, lds, x, stats, wp, bias, out, st);
        else if (g.ntaps == 9) launch_conv_bf16_t<PARTS, WM, WN, 6, 9, false>(g, lds, x, stats, wp, bias, out, st);
        else if (plain) launch_conv_bf16_t<PARTS, WM, WN, 6, 0, true>(g, lds, x, stats, wp, bias, out, st);
        else launch_conv_bf16_t<PARTS, WM, WN, 6, 0, false>(g, lds, x, stats, wp, bias, out, st);
    }
    MV_LAUNCH_CHECK("conv_bf16_kernel");
    return 0;
}

static int launch_conv_bf16(const ConvGeom& g, int parts, const float* x, const float* stats, const char* wp,
                            const float* bias, float* out, hipStream_t st) {
    MV_REQUIRE(g.Cin % BF_CK == 0, E_SHAPE, "mdnvae_conv2d_bf16: Cin %d must be a multiple of 16", g.Cin);
    if (parts == 1) return launch_conv_bf16_p<1, 2, 3>(g, x, stats, wp, bias, out, st);
    if (parts == 2) return launch_conv_bf16_p<2, 2, 2>(g, x, stats, wp, bias, out, st);
    // three parts: 128-row tiles while two workgroups fit a CU (80 KiB each); else one workgroup with 256 rows,
    // whose longer MFMA phase hides more of its own staging
    const int rows128 = std::max(((128 + g.SW - 2) / g.SW) * g.S + g.KHe, (128 / g.SW) * g.S + 2 * g.KHe);
    const size_t lds128 = (size_t)3 * ((size_t)MAX_TAPS * 2 * 1024 + (size_t)2 * rows128 * g.IWp * 16);
    const int rows256 = std::max(((256 + g.SW - 2) / g.SW) * g.S + g.KHe, (256 / g.SW) * g.S + 2 * g.KHe);
    const size_t lds256 = (size_t)3 * ((size_t)MAX_TAPS * 2 * 1024 + (size_t)2 * rows256 * g.IWp * 16);
    if (lds128 <= 80 * 1024 || lds256 > 160 * 1024 || g.SH * g.SW < 256)
        return launch_conv_bf16_p<3, 1, 2>(g, x, stats, wp, bias, out, st);
    return launch_conv_bf16_p<3, 2, 2>(g, x, stats, wp, bias, out, st);
}

static int wgrad_groups(int B, int ntiles) {
    int G = 512 / ntiles;
    if (G < 1) G = 1;
    if (G > B) G = B;
    return G;
}

// weight gradient of the single-channel stem; g: geometry as mdnvae_conv2d_wgrad fills it
static int stem_wgrad_run(ConvGeom g, int KS, const float* x, const void* grad_out, float* workspace, float* grad_w, bool bf16,
                          hipStream_t st, int gpitch16 = 0) {
    const int B = g.B, Cout = g.Cout, OH = g.SH, OW = g.SW, KK = KS * KS;
    const int Gz0 = wgrad_groups(B, 1);
    const int spg = (B + Gz0 - 1) / Gz0, Gz = (B + spg - 1) / spg;
    const int ptw = 256;
    const int rows = (rows_touched(ptw, OW, OH) - 1) * g.S + KS;
    g.slab_stride = rows * g.IWp + 2;
    size_t lds = ((size_t)64 * (ptw + 1) + g.slab_stride) * 4;
    const size_t red = (size_t)4 * 64 * 65 * 4;
    if (lds < red) lds = red;
    MV_REQUIRE(lds <= 160 * 1024, E_SHAPE, "mdnvae_conv2d_wgrad: needs %zu bytes of LDS", lds);
    if (gpitch16) {
        set_lds_limit(reinterpret_cast<const void*>(conv_stem_wgrad_kernel<true, true>));
        hipLaunchKernelGGL((conv_stem_wgrad_kernel<true, true>), dim3(1, 1, Gz), dim3(256), lds, st, g, KS, x, grad_out, gpitch16, ptw, spg, workspace);
    } else if (bf16) {
        set_lds_limit(reinterpret_cast<const void*>(conv_stem_wgrad_kernel<true, false>));
        hipLaunchKernelGGL((conv_stem_wgrad_kernel<true, false>), dim3(1, 1, Gz), dim3(256), lds, st, g, KS, x, grad_out, 0, ptw, spg, workspace);
    } else {
        set_lds_limit(reinterpret_cast<const void*>(conv_stem_wgrad_kernel<false, false>));
        hipLaunchKernelGGL((conv_stem_wgrad_kernel<false, false>), dim3(1, 1, Gz), dim3(256), lds, st, g, KS, x, grad_out, 0, ptw, spg, workspace);
    }
    MV_LAUNCH_CHECK("conv_stem_wgrad_kernel");
    const long n = (long)Cout * KK;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, workspace, Gz, n, grad_w);
    MV_LAUNCH_CHECK("sum_partials_kernel");
    return 0;
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

size_t mdnvae_conv_packed_floats(int Cout, int Cin, int KS, int transposed) {
    const int KK = KS * KS;
    if (Cin == 1 && !transposed) return (size_t)((KK + 1) & ~1) * round_up(Cout, 64);
    if (transposed) return (size_t)KK * Cout * round_up(Cin, 96);
    return (size_t)KK * Cin * round_up(Cout, 96);
}

int mdnvae_conv_pack_weights(const float* w, int Cout, int Cin, int KS, float* packed_fwd, float* packed_bwd,
                             void* stream) {
    MV_REQUIRE(w && packed_fwd, E_ARG, "mdnvae_conv_pack_weights: null pointer");
    MV_REQUIRE(Cout >= 1 && Cin >= 1 && KS >= 1 && KS <= 7, E_SHAPE, "mdnvae_conv_pack_weights: bad shape");
    const int KK = KS * KS;
    if (Cin == 1) {
        const int KKP = (KK + 1) & ~1, CoutP = round_up(Cout, 64);
        hipLaunchKernelGGL(conv_pack_stem_kernel, dim3((KKP * CoutP + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                           Cout, KK, KKP, CoutP, packed_fwd);
        MV_LAUNCH_CHECK("conv_pack_stem_kernel");
        return 0;
    }
    const int CoutP = round_up(Cout, 96), CinP = round_up(Cin, 96);
    const long nf = (long)KK * Cin * CoutP, nb = packed_bwd ? (long)KK * Cout * CinP : 0;
    const long n = nf > nb ? nf : nb;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout,
                       Cin, KK, CoutP, CinP, packed_fwd, packed_bwd);
    MV_LAUNCH_CHECK("conv_pack_kernel");
    return 0;
}

}  // extern "C"

// Forward convolution (transposed = 0) or data gradient (transposed = 1); bf16: packed_w holds bf16 fragments
static int conv2d_run(const float* x, const float* in_stats, int relu_in, const void* packed_w_, const float* bias,
                      float* out, int B, int Cin, int H, int W, int Cout, int OH, int OW, int KS, int stride, int pad,
                      int reflect, int transposed, void* stream, int bf16, float* tile_stats = nullptr) {
    const float* packed_w = static_cast<const float*>(packed_w_);
    auto launch = [&](const ConvGeom& gg, const float* bb) {
        return bf16 ? launch_conv_bf16(gg, bf16, x, in_stats, static_cast<const char*>(packed_w_), bb, out, (hipStream_t)stream)
                    : launch_conv(gg, x, in_stats, packed_w, bb, out, (hipStream_t)stream);
    };
    MV_REQUIRE(!(bf16 && Cin == 1), E_SHAPE, "mdnvae_conv2d_bf16: the single-channel stem runs in fp32 (mdnvae_conv2d)");
    MV_REQUIRE(B >= 0, E_ARG, "mdnvae_conv2d: negative batch");
    if (B == 0) return 0;
    MV_REQUIRE(x && packed_w && out, E_ARG, "mdnvae_conv2d: null pointer");
    MV_REQUIRE(B <= 65535, E_SHAPE, "mdnvae_conv2d: batch %d exceeds the grid limit 65535", B);
    MV_REQUIRE(stride == 1 || stride == 2, E_SHAPE, "mdnvae_conv2d: stride %d", stride);
    MV_REQUIRE(!(reflect && (transposed || stride != 1)), E_SHAPE, "mdnvae_conv2d: reflection padding needs stride 1");
    MV_REQUIRE(Cin == 1 || Cin % 8 == 0, E_SHAPE, "mdnvae_conv2d: Cin %d must be 1 or a multiple of 8", Cin);
    MV_REQUIRE(!(Cin == 1 && (transposed || in_stats)), E_SHAPE, "mdnvae_conv2d: single-channel input is forward only");
    MV_REQUIRE((Cin == 1 && KS == 7) || (Cin > 1 && KS == 3), E_SHAPE,
               "mdnvae_conv2d: supported kernels: 7x7 on one channel, 3x3 on many (got k%d, Cin %d)", KS, Cin);
    hipStream_t st = (hipStream_t)stream;
    ConvGeom g{};
    g.B = B; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.OHf = OH; g.OWf = OW;
    g.relu_in = relu_in;
    g.osy = g.osx = 1; g.oy0 = g.ox0 = 0;
    if (!transposed) {
        MV_REQUIRE(OH == (H + 2 * pad - KS) / stride + 1 && OW == (W + 2 * pad - KS) / stride + 1, E_SHAPE,
                   "mdnvae_conv2d: output %dx%d does not match input %dx%d k%d s%d p%d", OH, OW, H, W, KS, stride, pad);
        MV_REQUIRE(!reflect || (pad < H && pad < W), E_SHAPE, "mdnvae_conv2d: reflection pad %d too large", pad);
        g.SH = OH; g.SW = OW; g.S = stride; g.pad = pad; g.reflect = reflect; g.KHe = KS;
        finish_geom(g, KS);
        if (Cin == 1) {
            constexpr int WAVES = 4, WM = 2, WN = 2, PT = WAVES * WM * 32;
            g.CoutP = round_up(Cout, 64);
            const int rows = (rows_touched(PT, g.SW, g.SH) - 1) * g.S + KS;
            g.slab_stride = rows * g.IWp + 1;
            size_t lds = ((size_t)g.slab_stride + (size_t)((KS * KS + 1) & ~1) * WN * 32) * 4;
            if (lds < (size_t)32 * (PT + 4) * 4) lds = (size_t)32 * (PT + 4) * 4;        // the write-out's [32 channels][PT + 4]
            MV_REQUIRE(lds <= 160 * 1024, E_SHAPE, "mdnvae_conv2d: tile needs %zu bytes of LDS", lds);
            set_lds_limit(reinterpret_cast<const void*>(conv_stem_kernel<WAVES, WM, WN>));
            dim3 grid((OH * OW + PT - 1) / PT, g.CoutP / (WN * 32), B);
            hipLaunchKernelGGL((conv_stem_kernel<WAVES, WM, WN>), grid, dim3(WAVES * 64), lds, st, g, KS, x, packed_w,
                               bias, out, tile_stats);
            MV_LAUNCH_CHECK("conv_stem_kernel");
            return 0;
        }
        g.CoutP = round_up(Cout, bf16 ? 192 : 96);
        g.ntaps = 9;
        for (int t = 0; t < 9; ++t) {
            g.tap_off[t] = (t / 3) * g.IWp + (t % 3);
            g.tap_w[t] = t;
        }
        return launch(g, bias);
    }
    // ---- data gradient: x = gradient w.r.t. the conv output [B,Cin,H,W], out = gradient w.r.t. its input
    MV_REQUIRE(pad >= 0 && pad <= KS - 1, E_SHAPE, "mdnvae_conv2d: transposed padding %d", pad);
    MV_REQUIRE(H == (OH + 2 * pad - KS) / stride + 1 && W == (OW + 2 * pad - KS) / stride + 1, E_SHAPE,
               "mdnvae_conv2d: gradient %dx%d is not the k%d s%d p%d image of a %dx%d plane", H, W, KS, stride, pad, OH,
               OW);
    g.CoutP = round_up(Cout, bf16 ? 192 : 96);
    g.reflect = 0;
    g.S = 1;
    if (stride == 1) {
        g.SH = OH; g.SW = OW; g.pad = KS - 1 - pad; g.KHe = KS;
        finish_geom(g, KS);
        g.ntaps = 9;
        for (int t = 0; t < 9; ++t) {
            g.tap_off[t] = (t / 3) * g.IWp + (t % 3);
            g.tap_w[t] = t;                       // the packed copy is already flipped
        }
        return launch(g, nullptr);
    }
    // stride 2, 3x3, pad 1: input pixel (2a+py, 2b+px) receives kernel rows ky with (py+1-ky) even
    MV_REQUIRE(KS == 3 && pad == 1, E_SHAPE, "mdnvae_conv2d: stride-2 data gradient supports k3 p1");
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            ConvGeom q = g;
            q.SH = (OH - py + 1) / 2; q.SW = (OW - px + 1) / 2;
            if (q.SH <= 0 || q.SW <= 0) continue;
            q.osy = q.osx = 2; q.oy0 = py; q.ox0 = px; q.pad = 0;
            q.KHe = py ? 2 : 1;
            finish_geom(q, px ? 2 : 1);
            // (ky, dy): py == 0 -> (1, 0);  py == 1 -> (0, 1), (2, 0)
            const int kys[2][2] = {{1, -1}, {0, 2}}, dys[2][2] = {{0, 0}, {1, 0}};
            q.ntaps = 0;
            for (int iy = 0; iy < (py ? 2 : 1); ++iy)
                for (int ix = 0; ix < (px ? 2 : 1); ++ix) {
                    const int ky = kys[py][iy], kx = kys[px][ix];
                    q.tap_off[q.ntaps] = dys[py][iy] * q.IWp + dys[px][ix];
                    q.tap_w[q.ntaps] = (2 - ky) * 3 + (2 - kx);      // index in the flipped packed copy
                    ++q.ntaps;
                }
            if (int rc = launch(q, nullptr)) return rc;
        }
    return 0;
}

extern "C" {

int mdnvae_conv2d(const float* x, const float* in_stats, int relu_in, const float* packed_w, const float* bias,
                  float* out, int B, int Cin, int H, int W, int Cout, int OH, int OW, int KS, int stride, int pad,
                  int reflect, int transposed, void* stream) {
    return conv2d_run(x, in_stats, relu_in, packed_w, bias, out, B, Cin, H, W, Cout, OH, OW, KS, stride, pad, reflect,
                      transposed, stream, 0);
}

size_t mdnvae_conv_stem_stats_workspace_floats(int B, int Cout, int OH, int OW) {
    return (size_t)B * Cout * ((OH * OW + 255) / 256) * 2;
}

int mdnvae_conv_stem_stats(const float* x, const float* packed_w, const float* bias, float* out, int B, int H, int W, int Cout,
                           int KS, int pad, float* workspace, float* stats, void* stream) {
    MV_REQUIRE(B >= 0 && H >= 1 && W >= 1, E_ARG, "mdnvae_conv_stem_stats: bad size");
    if (B == 0) return 0;
    MV_REQUIRE(workspace && stats, E_ARG, "mdnvae_conv_stem_stats: null pointer");
    const int OH = H + 2 * pad - KS + 1, OW = W + 2 * pad - KS + 1;
    const int rc = conv2d_run(x, nullptr, 0, packed_w, bias, out, B, 1, H, W, Cout, OH, OW, KS, 1, pad, 1, 0, stream, 0, workspace);
    if (rc) return rc;
    const long nplanes = (long)B * Cout;
    hipLaunchKernelGGL(stem_stats_finalize_kernel, dim3((unsigned)((nplanes + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       workspace, nplanes, (OH * OW + 255) / 256, OH * OW, 256, 1e-5f, stats);
    MV_LAUNCH_CHECK("stem_stats_finalize_kernel");
    return 0;
}

size_t mdnvae_conv_packed_bytes_bf16(int Cout, int Cin, int KS, int transposed, int parts) {
    if (KS != 3 || Cin % 16 || Cout % 16 || parts < 1 || parts > 3) return 0;
    return (size_t)parts * 9 * 2 * (transposed ? (size_t)Cout * round_up(Cin, 192) : (size_t)Cin * round_up(Cout, 192));
}

int mdnvae_conv_pack_weights_bf16(const float* w, int Cout, int Cin, int KS, int parts, void* packed_fwd, void* packed_bwd,
                                  void* stream) {
    MV_REQUIRE(w && packed_fwd, E_ARG, "mdnvae_conv_pack_weights_bf16: null pointer");
    MV_REQUIRE(parts >= 1 && parts <= 3, E_ARG, "mdnvae_conv_pack_weights_bf16: parts = %d (1, 2 or 3)", parts);
    MV_REQUIRE(KS == 3 && Cin >= 16 && Cin % 16 == 0 && Cout % 16 == 0, E_SHAPE,
               "mdnvae_conv_pack_weights_bf16: 3x3 kernels with channel counts that are multiples of 16 (got k%d, %d -> %d)",
               KS, Cin, Cout);
    const int CoutP = round_up(Cout, 192), CinP = round_up(Cin, 192);
    const long nf = 9L * Cin * CoutP, nb = packed_bwd ? 9L * Cout * CinP : 0;
    const long n = nf > nb ? nf : nb;
    hipLaunchKernelGGL(conv_pack_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout,
                       Cin, CoutP, CinP, parts, static_cast<__bf16*>(packed_fwd), static_cast<__bf16*>(packed_bwd));
    MV_LAUNCH_CHECK("conv_pack_bf16_kernel");
    return 0;
}

int mdnvae_conv2d_bf16(const float* x, const float* in_stats, int relu_in, const void* packed_w, const float* bias,
                       float* out, int B, int Cin, int H, int W, int Cout, int OH, int OW, int KS, int stride, int pad,
                       int reflect, int transposed, int parts, void* stream) {
    MV_REQUIRE(parts >= 1 && parts <= 3, E_ARG, "mdnvae_conv2d_bf16: parts = %d (1, 2 or 3)", parts);
    return conv2d_run(x, in_stats, relu_in, packed_w, bias, out, B, Cin, H, W, Cout, OH, OW, KS, stride, pad, reflect,
                      transposed, stream, parts);
}

size_t mdnvae_conv_wgrad_workspace_floats(int B, int Cout, int Cin, int KS) {
    if (Cin == 1) return (size_t)wgrad_groups(B, 1) * Cout * KS * KS;
    const int ntiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    return (size_t)wgrad_groups(B, ntiles) * KS * KS * Cout * Cin;
}

// dW [Cout,Cin,KS,KS] = sum over samples and pixels of grad_out x (padded, normalised) input.
int mdnvae_conv2d_wgrad(const float* x, const float* in_stats, int relu_in, const float* grad_out, float* workspace,
                        float* grad_w, int B, int Cin, int H, int W, int Cout, int OH, int OW, int KS, int stride,
                        int pad, int reflect, void* stream) {
    MV_REQUIRE(B >= 1, E_ARG, "mdnvae_conv2d_wgrad: empty batch");
    MV_REQUIRE(x && grad_out && workspace && grad_w, E_ARG, "mdnvae_conv2d_wgrad: null pointer");
    MV_REQUIRE((KS == 3 && Cin > 1) || (KS == 7 && Cin == 1), E_SHAPE,
               "mdnvae_conv2d_wgrad: supported: 3x3 multi-channel or 7x7 single-channel (got k%d Cin %d)", KS, Cin);
    MV_REQUIRE(OH == (H + 2 * pad - KS) / stride + 1 && OW == (W + 2 * pad - KS) / stride + 1, E_SHAPE,
               "mdnvae_conv2d_wgrad: output size mismatch");
    MV_REQUIRE(Cin > 1 || Cout <= 64, E_SHAPE, "mdnvae_conv2d_wgrad: single-channel input supports Cout <= 64");
    ConvGeom g{};
    g.B = B; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.OHf = OH; g.OWf = OW; g.SH = OH; g.SW = OW;
    g.osy = g.osx = 1;
    g.S = stride; g.pad = pad; g.reflect = reflect; g.relu_in = relu_in; g.KHe = KS;
    finish_geom(g, KS);
    const int KK = KS * KS;
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 1) return stem_wgrad_run(g, KS, x, grad_out, workspace, grad_w, false, st);
    const int ntiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    const int Gz0 = wgrad_groups(B, ntiles);
    const int spg = (B + Gz0 - 1) / Gz0, Gz = (B + spg - 1) / spg;
    const long total = (long)KK * Cout * Cin;
    if (KS == 3 && stride == 1 && pad == 1 && reflect && (in_stats || !relu_in) && H >= 3 && W >= 3 && W <= 64) {
        // flat 16-byte staging: whole grid rows per chunk, as many as leave room for two workgroups per CU
        int nr = 0, GS = 0, SS = 0;
        size_t lds = 0;
        for (int k = 1; k <= H && k * W <= 128; ++k) {
            const int gs = (k * W + 3 + 3) & ~3, ss = ((k + 2) * W + 3 + 3) & ~3;
            const size_t need = ((size_t)2 * 64 * gs + (size_t)2 * 64 * ss) * 4;
            if (need <= WGRAD_FLAT_LDS) { nr = k; GS = gs; SS = ss; lds = need; }
        }
        if (nr > 0) {
            set_lds_limit(reinterpret_cast<const void*>(conv_wgrad_flat_kernel<false>));
            set_lds_limit(reinterpret_cast<const void*>(conv_wgrad_flat_kernel<true>));
            dim3 grid((Cout + 63) / 64, (Cin + 63) / 64, Gz);
            if (in_stats)
                hipLaunchKernelGGL(conv_wgrad_flat_kernel<true>, grid, dim3(WG_NT), lds, st, B, Cin, Cout, H, W, x, in_stats,
                                   relu_in, grad_out, nr, GS, SS, spg, workspace);
            else
                hipLaunchKernelGGL(conv_wgrad_flat_kernel<false>, grid, dim3(WG_NT), lds, st, B, Cin, Cout, H, W, x, in_stats,
                                   relu_in, grad_out, nr, GS, SS, spg, workspace);
            MV_LAUNCH_CHECK("conv_wgrad_flat_kernel");
            hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                               workspace, Gz, KK, Cout, Cin, grad_w);
            MV_LAUNCH_CHECK("conv_wgrad_reduce_kernel");
            return 0;
        }
    }
    // pixel chunk = whole grid rows (so every chunk of a plane has the same slab shape); the
    // largest one whose double-buffered tiles leave room for two workgroups per CU, else for one
    const int cslab = Cin < 64 ? Cin : 64;
    int best_ptw = 0;
    size_t best_lds = 0;
    auto lds_of = [&](int ptw) {
        const int rows = (rows_touched(ptw, OW, OH) - 1) * stride + KS;
        const size_t ss = (size_t)((rows * g.IWp) | 1);
        return ((size_t)2 * 64 * (ptw + 1) + 2 * cslab * ss + 2 * ss) * 4;
    };
    for (size_t cap : {(size_t)80 * 1024, (size_t)160 * 1024}) {
        for (int k = 1; k * OW <= 128; ++k) {
            int ptw = k * OW;
            if (ptw & 1) continue;
            const int ss = (((rows_touched(ptw, OW, OH) - 1) * stride + KS) * g.IWp) | 1;
            if (ss <= 8 * 64 && lds_of(ptw) <= cap && ptw > best_ptw) { best_ptw = ptw; best_lds = lds_of(ptw); }
        }
        if (best_ptw) break;
    }
    MV_REQUIRE(best_ptw > 0, E_SHAPE, "mdnvae_conv2d_wgrad: no pixel chunk of whole rows fits in LDS (OW=%d)", OW);
    const int ptw = best_ptw;
    g.slab_stride = (((rows_touched(ptw, OW, OH) - 1) * stride + KS) * g.IWp) | 1;   // odd: conflict-free B reads
    MV_REQUIRE(g.slab_stride <= 8 * 64, E_SHAPE, "mdnvae_conv2d_wgrad: slab of %d floats per channel is too large",
               g.slab_stride);
    set_lds_limit(reinterpret_cast<const void*>(conv_wgrad_kernel<false>));
    set_lds_limit(reinterpret_cast<const void*>(conv_wgrad_kernel<true>));
    dim3 grid((Cout + 63) / 64, (Cin + 63) / 64, Gz);
    if (in_stats)
        hipLaunchKernelGGL(conv_wgrad_kernel<true>, grid, dim3(WG_NT), best_lds, st, g, x, in_stats, grad_out, ptw,
                           cslab, spg, workspace);
    else
        hipLaunchKernelGGL(conv_wgrad_kernel<false>, grid, dim3(WG_NT), best_lds, st, g, x, in_stats, grad_out, ptw,
                           cslab, spg, workspace);
    MV_LAUNCH_CHECK("conv_wgrad_kernel");
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, workspace, Gz,
                       KK, Cout, Cin, grad_w);
    MV_LAUNCH_CHECK("conv_wgrad_reduce_kernel");
    return 0;
}

}  // extern "C"

template <int S, int PARTS, int RW>
static void launch_wgrad_bf16(const WgradBf16& p, dim3 grid, size_t lds, const float* x, const float* stats,
                              const float* grad_out, float* workspace, hipStream_t st) {
    set_lds_limit(reinterpret_cast<const void*>(conv_wgrad_bf16_kernel<S, PARTS, RW>));
    hipLaunchKernelGGL((conv_wgrad_bf16_kernel<S, PARTS, RW>), grid, dim3(256), lds, st, p, x, stats, grad_out, workspace);
}

extern "C" int mdnvae_conv_stem_wgrad_bf16(const float* x, const void* grad_out, float* workspace, float* grad_w, int B, int H,
                                           int W, int Cout, int KS, int pad, int grad16, void* stream) {
    MV_REQUIRE(grad16 == 0 || grad16 == 1, E_ARG, "mdnvae_conv_stem_wgrad_bf16: grad16 = %d", grad16);
    MV_REQUIRE(B >= 1 && H >= 1 && W >= 1, E_ARG, "mdnvae_conv_stem_wgrad_bf16: bad size");
    MV_REQUIRE(x && grad_out && workspace && grad_w, E_ARG, "mdnvae_conv_stem_wgrad_bf16: null pointer");
    MV_REQUIRE(KS == 7 && Cout <= 64 && pad < H && pad < W, E_SHAPE, "mdnvae_conv_stem_wgrad_bf16: 7x7 stem with at most 64 output channels");
    ConvGeom g{};
    g.B = B; g.Cin = 1; g.H = H; g.W = W; g.Cout = Cout;
    g.OHf = g.SH = H + 2 * pad - KS + 1;
    g.OWf = g.SW = W + 2 * pad - KS + 1;
    g.osy = g.osx = 1;
    g.S = 1; g.pad = pad; g.reflect = 1; g.relu_in = 0; g.KHe = KS;
    finish_geom(g, KS);
    return stem_wgrad_run(g, KS, x, grad_out, workspace, grad_w, true, (hipStream_t)stream,
                          grad16 ? (g.SH * g.SW + 7) / 8 * 8 : 0);
}

extern "C" int mdnvae_conv2d_wgrad_bf16(const float* x, const float* in_stats, int relu_in, const float* grad_out,
                                        float* workspace, float* grad_w, int B, int Cin, int H, int W, int Cout, int OH,
                                        int OW, int KS, int stride, int pad, int reflect, int parts, void* stream) {
    MV_REQUIRE(B >= 1, E_ARG, "mdnvae_conv2d_wgrad_bf16: empty batch");
    MV_REQUIRE(parts >= 1 && parts <= 3, E_ARG, "mdnvae_conv2d_wgrad_bf16: parts = %d (1, 2 or 3)", parts);
    MV_REQUIRE(x && grad_out && workspace && grad_w, E_ARG, "mdnvae_conv2d_wgrad_bf16: null pointer");
    MV_REQUIRE(KS == 3 && Cin > 1 && pad == 1 && (stride == 1 || stride == 2), E_SHAPE,
               "mdnvae_conv2d_wgrad_bf16: 3x3, pad 1, stride 1 or 2, more than one input channel (got k%d p%d s%d Cin %d)", KS,
               pad, stride, Cin);
    MV_REQUIRE(OH == (H + 2 * pad - KS) / stride + 1 && OW == (W + 2 * pad - KS) / stride + 1, E_SHAPE,
               "mdnvae_conv2d_wgrad_bf16: output size mismatch");
    MV_REQUIRE(!reflect || (H >= 2 && W >= 2), E_SHAPE, "mdnvae_conv2d_wgrad_bf16: reflection needs a plane of 2x2 or more");
    WgradBf16 p{};
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.pad = pad;
    p.reflect = reflect; p.relu_in = relu_in;
    p.G8 = (OW + 7) / 8;
    // the largest chunk of output rows whose tiles leave room for two workgroups
    // whole-row staging for the square stride-1 reflect layers of the residual blocks
    const int rw = (stride == 1 && reflect && H == W && (W == 22 || W == 17)) ? W : 0;
    p.ru = stride * (p.G8 + 1) + (rw ? 1 : 0);
    auto lds_of = [&](int nr) {
        const int rows = (nr - 1) * stride + 3;
        return (size_t)parts * 64 * (((nr * p.G8) | 1) + ((rows * p.ru) | 1)) * 16;
    };
    int best = 0;
    for (size_t cap : {(size_t)78 * 1024, (size_t)156 * 1024}) {      // room for two workgroups per CU, else for one
        for (int nr = 1; nr <= OH && nr <= 16; ++nr)
            if (lds_of(nr) <= cap) best = nr;
        if (best) break;
    }
    MV_REQUIRE(best > 0, E_SHAPE, "mdnvae_conv2d_wgrad_bf16: no chunk of output rows fits in LDS (OW = %d)", OW);
    p.nr = best;
    p.rows = (best - 1) * stride + 3;
    p.gunits = (best * p.G8) | 1;
    p.sunits = (p.rows * p.ru) | 1;
    const size_t lds = lds_of(best);
    const int ntiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    const int Gz0 = wgrad_groups(B, ntiles);
    p.spg = (B + Gz0 - 1) / Gz0;
    const int Gz = (B + p.spg - 1) / p.spg;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((Cout + 63) / 64, (Cin + 63) / 64, Gz);
#define MV_WG(S_, RW_)                                                                                  \
    do {                                                                                                \
        if (parts == 1) launch_wgrad_bf16<S_, 1, RW_>(p, grid, lds, x, in_stats, grad_out, workspace, st);      \
        else if (parts == 2) launch_wgrad_bf16<S_, 2, RW_>(p, grid, lds, x, in_stats, grad_out, workspace, st); \
        else launch_wgrad_bf16<S_, 3, RW_>(p, grid, lds, x, in_stats, grad_out, workspace, st);                 \
    } while (0)
    if (rw == 22) MV_WG(1, 22);
    else if (rw == 17) MV_WG(1, 17);
    else if (stride == 1) MV_WG(1, 0);
    else MV_WG(2, 0);
#undef MV_WG
    MV_LAUNCH_CHECK("conv_wgrad_bf16_kernel");
    const long total = 9L * Cout * Cin;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, workspace, Gz, 9,
                       Cout, Cin, grad_w);
    MV_LAUNCH_CHECK("conv_wgrad_reduce_kernel");
    return 0;
}

