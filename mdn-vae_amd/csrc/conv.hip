// Direct convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains).
//
// Replaces the ATen conv2d / convolution_backward calls of the encoder stack
//   function/VAE_CNN_NN.py:110-134 (stem 7x7 reflect-3, two stride-2 3x3, ten 3x3 reflect-1 convs)
// Activations stay NCHW fp32.  One workgroup owns one sample, a tile of PT consecutive output
// pixels and a tile of output channels; per chunk of CK input channels it stages
//   * the input rows that tile needs, with padding (zero / reflection), optional instance-norm
//     + ReLU of the producer layer, and optional zero-insertion (transposed stride-2 conv)
//     already applied, as slab[c][row][col] in LDS, and
//   * the weights of the chunk as wl[tap][c][n] in LDS,
// and then runs a tap loop of 32x32x2 MFMAs: A = 32 output pixels x 2 input channels read from
// the slab at a per-lane base + a per-tap scalar offset, B = 2 input channels x 32 output
// channels.  The same kernel computes the data gradient (correlation of the zero-padded,
// zero-inserted output gradient with the flipped, transposed weights).
//
// The weight gradient is its own kernel: D[n][c] per tap accumulates over pixels
// (A = grad_out[n][pixel], B = slab[c][pixel+tap]); sample groups write partial sums that a
// second kernel adds in a fixed order.
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct ConvGeom {
    int B, Cin, H, W;      // input tensor [B,Cin,H,W]
    int Cout, OH, OW;      // output tensor [B,Cout,OH,OW]
    int KS, S;             // kernel size; stride from output pixel to virtual-input pixel
    int pad, reflect;      // padding of the virtual input (reflect: 1 = reflection, 0 = zeros)
    int dil;               // virtual input = input with (dil-1) zeros inserted between pixels
    int VH, VW;            // virtual input size (>= (H-1)*dil+1; extra rows/cols read as zero)
    int CoutP;             // padded output-channel count of the packed weights
    int CK;                // input channels per chunk (even)
    int relu_in;           // apply ReLU after the (optional) normalisation on load
    int IWp;               // slab row length  = (OW-1)*S + KS
    int slab_stride;       // floats per channel in the slab
};

// value of the (padded, dilated, normalised) virtual input at padded coords (pr, pc)
__device__ __forceinline__ float virt_input(const ConvGeom& g, const float* __restrict__ plane, int pr, int pc,
                                            float mean, float rstd, bool norm) {
    int vy = pr - g.pad, vx = pc - g.pad;
    bool ok;
    if (g.reflect) {
        vy = vy < 0 ? -vy : (vy >= g.VH ? 2 * (g.VH - 1) - vy : vy);
        vx = vx < 0 ? -vx : (vx >= g.VW ? 2 * (g.VW - 1) - vx : vx);
        ok = true;
    } else {
        ok = vy >= 0 && vy < g.VH && vx >= 0 && vx < g.VW;
    }
    if (g.dil == 2) {
        ok = ok && !((vy | vx) & 1);
        vy >>= 1;
        vx >>= 1;
    }
    ok = ok && vy < g.H && vx < g.W;
    float v = 0.0f;
    if (ok) {
        v = plane[vy * g.W + vx];
        if (norm) v = (v - mean) * rstd;
        if (g.relu_in) v = fmaxf(v, 0.0f);
    }
    return v;
}

// ---------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ---------------------------------------------------------------------------------------
template <int WAVES, int WM, int WN>
__global__ __launch_bounds__(WAVES * 64) void conv_mfma_kernel(ConvGeom g, const float* __restrict__ x,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ wp,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = WAVES * 64;
    constexpr int PT = WAVES * WM * 32;
    constexpr int NTILE = WN * 32;
    const int KK = g.KS * g.KS;
    float* slab = smem;
    float* wl = smem + g.CK * g.slab_stride;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int P = g.OH * g.OW;
    const int p0 = blockIdx.x * PT, n0 = blockIdx.y * NTILE, b = blockIdx.z;
    const int oy0 = p0 / g.OW;
    const int plast = min(P - 1, p0 + PT - 1);
    const int nrows = (plast / g.OW - oy0) * g.S + g.KS;
    const int pr0 = oy0 * g.S;
    const int slab_elems = nrows * g.IWp;

    int lane_base[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int p = min(P - 1, p0 + (wave * WM + m) * 32 + l32);
        const int oy = p / g.OW, ox = p - oy * g.OW;
        lane_base[m] = (oy - oy0) * g.S * g.IWp + ox * g.S;
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[m][n][j] = 0.0f;

    const long in_plane = (long)g.H * g.W;
    for (int c0 = 0; c0 < g.Cin; c0 += g.CK) {
        __syncthreads();
        for (int idx = tid; idx < g.CK * slab_elems; idx += NT) {
            const int c = idx / slab_elems, rem = idx - c * slab_elems;
            const int r = rem / g.IWp, col = rem - r * g.IWp;
            const long pl = (long)b * g.Cin + c0 + c;
            float mean = 0.0f, rstd = 1.0f;
            if (stats) {
                mean = stats[2 * pl];
                rstd = stats[2 * pl + 1];
            }
            slab[c * g.slab_stride + rem] = virt_input(g, x + pl * in_plane, pr0 + r, col, mean, rstd, stats != nullptr);
        }
        const int nq = NTILE / 4;
        for (int i4 = tid; i4 < KK * g.CK * nq; i4 += NT) {
            const int row = i4 / nq, q = i4 - row * nq;
            const int tap = row / g.CK, c = row - tap * g.CK;
            const float4 v = *reinterpret_cast<const float4*>(wp + ((long)(tap * g.Cin + c0 + c)) * g.CoutP + n0 + q * 4);
            *reinterpret_cast<float4*>(wl + row * NTILE + q * 4) = v;
        }
        __syncthreads();
        for (int ky = 0; ky < g.KS; ++ky) {
            for (int kx = 0; kx < g.KS; ++kx) {
                const float* sl = slab + h * g.slab_stride + ky * g.IWp + kx;
                const float* wr = wl + ((ky * g.KS + kx) * g.CK + h) * NTILE + l32;
                for (int cp = 0; cp < g.CK / 2; ++cp) {
                    float a[WM], bb[WN];
#pragma unroll
                    for (int m = 0; m < WM; ++m) a[m] = sl[cp * 2 * g.slab_stride + lane_base[m]];
#pragma unroll
                    for (int n = 0; n < WN; ++n) bb[n] = wr[cp * 2 * NTILE + n * 32];
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[n], acc[m][n], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: accumulator register j of a 32x32 tile holds row (j&3)+8*(j>>2)+4*h, column l32
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int ch = n0 + n * 32 + l32;
        if (ch < g.Cout) {
            const float bv = bias ? bias[ch] : 0.0f;
            float* oplane = out + ((long)b * g.Cout + ch) * P;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                const int pbase = p0 + (wave * WM + m) * 32 + 4 * h;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int p = pbase + (j & 3) + 8 * (j >> 2);
                    if (p < P) oplane[p] = acc[m][n][j] + bv;
                }
            }
        }
    }
}

// Stem: one input channel, so the MFMA K dimension runs over the taps (49 -> 50, the last with
// zero weight).  wp is [50][CoutP].
template <int WAVES, int WM, int WN>
__global__ __launch_bounds__(WAVES * 64) void conv_stem_kernel(ConvGeom g, const float* __restrict__ x,
                                                               const float* __restrict__ wp,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = WAVES * 64;
    constexpr int PT = WAVES * WM * 32;
    constexpr int NTILE = WN * 32;
    const int KK = g.KS * g.KS;
    const int KKP = (KK + 1) & ~1;
    float* slab = smem;
    float* wl = smem + g.slab_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int P = g.OH * g.OW;
    const int p0 = blockIdx.x * PT, n0 = blockIdx.y * NTILE, b = blockIdx.z;
    const int oy0 = p0 / g.OW;
    const int plast = min(P - 1, p0 + PT - 1);
    const int nrows = (plast / g.OW - oy0) * g.S + g.KS;
    const int pr0 = oy0 * g.S;
    const int slab_elems = nrows * g.IWp;
    int lane_base[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int p = min(P - 1, p0 + (wave * WM + m) * 32 + l32);
        const int oy = p / g.OW, ox = p - oy * g.OW;
        lane_base[m] = (oy - oy0) * g.S * g.IWp + ox * g.S;
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[m][n][j] = 0.0f;

    const float* plane = x + (long)b * g.H * g.W;
    for (int idx = tid; idx < slab_elems; idx += NT) {
        const int r = idx / g.IWp, col = idx - r * g.IWp;
        slab[idx] = virt_input(g, plane, pr0 + r, col, 0.0f, 1.0f, false);
    }
    for (int i = tid; i < KKP * NTILE; i += NT) {
        const int tap = i / NTILE, n = i - tap * NTILE;
        wl[i] = wp[(long)tap * g.CoutP + n0 + n];
    }
    __syncthreads();
    for (int step = 0; step < KKP / 2; ++step) {
        const int tap = 2 * step + h;
        const int ta = min(tap, KK - 1);            // address of the padded tap stays inside the slab
        const int ky = ta / g.KS, kx = ta - ky * g.KS;
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
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int ch = n0 + n * 32 + l32;
        if (ch < g.Cout) {
            const float bv = bias ? bias[ch] : 0.0f;
            float* oplane = out + ((long)b * g.Cout + ch) * P;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                const int pbase = p0 + (wave * WM + m) * 32 + 4 * h;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int p = pbase + (j & 3) + 8 * (j >> 2);
                    if (p < P) oplane[p] = acc[m][n][j] + bv;
                }
            }
        }
    }
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
// Workgroup tile: 64 output channels x 64 input channels x all taps (KK <= 9); 4 waves as
// 2 (n) x 2 (c).  K dimension = output pixels of the samples of this group.
// partial [G][KK][Cout][Cin]
constexpr int WG_NT = 256;
template <int KKT>
__global__ __launch_bounds__(WG_NT) void conv_wgrad_kernel(ConvGeom g, const float* __restrict__ x,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gout, int ptw,
                                                           int samples_per_group, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int gl_stride = ptw + 1;
    float* gl = smem;                               // [64][ptw+1]
    float* slab = smem + 64 * gl_stride;            // [64][slab_stride]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int wn = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int P = g.OH * g.OW;
    const int b_begin = blockIdx.z * samples_per_group;
    const int b_end = min(g.B, b_begin + samples_per_group);
    const long in_plane = (long)g.H * g.W;

    f32x16 acc[KKT];
#pragma unroll
    for (int t = 0; t < KKT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.0f;

    for (int b = b_begin; b < b_end; ++b) {
        for (int p0 = 0; p0 < P; p0 += ptw) {
            const int oy0 = p0 / g.OW;
            const int plast = min(P - 1, p0 + ptw - 1);
            const int nrows = (plast / g.OW - oy0) * g.S + g.KS;
            const int pr0 = oy0 * g.S;
            const int slab_elems = nrows * g.IWp;
            __syncthreads();
            for (int idx = tid; idx < 64 * ptw; idx += WG_NT) {
                const int n = idx / ptw, pp = idx - n * ptw;
                float v = 0.0f;
                if (n0 + n < g.Cout && p0 + pp < P) v = gout[((long)b * g.Cout + n0 + n) * P + p0 + pp];
                gl[n * gl_stride + pp] = v;
            }
            for (int idx = tid; idx < 64 * slab_elems; idx += WG_NT) {
                const int c = idx / slab_elems, rem = idx - c * slab_elems;
                float v = 0.0f;
                if (c0 + c < g.Cin) {
                    const int r = rem / g.IWp, col = rem - r * g.IWp;
                    const long pl = (long)b * g.Cin + c0 + c;
                    float mean = 0.0f, rstd = 1.0f;
                    if (stats) {
                        mean = stats[2 * pl];
                        rstd = stats[2 * pl + 1];
                    }
                    v = virt_input(g, x + pl * in_plane, pr0 + r, col, mean, rstd, stats != nullptr);
                }
                slab[c * g.slab_stride + rem] = v;
            }
            __syncthreads();
            const float* ga = gl + (wn * 32 + l32) * gl_stride + h;
            const float* sb = slab + (wc * 32 + l32) * g.slab_stride;
            // pixel (p0 + k + h) -> slab offset, advanced by two pixels per step
            int pix = p0 + h;
            int oy = pix / g.OW, ox = pix - oy * g.OW;
            const int npairs = (min(ptw, P - p0) + 1) >> 1;
            for (int k = 0; k < npairs; ++k) {
                const float a = ga[2 * k];
                // a pixel past the end of the plane carries a zero gradient; clamp its address
                const int oyc = min(oy, g.OH - 1);
                const int off = (oyc - oy0) * g.S * g.IWp + ox * g.S;
#pragma unroll
                for (int t = 0; t < KKT; ++t) {
                    const int ky = t / 3, kx = t - ky * 3;      // KKT == 9 -> 3x3 taps
                    const float bv = sb[off + ky * g.IWp + kx];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
                }
                ox += 2;
                if (ox >= g.OW) {
                    ox -= g.OW;
                    ++oy;
                }
            }
        }
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

// dW[n][c][tap] = sum_z partial[z][tap][n][c]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int G, int KK,
                                                                int Cout, int Cin, float* __restrict__ dw) {
    const long total = (long)KK * Cout * Cin;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float s = 0.0f;
    for (int z = 0; z < G; ++z) s += partial[(long)z * total + idx];
    const int c = (int)(idx % Cin);
    const long t = idx / Cin;
    const int n = (int)(t % Cout), tap = (int)(t / Cout);
    dw[((long)n * Cin + c) * KK + tap] = s;
}

// Stem weight gradient (one input channel): D[n][tap] accumulates over pixels.
// 4 waves split the pixels of each chunk; their accumulators are added through LDS at the end.
// partial [G][Cout][KK]
__global__ __launch_bounds__(256) void conv_stem_wgrad_kernel(ConvGeom g, const float* __restrict__ x,
                                                              const float* __restrict__ gout, int ptw,
                                                              int samples_per_group, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KK = g.KS * g.KS;
    const int gl_stride = ptw + 1;
    float* gl = smem;                          // [64][ptw+1]
    float* slab = smem + 64 * gl_stride;       // [slab_stride] (+ one zero word at the end)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int P = g.OH * g.OW;
    const int b_begin = blockIdx.z * samples_per_group;
    const int b_end = min(g.B, b_begin + samples_per_group);
    // per-lane tap offsets for the two tap tiles (taps >= KK read the zero word)
    int toff[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int tap = jt * 32 + l32;
        const int ky = tap / g.KS, kx = tap - ky * g.KS;
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

    for (int b = b_begin; b < b_end; ++b) {
        const float* plane = x + (long)b * g.H * g.W;
        for (int p0 = 0; p0 < P; p0 += ptw) {
            const int oy0 = p0 / g.OW;
            const int plast = min(P - 1, p0 + ptw - 1);
            const int nrows = (plast / g.OW - oy0) * g.S + g.KS;
            const int pr0 = oy0 * g.S;
            const int slab_elems = nrows * g.IWp;
            __syncthreads();
            for (int idx = tid; idx < 64 * ptw; idx += 256) {
                const int n = idx / ptw, pp = idx - n * ptw;
                float v = 0.0f;
                if (n < g.Cout && p0 + pp < P) v = gout[((long)b * g.Cout + n) * P + p0 + pp];
                gl[n * gl_stride + pp] = v;
            }
            for (int idx = tid; idx < slab_elems; idx += 256) {
                const int r = idx / g.IWp, col = idx - r * g.IWp;
                slab[idx] = virt_input(g, plane, pr0 + r, col, 0.0f, 1.0f, false);
            }
            if (tid == 0) slab[zero_at] = 0.0f;
            __syncthreads();
            // wave w takes pixel pairs w, w+4, w+8, ...
            const int npairs = (min(ptw, P - p0) + 1) >> 1;
            for (int k = wave; k < npairs; k += 4) {
                const int pix = min(P - 1, p0 + 2 * k + h);
                const int oy = pix / g.OW, ox = pix - oy * g.OW;
                const int off = (oy - oy0) * g.S * g.IWp + ox * g.S;
                float a[2], bv[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = gl[(i * 32 + l32) * gl_stride + 2 * k + h];
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) bv[jt] = slab[toff[jt] >= 0 ? off + toff[jt] : zero_at];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt)
                        acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[jt], acc[i][jt], 0, 0, 0);
            }
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

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int G, long n,
                                                           float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    float s = 0.0f;
    for (int z = 0; z < G; ++z) s += partial[(long)z * n + idx];
    out[idx] = s;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct ConvPlan {
    ConvGeom g;
    int waves, wm, wn;   // tile shape
    int pt, ntile;
    size_t lds;
};

static int out_rows_touched(int pt, int OW, int OH) {
    int r = (pt + OW - 2) / OW + 1;
    return r < OH ? r : OH;
}

static void set_attr_once(const void* fn) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <int WAVES, int WM, int WN>
static int launch_conv(const ConvGeom& g, size_t lds, const float* x, const float* stats, const float* wp,
                       const float* bias, float* out, bool stem, hipStream_t st) {
    constexpr int PT = WAVES * WM * 32, NTILE = WN * 32;
    const int P = g.OH * g.OW;
    dim3 grid((P + PT - 1) / PT, g.CoutP / NTILE, g.B);
    if (stem) {
        static bool once = false;
        if (!once) { set_attr_once(reinterpret_cast<const void*>(conv_stem_kernel<WAVES, WM, WN>)); once = true; }
        hipLaunchKernelGGL((conv_stem_kernel<WAVES, WM, WN>), grid, dim3(WAVES * 64), lds, st, g, x, wp, bias, out);
    } else {
        static bool once = false;
        if (!once) { set_attr_once(reinterpret_cast<const void*>(conv_mfma_kernel<WAVES, WM, WN>)); once = true; }
        hipLaunchKernelGGL((conv_mfma_kernel<WAVES, WM, WN>), grid, dim3(WAVES * 64), lds, st, g, x, stats, wp, bias,
                           out);
    }
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

// Forward convolution (transposed = 0) or data gradient (transposed = 1).
int mdnvae_conv2d(const float* x, const float* in_stats, int relu_in, const float* packed_w, const float* bias,
                  float* out, int B, int Cin, int H, int W, int Cout, int OH, int OW, int KS, int stride, int pad,
                  int reflect, int transposed, void* stream) {
    MV_REQUIRE(B >= 0, E_ARG, "mdnvae_conv2d: negative batch");
    if (B == 0) return 0;
    MV_REQUIRE(x && packed_w && out, E_ARG, "mdnvae_conv2d: null pointer");
    MV_REQUIRE(B <= 65535, E_SHAPE, "mdnvae_conv2d: batch %d exceeds the grid limit 65535", B);
    MV_REQUIRE(KS == 3 || KS == 7, E_SHAPE, "mdnvae_conv2d: kernel size %d (3 or 7 supported)", KS);
    MV_REQUIRE(stride == 1 || stride == 2, E_SHAPE, "mdnvae_conv2d: stride %d", stride);
    MV_REQUIRE(!(reflect && (transposed || stride != 1)), E_SHAPE, "mdnvae_conv2d: reflection padding needs stride 1");
    MV_REQUIRE(Cin == 1 || Cin % 8 == 0, E_SHAPE, "mdnvae_conv2d: Cin %d must be 1 or a multiple of 8", Cin);
    MV_REQUIRE(!(Cin == 1 && (transposed || in_stats)), E_SHAPE, "mdnvae_conv2d: single-channel input is forward only");
    ConvGeom g{};
    g.B = B; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.OH = OH; g.OW = OW; g.KS = KS;
    g.relu_in = relu_in;
    if (!transposed) {
        g.S = stride; g.pad = pad; g.reflect = reflect; g.dil = 1; g.VH = H; g.VW = W;
        MV_REQUIRE(OH == (H + 2 * pad - KS) / stride + 1 && OW == (W + 2 * pad - KS) / stride + 1, E_SHAPE,
                   "mdnvae_conv2d: output %dx%d does not match input %dx%d k%d s%d p%d", OH, OW, H, W, KS, stride, pad);
        MV_REQUIRE(!reflect || (pad < H && pad < W), E_SHAPE, "mdnvae_conv2d: reflection pad %d too large", pad);
    } else {
        // correlation of the zero-inserted, zero-padded gradient with the flipped weights:
        // here (H,W) is the gradient w.r.t. the forward output and (OH,OW) the forward input plane
        g.S = 1; g.pad = KS - 1 - pad; g.reflect = 0; g.dil = stride;
        g.VH = 1 << 28; g.VW = 1 << 28;   // the upper bound is the gradient's own extent (H, W)
        MV_REQUIRE(g.pad >= 0, E_SHAPE, "mdnvae_conv2d: transposed padding %d", pad);
        MV_REQUIRE(H == (OH + 2 * pad - KS) / stride + 1 && W == (OW + 2 * pad - KS) / stride + 1, E_SHAPE,
                   "mdnvae_conv2d: gradient %dx%d is not the k%d s%d p%d image of a %dx%d plane", H, W, KS, stride, pad,
                   OH, OW);
    }
    g.IWp = (OW - 1) * g.S + KS;
    const int P = OH * OW;
    const bool stem = (Cin == 1);
    int waves, wm, wn;
    if (stem) {
        waves = 4; wm = 2; wn = 2;
        g.CoutP = round_up(Cout, 64);
    } else {
        wn = 3;
        g.CoutP = round_up(Cout, 96);
        // pixel tile of 256, 320 or 384: the one that wastes the smallest share of MFMA rows
        wm = 2; waves = 4;
        double best = 2.0;
        for (int wv = 4; wv <= 6; ++wv) {
            const int t = wv * 64;
            const double waste = (double)(round_up(P, t) - P) / round_up(P, t);
            if (waste < best - 1e-9) { best = waste; waves = wv; }
        }
    }
    const int pt = waves * wm * 32;
    const int rows = (out_rows_touched(pt, OW, OH) - 1) * g.S + KS;
    g.slab_stride = rows * g.IWp + 1;
    const int KK = KS * KS;
    size_t lds;
    if (stem) {
        g.CK = 1;
        lds = ((size_t)g.slab_stride + (size_t)((KK + 1) & ~1) * wn * 32) * 4;
    } else {
        g.CK = 8;
        lds = ((size_t)g.CK * g.slab_stride + (size_t)KK * g.CK * wn * 32) * 4;
        if (lds > 64 * 1024 && Cin % 4 == 0) {
            g.CK = 4;
            lds = ((size_t)g.CK * g.slab_stride + (size_t)KK * g.CK * wn * 32) * 4;
        }
    }
    MV_REQUIRE(lds <= 160 * 1024, E_SHAPE, "mdnvae_conv2d: tile needs %zu bytes of LDS", lds);
    hipStream_t st = (hipStream_t)stream;
    if (stem) launch_conv<4, 2, 2>(g, lds, x, nullptr, packed_w, bias, out, true, st);
    else if (waves == 6) launch_conv<6, 2, 3>(g, lds, x, in_stats, packed_w, bias, out, false, st);
    else if (waves == 5) launch_conv<5, 2, 3>(g, lds, x, in_stats, packed_w, bias, out, false, st);
    else launch_conv<4, 2, 3>(g, lds, x, in_stats, packed_w, bias, out, false, st);
    MV_LAUNCH_CHECK("conv_mfma_kernel");
    return 0;
}

size_t mdnvae_conv_wgrad_workspace_floats(int B, int Cout, int Cin, int KS) {
    const int G = B < 64 ? B : 64;
    return (size_t)G * KS * KS * Cout * Cin;
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
    g.B = B; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.OH = OH; g.OW = OW; g.KS = KS;
    g.S = stride; g.pad = pad; g.reflect = reflect; g.dil = 1; g.VH = H; g.VW = W; g.relu_in = relu_in;
    g.IWp = (OW - 1) * stride + KS;
    const int KK = KS * KS;
    const int G = B < 64 ? B : 64;
    const int spg = (B + G - 1) / G;
    const int Gz = (B + spg - 1) / spg;
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 1) {
        const int ptw = 256;
        const int rows = (out_rows_touched(ptw, OW, OH) - 1) * stride + KS;
        g.slab_stride = rows * g.IWp + 2;
        size_t lds = ((size_t)64 * (ptw + 1) + g.slab_stride) * 4;
        const size_t red = (size_t)4 * 64 * 65 * 4;
        if (lds < red) lds = red;
        MV_REQUIRE(lds <= 160 * 1024, E_SHAPE, "mdnvae_conv2d_wgrad: needs %zu bytes of LDS", lds);
        static bool once = false;
        if (!once) { set_attr_once(reinterpret_cast<const void*>(conv_stem_wgrad_kernel)); once = true; }
        hipLaunchKernelGGL(conv_stem_wgrad_kernel, dim3(1, 1, Gz), dim3(256), lds, st, g, x, grad_out, ptw, spg,
                           workspace);
        MV_LAUNCH_CHECK("conv_stem_wgrad_kernel");
        const long n = (long)Cout * KK;
        hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace, Gz, n,
                           grad_w);
        MV_LAUNCH_CHECK("sum_partials_kernel");
        return 0;
    }
    // pixel chunk: the largest of 128/64/32 whose tiles fit in LDS
    int ptw = 128;
    size_t lds = 0;
    for (; ptw >= 32; ptw >>= 1) {
        const int rows = (out_rows_touched(ptw, OW, OH) - 1) * stride + KS;
        g.slab_stride = (rows * g.IWp) | 1;
        lds = ((size_t)64 * (ptw + 1) + (size_t)64 * g.slab_stride) * 4;
        if (lds <= 150 * 1024) break;
    }
    MV_REQUIRE(ptw >= 32, E_SHAPE, "mdnvae_conv2d_wgrad: tile needs %zu bytes of LDS", lds);
    static bool once = false;
    if (!once) { set_attr_once(reinterpret_cast<const void*>(conv_wgrad_kernel<9>)); once = true; }
    dim3 grid((Cout + 63) / 64, (Cin + 63) / 64, Gz);
    hipLaunchKernelGGL(conv_wgrad_kernel<9>, grid, dim3(WG_NT), lds, st, g, x, in_stats, grad_out, ptw, spg, workspace);
    MV_LAUNCH_CHECK("conv_wgrad_kernel");
    const long total = (long)KK * Cout * Cin;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, workspace, Gz,
                       KK, Cout, Cin, grad_w);
    MV_LAUNCH_CHECK("conv_wgrad_reduce_kernel");
    return 0;
}

}  // extern "C"
