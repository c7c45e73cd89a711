// Instance-norm statistics, normalise(+ReLU)(+residual) materialisation, the fused
// instance-norm/ReLU backward (with the reflection-pad fold on load), gradient fold-add,
// and the 8x8 max pool, all on NCHW fp32 planes.
//
// Reference semantics: nn.InstanceNorm2d defaults (affine=False, eps=1e-5, biased variance,
// no running statistics) and nn.ReLU at function/VAE_CNN_NN.py:113-128 and :87-98,
// nn.MaxPool2d(8, 8) at :137,158.  All are HBM-bound; a plane (<= 8192 floats) is held in
// registers between its reduction and its write, so each tensor is read once.
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

constexpr float IN_EPS = 1e-5f;
constexpr int PLANE_MAX = 8192;     // 256 threads x 32 elements

// A plane is owned by TPP threads (64 = one wave, or 256 = the workgroup), EPT elements each.
template <int TPP>
__device__ __forceinline__ float plane_sum(float v, float* scratch, int sub) {
    if (TPP == 64) {
        return wave_sum_all(v);
    } else {
        v = wave_sum(v);
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) scratch[wid] = v;
        __syncthreads();
        return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    }
}

// stats[plane] = {mean, 1/sqrt(var+eps)}
template <int TPP, int EPT>
__global__ __launch_bounds__(256) void plane_stats_kernel(const float* __restrict__ x, long nplanes, int P,
                                                          float* __restrict__ stats) {
    __shared__ float scratch[4];
    constexpr int PPB = 256 / TPP;
    const int sub = threadIdx.x / TPP, t = threadIdx.x % TPP;
    const long plane = (long)blockIdx.x * PPB + sub;
    const bool live = plane < nplanes;
    const float* px = x + (live ? plane : 0) * (long)P;
    float v[EPT];
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = t + e * TPP;
        v[e] = (live && i < P) ? px[i] : 0.0f;
        s += v[e];
    }
    const float mean = plane_sum<TPP>(s, scratch, sub) / (float)P;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = t + e * TPP;
        const float d = (i < P) ? v[e] - mean : 0.0f;
        q += d * d;
    }
    const float var = plane_sum<TPP>(q, scratch, sub) / (float)P;
    if (live && t == 0) {
        stats[2 * plane] = mean;
        stats[2 * plane + 1] = 1.0f / sqrtf(var + IN_EPS);
    }
}

// Fused statistics + normalise: out = (skip ? skip : 0) + [relu]((raw - mean) * rstd), and
// stats[plane] = {mean, rstd}.  The plane stays in registers between the two reductions and the
// write, so raw is read once.
template <int TPP, int EPT>
__global__ __launch_bounds__(256) void inorm_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ skip,
                                                        int relu, long nplanes, int P, float* __restrict__ out,
                                                        float* __restrict__ stats) {
    __shared__ float scratch[4];
    constexpr int PPB = 256 / TPP;
    const int sub = threadIdx.x / TPP, t = threadIdx.x % TPP;
    const long plane = (long)blockIdx.x * PPB + sub;
    const bool live = plane < nplanes;
    const long pl = live ? plane : 0;
    const float* px = raw + pl * (long)P;
    float v[EPT];
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = t + e * TPP;
        v[e] = (live && i < P) ? px[i] : 0.0f;
        s += v[e];
    }
    const float mean = plane_sum<TPP>(s, scratch, sub) / (float)P;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = t + e * TPP;
        const float d = (i < P) ? v[e] - mean : 0.0f;
        q += d * d;
    }
    const float var = plane_sum<TPP>(q, scratch, sub) / (float)P;
    const float rstd = 1.0f / sqrtf(var + IN_EPS);
    if (live && t == 0) {
        stats[2 * plane] = mean;
        stats[2 * plane + 1] = rstd;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = t + e * TPP;
        if (live && i < P) {
            float o = (v[e] - mean) * rstd;
            if (relu) o = fmaxf(o, 0.0f);
            if (skip) o += skip[pl * (long)P + i];
            out[pl * (long)P + i] = o;
        }
    }
}

// out = (skip ? skip : 0) + act((raw - mean) * rstd)
__global__ __launch_bounds__(256) void norm_act_add_kernel(const float* __restrict__ raw, const float* __restrict__ stats,
                                                           const float* __restrict__ skip, int relu, long n, int P,
                                                           float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long plane = i / P;
        float v = (raw[i] - stats[2 * plane]) * stats[2 * plane + 1];
        if (relu) v = fmaxf(v, 0.0f);
        if (skip) v += skip[i];
        out[i] = v;
    }
}

// Gradient of a reflection-padded (pad 1) plane folded back onto the un-padded plane:
// element (y,x) of an HxW plane collects padded positions (y+1,x+1) plus the mirrored border.
__device__ __forceinline__ float fold_load(const float* __restrict__ gp, int y, int x, int H, int W) {
    const int Wp = W + 2;
    const int r0 = y + 1, c0 = x + 1;
    const int r1 = (y == 1) ? 0 : ((y == H - 2) ? H + 1 : -1);
    const int r2 = (y == 1 && y == H - 2) ? H + 1 : -1;
    const int c1 = (x == 1) ? 0 : ((x == W - 2) ? W + 1 : -1);
    const int c2 = (x == 1 && x == W - 2) ? W + 1 : -1;
    float s = 0.0f;
    const int rows[3] = {r0, r1, r2};
    const int cols[3] = {c0, c1, c2};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (rows[a] >= 0 && cols[b] >= 0) s += gp[rows[a] * Wp + cols[b]];
    return s;
}

// g_raw = rstd * (g - mean(g) - y*mean(g*y)),  y = (raw-mean)*rstd,  g = g_act * [y>0 if relu]
// FOLD: g_act is given on the reflection-padded plane [(H+2)x(W+2)].
// plane_gsum[plane] = sum of g_raw over the plane (the conv-bias gradient contribution).
template <int TPP, int EPT, bool FOLD>
__global__ __launch_bounds__(256) void inorm_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ stats,
                                                        const float* __restrict__ gact, int relu, long nplanes, int H,
                                                        int W, float* __restrict__ graw,
                                                        float* __restrict__ plane_gsum) {
    __shared__ float scratch[4];
    constexpr int PPB = 256 / TPP;
    const int P = H * W;
    const int sub = threadIdx.x / TPP, t = threadIdx.x % TPP;
    const long plane = (long)blockIdx.x * PPB + sub;
    const bool live = plane < nplanes;
    const long pl = live ? plane : 0;
    const float* praw = raw + pl * (long)P;
    const float* pg = gact + pl * (long)(FOLD ? (H + 2) * (W + 2) : P);
    const float mean = stats[2 * pl], rstd = stats[2 * pl + 1];
    float y[EPT], g[EPT];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = t + e * TPP;
        y[e] = 0.0f;
        g[e] = 0.0f;
        if (live && i < P) {
            y[e] = (praw[i] - mean) * rstd;
            float gv;
            if (FOLD) {
                const int yy = i / W, xx = i - yy * W;
                gv = fold_load(pg, yy, xx, H, W);
            } else {
                gv = pg[i];
            }
            if (relu && !(y[e] > 0.0f)) gv = 0.0f;
            g[e] = gv;
            s1 += gv;
            s2 += gv * y[e];
        }
    }
    const float m1 = plane_sum<TPP>(s1, scratch, sub) / (float)P;
    const float m2 = plane_sum<TPP>(s2, scratch, sub) / (float)P;
    float gs = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = t + e * TPP;
        if (live && i < P) {
            const float o = rstd * (g[e] - m1 - y[e] * m2);
            graw[pl * (long)P + i] = o;
            gs += o;
        }
    }
    if (plane_gsum) {
        const float tot = plane_sum<TPP>(gs, scratch, sub);
        if (live && t == 0) plane_gsum[plane] = tot;
    }
}

// out = base + fold(gp)   (gradient through a residual skip plus a reflection-padded conv input)
__global__ __launch_bounds__(256) void fold_add_kernel(const float* __restrict__ base, const float* __restrict__ gp,
                                                       long nplanes, int H, int W, float* __restrict__ out) {
    const int P = H * W;
    const long n = nplanes * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long plane = i / P;
        const int r = (int)(i - plane * P);
        const int y = r / W, x = r - y * W;
        float v = fold_load(gp + plane * (long)(H + 2) * (W + 2), y, x, H, W);
        if (base) v += base[i];
        out[i] = v;
    }
}

// 8x8/8 max pool: feat[plane][py][px], argmax = flat index inside the plane (first maximum in
// row-major window order)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, long nplanes, int H, int W,
                                                          int K, int PH, int PW, float* __restrict__ feat,
                                                          int* __restrict__ argmax) {
    const long n = nplanes * PH * PW;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long plane = i / (PH * PW);
    const int r = (int)(i - plane * PH * PW);
    const int py = r / PW, px = r - py * PW;
    const float* p = x + plane * (long)H * W;
    float best = -INFINITY;
    int bi = (py * K) * W + px * K;
    for (int dy = 0; dy < K; ++dy)
        for (int dx = 0; dx < K; ++dx) {
            const int idx = (py * K + dy) * W + px * K + dx;
            const float v = p[idx];
            if (v > best) {
                best = v;
                bi = idx;
            }
        }
    feat[i] = best;
    argmax[i] = bi;
}

// K * K <= 64: one wave per pooling window, one entry per lane, (value, index) reduced over the wave; among equal
// values the lowest index wins, as in the serial scan above
__global__ __launch_bounds__(256) void maxpool_fwd_wave_kernel(const float* __restrict__ x, long nplanes, int H, int W,
                                                               int K, int PH, int PW, float* __restrict__ feat,
                                                               int* __restrict__ argmax) {
    const long n = nplanes * PH * PW;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    const long plane = i / (PH * PW);
    const int r = (int)(i - plane * PH * PW);
    const int py = r / PW, px = r - py * PW;
    const int dy = lane / K, dx = lane - dy * K;
    const bool in = lane < K * K;
    int bi = in ? (py * K + dy) * W + px * K + dx : 0x7fffffff;
    float best = in ? x[plane * (long)H * W + bi] : -INFINITY;
    if (!(best == best)) best = -INFINITY;             // a NaN never wins (v > best is false for it above)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        feat[i] = best;
        argmax[i] = bi;
    }
}

// K = 8 with 2 x 2 windows (planes of 16 .. 23 rows and columns, the encoder's 22 x 22 and 17 x 17): one wave per plane.
// Lane l scans row l / 4, columns 4 (l % 4) .. + 3 of the covered 16 x 16 corner -- a wave load touches whole 64-byte
// row pieces -- and the 16 lanes of a window (lane bit 5 = window row, bit 1 = window column) reduce by exchanges over
// bits 0, 2, 3, 4.  Same result as the serial scan: the first maximum in row-major order, NaNs never win.
__global__ __launch_bounds__(256) void maxpool8_plane_kernel(const float* __restrict__ x, long nplanes, int H, int W,
                                                             float* __restrict__ feat, int* __restrict__ argmax) {
    const long plane = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= nplanes) return;
    const int lane = threadIdx.x & 63;
    const int first = (lane >> 2) * W + (lane & 3) * 4;
    const float* p = x + plane * (long)H * W + first;
    float best = -INFINITY;
    int bi = first;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = p[j];
        if (v > best) {
            best = v;
            bi = first + j;
        }
    }
#pragma unroll
    for (int o = 1; o <= 16; o = o == 1 ? 4 : o * 2) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if ((lane & 0x1d) == 0) {
        const long w = plane * 4 + (lane >> 5) * 2 + ((lane >> 1) & 1);
        feat[w] = best;
        argmax[w] = bi;
    }
}

// its backward: one wave per plane writes the plane, zero except at the four recorded positions
__global__ __launch_bounds__(256) void maxpool8_bwd_plane_kernel(const float* __restrict__ gfeat, const int* __restrict__ argmax,
                                                                 long nplanes, int P, float* __restrict__ gx) {
    const long plane = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= nplanes) return;
    const int lane = threadIdx.x & 63;
    const int4 am = *reinterpret_cast<const int4*>(argmax + plane * 4);
    const float4 gf = *reinterpret_cast<const float4*>(gfeat + plane * 4);
    float* o = gx + plane * (long)P;
    for (int e = lane; e < P; e += 64) {
        float v = 0.0f;
        if (e == am.x) v = gf.x;
        if (e == am.y) v = gf.y;
        if (e == am.z) v = gf.z;
        if (e == am.w) v = gf.w;
        o[e] = v;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ gfeat, const int* __restrict__ argmax,
                                                          long nplanes, int H, int W, int K, int PH, int PW,
                                                          float* __restrict__ gx) {
    const int P = H * W;
    const long n = nplanes * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long plane = i / P;
        const int r = (int)(i - plane * P);
        const int y = r / W, x = r - y * W;
        const int py = y / K, px = x / K;
        float v = 0.0f;
        if (py < PH && px < PW) {
            const long w = plane * PH * PW + py * PW + px;
            if (argmax[w] == r) v = gfeat[w];
        }
        gx[i] = v;
    }
}

static int ew_grid(long n) {
    long g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// dispatch on plane size: one wave per plane up to 1024 elements, a workgroup per plane above
#define MV_PLANE_DISPATCH(P, CALL)                           \
    do {                                                     \
        if ((P) <= 256) { CALL(64, 4); }                     \
        else if ((P) <= 512) { CALL(64, 8); }                \
        else if ((P) <= 1024) { CALL(64, 16); }              \
        else if ((P) <= 2048) { CALL(256, 8); }              \
        else if ((P) <= 4096) { CALL(256, 16); }             \
        else { CALL(256, 32); }                              \
    } while (0)

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

int mdnvae_plane_stats(const float* x, long nplanes, int P, float* stats, void* stream) {
    MV_REQUIRE(nplanes >= 0 && P >= 1, E_ARG, "mdnvae_plane_stats: bad size");
    MV_REQUIRE(P <= PLANE_MAX, E_SHAPE, "mdnvae_plane_stats: plane of %d elements exceeds %d", P, PLANE_MAX);
    if (nplanes == 0) return 0;
    MV_REQUIRE(x && stats, E_ARG, "mdnvae_plane_stats: null pointer");
    hipStream_t st = (hipStream_t)stream;
#define CALL(TPP, EPT)                                                                                          \
    hipLaunchKernelGGL((plane_stats_kernel<TPP, EPT>), dim3((unsigned)((nplanes + (256 / TPP) - 1) / (256 / TPP))), \
                       dim3(256), 0, st, x, nplanes, P, stats)
    MV_PLANE_DISPATCH(P, CALL);
#undef CALL
    MV_LAUNCH_CHECK("plane_stats_kernel");
    return 0;
}

int mdnvae_inorm_fwd(const float* raw, const float* skip, int relu, long nplanes, int P, float* out, float* stats,
                     void* stream) {
    MV_REQUIRE(nplanes >= 0 && P >= 1, E_ARG, "mdnvae_inorm_fwd: bad size");
    MV_REQUIRE(P <= PLANE_MAX, E_SHAPE, "mdnvae_inorm_fwd: plane of %d elements exceeds %d", P, PLANE_MAX);
    if (nplanes == 0) return 0;
    MV_REQUIRE(raw && out && stats, E_ARG, "mdnvae_inorm_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
#define CALL(TPP, EPT)                                                                                        \
    hipLaunchKernelGGL((inorm_fwd_kernel<TPP, EPT>), dim3((unsigned)((nplanes + (256 / TPP) - 1) / (256 / TPP))), \
                       dim3(256), 0, st, raw, skip, relu, nplanes, P, out, stats)
    MV_PLANE_DISPATCH(P, CALL);
#undef CALL
    MV_LAUNCH_CHECK("inorm_fwd_kernel");
    return 0;
}

int mdnvae_norm_act_add(const float* raw, const float* stats, const float* skip, int relu, long nplanes, int P,
                        float* out, void* stream) {
    MV_REQUIRE(nplanes >= 0 && P >= 1, E_ARG, "mdnvae_norm_act_add: bad size");
    if (nplanes == 0) return 0;
    MV_REQUIRE(raw && stats && out, E_ARG, "mdnvae_norm_act_add: null pointer");
    const long n = nplanes * P;
    hipLaunchKernelGGL(norm_act_add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, raw, stats, skip, relu,
                       n, P, out);
    MV_LAUNCH_CHECK("norm_act_add_kernel");
    return 0;
}

int mdnvae_inorm_bwd(const float* raw, const float* stats, const float* grad_act, int relu, int folded,
                     long nplanes, int H, int W, float* grad_raw, float* plane_gsum, void* stream) {
    MV_REQUIRE(nplanes >= 0 && H >= 1 && W >= 1, E_ARG, "mdnvae_inorm_bwd: bad size");
    const int P = H * W;
    MV_REQUIRE(P <= PLANE_MAX, E_SHAPE, "mdnvae_inorm_bwd: plane of %d elements exceeds %d", P, PLANE_MAX);
    MV_REQUIRE(!folded || (H >= 3 && W >= 3), E_SHAPE, "mdnvae_inorm_bwd: folded gradient needs H,W >= 3");
    if (nplanes == 0) return 0;
    MV_REQUIRE(raw && stats && grad_act && grad_raw, E_ARG, "mdnvae_inorm_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
#define CALL(TPP, EPT)                                                                                               \
    do {                                                                                                             \
        dim3 grid((unsigned)((nplanes + (256 / TPP) - 1) / (256 / TPP)));                                            \
        if (folded)                                                                                                  \
            hipLaunchKernelGGL((inorm_bwd_kernel<TPP, EPT, true>), grid, dim3(256), 0, st, raw, stats, grad_act,     \
                               relu, nplanes, H, W, grad_raw, plane_gsum);                                           \
        else                                                                                                         \
            hipLaunchKernelGGL((inorm_bwd_kernel<TPP, EPT, false>), grid, dim3(256), 0, st, raw, stats, grad_act,    \
                               relu, nplanes, H, W, grad_raw, plane_gsum);                                           \
    } while (0)
    MV_PLANE_DISPATCH(P, CALL);
#undef CALL
    MV_LAUNCH_CHECK("inorm_bwd_kernel");
    return 0;
}

int mdnvae_fold_add(const float* base, const float* grad_padded, long nplanes, int H, int W, float* out,
                    void* stream) {
    MV_REQUIRE(nplanes >= 0 && H >= 3 && W >= 3, E_ARG, "mdnvae_fold_add: bad size");
    if (nplanes == 0) return 0;
    MV_REQUIRE(grad_padded && out, E_ARG, "mdnvae_fold_add: null pointer");
    hipLaunchKernelGGL(fold_add_kernel, dim3(ew_grid(nplanes * H * W)), dim3(256), 0, (hipStream_t)stream, base,
                       grad_padded, nplanes, H, W, out);
    MV_LAUNCH_CHECK("fold_add_kernel");
    return 0;
}

int mdnvae_maxpool_fwd(const float* x, long nplanes, int H, int W, int K, float* feat, int* argmax, void* stream) {
    MV_REQUIRE(nplanes >= 0 && K >= 1 && H >= K && W >= K, E_ARG, "mdnvae_maxpool_fwd: bad size");
    if (nplanes == 0) return 0;
    MV_REQUIRE(x && feat && argmax, E_ARG, "mdnvae_maxpool_fwd: null pointer");
    const int PH = H / K, PW = W / K;
    const long n = nplanes * PH * PW;
    if (K == 8 && PH == 2 && PW == 2 && (nplanes + 3) / 4 <= 0x7fffffffL)
        hipLaunchKernelGGL(maxpool8_plane_kernel, dim3((unsigned)((nplanes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                           nplanes, H, W, feat, argmax);
    else if (K * K <= 64 && (n + 3) / 4 <= 0x7fffffffL)
        hipLaunchKernelGGL(maxpool_fwd_wave_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                           nplanes, H, W, K, PH, PW, feat, argmax);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                           nplanes, H, W, K, PH, PW, feat, argmax);
    MV_LAUNCH_CHECK("maxpool_fwd_kernel");
    return 0;
}

int mdnvae_maxpool_bwd(const float* grad_feat, const int* argmax, long nplanes, int H, int W, int K, float* grad_x,
                       void* stream) {
    MV_REQUIRE(nplanes >= 0 && K >= 1 && H >= K && W >= K, E_ARG, "mdnvae_maxpool_bwd: bad size");
    if (nplanes == 0) return 0;
    MV_REQUIRE(grad_feat && argmax && grad_x, E_ARG, "mdnvae_maxpool_bwd: null pointer");
    if (K == 8 && H / K == 2 && W / K == 2 && (nplanes + 3) / 4 <= 0x7fffffffL)
        hipLaunchKernelGGL(maxpool8_bwd_plane_kernel, dim3((unsigned)((nplanes + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           grad_feat, argmax, nplanes, H * W, grad_x);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(nplanes * H * W)), dim3(256), 0, (hipStream_t)stream, grad_feat,
                           argmax, nplanes, H, W, K, H / K, W / K, grad_x);
    MV_LAUNCH_CHECK("maxpool_bwd_kernel");
    return 0;
}

}  // extern "C"
