// Adam over one flat parameter buffer (torch.optim.Adam defaults as used by the reference:
// function/VAE_CNN_NN.py:204, lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight decay, no
// amsgrad), same operation order as torch's single-tensor update:
//   m <- m + (1-b1)(g-m);  v <- b2*v + (1-b2) g*g;  p <- p - (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

// The step count of a captured (graph-replayed) step lives on the device: this one-thread kernel
// advances it and leaves {lr / (1 - b1^t), sqrt(1 - b2^t)} for the update kernel.
__global__ void adam_advance_kernel(int* __restrict__ step, float lr, float b1, float b2, float* __restrict__ coef) {
    const int t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    coef[0] = (float)((double)lr / bc1);
    coef[1] = (float)sqrt(bc2);
}

// one element's update; shared by the flat and the folded kernel so that both round alike
__device__ __forceinline__ float adam_update(float gg, float& mm, float& vv, float b1, float b2, float eps,
                                             float step_size, float bc2_sqrt, float grad_scale) {
    gg *= grad_scale;
    mm = mm + (1.0f - b1) * (gg - mm);
    vv = vv * b2 + (1.0f - b2) * gg * gg;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    return step_size * (mm / denom);
}

// Adam for the weight [H][N*N] of a symmetric-input first layer whose gradient exists only folded
// onto the upper triangle, gt [t][H] (w_ij and w_ji share one gradient value, hence one first and
// one second moment and one step): the moments are kept in the folded layout and both entries take
// the step - the same numbers as unfolding the gradient and running the flat kernel, without
// writing, zeroing and re-reading the unfolded gradient and with half the moment traffic.
// Two passes: this kernel advances the folded moments and overwrites each gradient value with its
// step; apply_folded_step_kernel then subtracts the step of (min(i,j), max(i,j)) from every entry
// of the weight.  (One fused pass would let the workgroup of a lower-triangle entry read moments
// that the workgroup of its mirror image has already advanced.)
__global__ __launch_bounds__(256) void adam_folded_moments_kernel(float* __restrict__ g, float* __restrict__ m,
                                                                  float* __restrict__ v, long n, float b1, float b2,
                                                                  float eps, float step_size, float bc2_sqrt,
                                                                  float grad_scale, const float* __restrict__ coef) {
    if (coef) {
        step_size = coef[0];
        bc2_sqrt = coef[1];
    }
    const long n4 = n >> 2;
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 gv = g4[i], mv = m4[i], vv = v4[i];
        gv.x = adam_update(gv.x, mv.x, vv.x, b1, b2, eps, step_size, bc2_sqrt, grad_scale);
        gv.y = adam_update(gv.y, mv.y, vv.y, b1, b2, eps, step_size, bc2_sqrt, grad_scale);
        gv.z = adam_update(gv.z, mv.z, vv.z, b1, b2, eps, step_size, bc2_sqrt, grad_scale);
        gv.w = adam_update(gv.w, mv.w, vv.w, b1, b2, eps, step_size, bc2_sqrt, grad_scale);
        g4[i] = gv;
        m4[i] = mv;
        v4[i] = vv;
    }
    const long t = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) g[t] = adam_update(g[t], m[t], v[t], b1, b2, eps, step_size, bc2_sqrt, grad_scale);
}

// w[h][i*N+j] -= step[t(min(i,j), max(i,j))][h]: a workgroup owns 64 entries p of a weight row for
// 64 output units h; lanes run along h while the steps are read (their rows are contiguous in h),
// the values cross a 64 x 64 LDS tile, and lanes run along p while the weight is updated.
__global__ __launch_bounds__(256) void apply_folded_step_kernel(float* __restrict__ w, const float* __restrict__ step,
                                                                int H, int N) {
    __shared__ float tile[64][65];          // [p][h]
    const int P = N * N;
    const int p0 = blockIdx.x * 64, h0 = blockIdx.y * 64;
    const int a = threadIdx.x & 63, b4 = threadIdx.x >> 6;
    for (int r = b4; r < 64; r += 4) {
        const int p = p0 + r, h = h0 + a;
        if (p < P && h < H) {
            const int i = p / N, j = p - i * N;
            const int lo = i <= j ? i : j, hi = i <= j ? j : i;
            tile[r][a] = step[(long)(lo * N - ((lo * (lo - 1)) >> 1) + (hi - lo)) * H + h];
        }
    }
    __syncthreads();
    for (int r = b4; r < 64; r += 4) {
        const int h = h0 + r, p = p0 + a;
        if (h < H && p < P) {
            const long o = (long)h * P + p;
            w[o] = w[o] - tile[a][r];
        }
    }
}

// coef (device, {step size, sqrt of the second bias correction}) overrides the two scalars when given
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float b1,
                                                   float b2, float eps, float step_size, float bc2_sqrt,
                                                   float grad_scale, const float* __restrict__ coef) {
    if (coef) {
        step_size = coef[0];
        bc2_sqrt = coef[1];
    }
    const long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    const long stride = (long)gridDim.x * 256;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        pp = pp - adam_update(gg, mm, vv, b1, b2, eps, step_size, bc2_sqrt, grad_scale);
    };
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 pv = p4[i], mv = m4[i], vv = v4[i];
        const float4 gv = g4[i];
        upd(pv.x, gv.x, mv.x, vv.x);
        upd(pv.y, gv.y, mv.y, vv.y);
        upd(pv.z, gv.z, mv.z, vv.z);
        upd(pv.w, gv.w, mv.w, vv.w);
        p4[i] = pv;
        m4[i] = mv;
        v4[i] = vv;
    }
    const long t = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) upd(p[t], g[t], m[t], v[t]);
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" int mdnvae_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, int step,
                                float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    MV_REQUIRE(n >= 0 && step >= 1, E_ARG, "mdnvae_adam_step: bad size or step");
    if (n == 0) return 0;
    MV_REQUIRE(param && grad && exp_avg && exp_avg_sq, E_ARG, "mdnvae_adam_step: null pointer");
    MV_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq), E_ALIGN,
               "mdnvae_adam_step: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    long grid = ((n >> 2) + 255) / 256;
    if (grid < 1) grid = 1;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale, (const float*)nullptr);
    MV_LAUNCH_CHECK("adam_kernel");
    return 0;
}

extern "C" int mdnvae_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                                    int* step_dev, float* coef_dev, float lr, float beta1, float beta2, float eps,
                                    float grad_scale, void* stream) {
    MV_REQUIRE(n >= 0, E_ARG, "mdnvae_adam_step_dev: bad size");
    MV_REQUIRE(step_dev && coef_dev, E_ARG, "mdnvae_adam_step_dev: the step counter and coefficient pair are device memory");
    if (n > 0) {
        MV_REQUIRE(param && grad && exp_avg && exp_avg_sq, E_ARG, "mdnvae_adam_step_dev: null pointer");
        MV_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq), E_ALIGN,
                   "mdnvae_adam_step_dev: buffers must be 16-byte aligned");
    }
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, lr, beta1, beta2, coef_dev);
    MV_LAUNCH_CHECK("adam_advance_kernel");
    if (n == 0) return 0;
    long grid = ((n >> 2) + 255) / 256;
    if (grid < 1) grid = 1;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, beta1, beta2, eps, 0.0f, 1.0f, grad_scale, (const float*)coef_dev);
    MV_LAUNCH_CHECK("adam_kernel");
    return 0;
}

extern "C" int mdnvae_adam_step_folded(float* param, float* grad_t, float* exp_avg_t, float* exp_avg_sq_t, int H,
                                       int N, int step, const float* coef_dev, float lr, float beta1, float beta2,
                                       float eps, float grad_scale, void* stream) {
    MV_REQUIRE(H >= 1 && N >= 1 && (long)N * N <= 0x7fffffffL, E_ARG, "mdnvae_adam_step_folded: bad shape");
    MV_REQUIRE(coef_dev || step >= 1, E_ARG, "mdnvae_adam_step_folded: step must be >= 1 without device coefficients");
    MV_REQUIRE(param && grad_t && exp_avg_t && exp_avg_sq_t, E_ARG, "mdnvae_adam_step_folded: null pointer");
    MV_REQUIRE(aligned16(grad_t) && aligned16(exp_avg_t) && aligned16(exp_avg_sq_t), E_ALIGN,
               "mdnvae_adam_step_folded: folded buffers must be 16-byte aligned");
    float step_size = 0.0f, bc2_sqrt = 1.0f;
    if (!coef_dev) {
        const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }
    const long n = (long)N * (N + 1) / 2 * H;
    long blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_folded_moments_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad_t,
                       exp_avg_t, exp_avg_sq_t, n, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale, coef_dev);
    MV_LAUNCH_CHECK("adam_folded_moments_kernel");
    const dim3 grid((unsigned)(((long)N * N + 63) / 64), (unsigned)((H + 63) / 64));
    MV_REQUIRE(grid.y <= 65535, E_SHAPE, "mdnvae_adam_step_folded: too many output units for one launch");
    hipLaunchKernelGGL(apply_folded_step_kernel, grid, dim3(256), 0, (hipStream_t)stream, param, grad_t, H, N);
    MV_LAUNCH_CHECK("apply_folded_step_kernel");
    return 0;
}
