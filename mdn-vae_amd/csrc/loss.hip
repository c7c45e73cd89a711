// Fused reparameterisation + KL + reconstruction-loss kernels (forward, backward and the
// one-pass forward+gradient form).  HBM-streaming: recon and x are read as one flat array
// of B*D floats with 16-byte non-temporal loads, 4 loads in flight per lane; sums are reduced
// lane -> wave (shuffle) -> workgroup (LDS) -> one double per workgroup, and a second
// one-workgroup launch adds the per-workgroup partials in a fixed order, so the loss is
// bit-reproducible run to run.
//
// Reference semantics:
//   reparameterize  function/VAE_CNN_NN.py:164-167   z = mu + eps*exp(0.5*logvar)
//   loss_function   function/VAE_CNN_NN.py:197-200   poisson (rate input) + KL
//                   function/MM_Gate.py:179-182      poisson (log input)  + KL
//                   function_FC/VAE_CNN_GCN.py:207-211  squared error     + KL
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

constexpr int LOSS_NT = 256;          // threads per workgroup
#ifndef LOSS_UNROLL_N
#define LOSS_UNROLL_N 2
#endif
#ifndef LOSS_MAX_BLOCKS_N
#define LOSS_MAX_BLOCKS_N 1024
#endif
#ifndef LOSS_NT_LOADS
#define LOSS_NT_LOADS 1
#endif
// Measured on MI355X at B=8192, N=87 (tools/loss_tune.sh): 1024 workgroups x 2 float4 per lane per
// array with non-temporal accesses streams at 6.4-6.5 TB/s; more bytes in flight per lane, more
// workgroups, or default-policy loads are all slower (5.3 TB/s at 2048 x 4, default policy).
constexpr int LOSS_UNROLL = LOSS_UNROLL_N;         // float4 chunks per lane per array per tile
constexpr int LOSS_MAX_BLOCKS = LOSS_MAX_BLOCKS_N; // 4 workgroups per CU

__device__ __forceinline__ float4 stream_load(const float4* p) {
#if LOSS_NT_LOADS
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void stream_store(float4* p, float4 v) {
#if LOSS_NT_LOADS
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
#else
    *p = v;
#endif
}

template <int MODE>
__device__ __forceinline__ float recon_value(float r, float x) {
    if (MODE == MDNVAE_RECON_POISSON_RATE) return r - x * logf(r + 1e-8f);
    if (MODE == MDNVAE_RECON_POISSON_LOG) return expf(r) - x * r;
    const float d = r - x;
    return d * d;
}
template <int MODE>
__device__ __forceinline__ float recon_slope(float r, float x) {
    if (MODE == MDNVAE_RECON_POISSON_RATE) return 1.0f - x / (r + 1e-8f);
    if (MODE == MDNVAE_RECON_POISSON_LOG) return expf(r) - x;
    return 2.0f * (r - x);
}

// One pass over recon/x.  WANT_LOSS: accumulate the reconstruction term; WANT_GRAD: write
// g * d(term)/d(recon).  The latent part (B*L elements) is done by the same grid after the
// stream: z (if asked for), the KL term, and the latent gradients.
template <int MODE, bool WANT_LOSS, bool WANT_GRAD>
__global__ __launch_bounds__(LOSS_NT) void loss_stream_kernel(
    const float* __restrict__ recon, const float* __restrict__ x, float* __restrict__ grad_recon,
    long n, const float* __restrict__ mu, const float* __restrict__ logvar,
    const float* __restrict__ eps, float* __restrict__ z, const float* __restrict__ grad_z,
    float* __restrict__ grad_mu, float* __restrict__ grad_logvar, long nl,
    const float* __restrict__ gscale_ptr, double* __restrict__ partials) {
    __shared__ double red[LOSS_NT / 64];
    // gscale_ptr[0] scales the reconstruction term's gradient, gscale_ptr[1] the KL term's
    const float g = (WANT_GRAD && gscale_ptr) ? gscale_ptr[0] : 1.0f;
    const float gk = (WANT_GRAD && gscale_ptr) ? gscale_ptr[1] : 1.0f;
    float acc = 0.0f;

    const long n4 = n >> 2;
    const float4* r4 = reinterpret_cast<const float4*>(recon);
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* g4 = reinterpret_cast<float4*>(grad_recon);
    constexpr long TILE = (long)LOSS_NT * LOSS_UNROLL;
    for (long base = (long)blockIdx.x * TILE; base < n4; base += (long)gridDim.x * TILE) {
        float4 rv[LOSS_UNROLL], xv[LOSS_UNROLL];
#pragma unroll
        for (int u = 0; u < LOSS_UNROLL; ++u) {
            const long i = base + u * LOSS_NT + threadIdx.x;
            if (i < n4) {
                rv[u] = stream_load(r4 + i);
                xv[u] = stream_load(x4 + i);
            }
        }
#pragma unroll
        for (int u = 0; u < LOSS_UNROLL; ++u) {
            const long i = base + u * LOSS_NT + threadIdx.x;
            if (i < n4) {
                if (WANT_LOSS) {
                    acc += (recon_value<MODE>(rv[u].x, xv[u].x) + recon_value<MODE>(rv[u].y, xv[u].y)) +
                           (recon_value<MODE>(rv[u].z, xv[u].z) + recon_value<MODE>(rv[u].w, xv[u].w));
                }
                if (WANT_GRAD) {
                    float4 o;
                    o.x = g * recon_slope<MODE>(rv[u].x, xv[u].x);
                    o.y = g * recon_slope<MODE>(rv[u].y, xv[u].y);
                    o.z = g * recon_slope<MODE>(rv[u].z, xv[u].z);
                    o.w = g * recon_slope<MODE>(rv[u].w, xv[u].w);
                    stream_store(g4 + i, o);
                }
            }
        }
    }
    // the (at most 3) elements past the last whole float4
    const long gtid = (long)blockIdx.x * LOSS_NT + threadIdx.x;
    const long tail0 = n4 << 2;
    if (gtid < n - tail0) {
        const float r = recon[tail0 + gtid], xx = x[tail0 + gtid];
        if (WANT_LOSS) acc += recon_value<MODE>(r, xx);
        if (WANT_GRAD) grad_recon[tail0 + gtid] = g * recon_slope<MODE>(r, xx);
    }

    // latent part
    float kl = 0.0f;
    const long nthreads = (long)gridDim.x * LOSS_NT;
    for (long i = gtid; i < nl; i += nthreads) {
        const float m = mu[i], lv = logvar[i];
        const float var = expf(lv);
        if (WANT_LOSS) kl += -0.5f * (1.0f + lv - m * m - var);
        float sd = 0.0f, e = 0.0f;
        if (eps) {
            sd = expf(0.5f * lv);
            e = eps[i];
            if (z) z[i] = m + e * sd;
        }
        if (WANT_GRAD && grad_mu) {
            float gm = gk * m, gl = gk * 0.5f * (var - 1.0f);
            if (grad_z) {
                const float gz = grad_z[i];
                gm += gz;
                gl += gz * e * 0.5f * sd;
            }
            grad_mu[i] = gm;
            grad_logvar[i] = gl;
        }
    }
    if (WANT_LOSS) {
        const double a = block_sum<LOSS_NT, double>((double)acc, red);
        const double k = block_sum<LOSS_NT, double>((double)kl, red);
        if (threadIdx.x == 0) {
            partials[2 * blockIdx.x] = a;
            partials[2 * blockIdx.x + 1] = k;
        }
    }
}

// out[0] = recon term + KL, out[1] = recon term, out[2] = KL
__global__ __launch_bounds__(PAIR_NT) void loss_finalize_kernel(const double* __restrict__ partials, int nblocks,
                                                            float* __restrict__ out) {
    __shared__ double red[PAIR_NT / 64];
    const PairSum t = block_pair_sum(partials, nblocks, red);
    if (threadIdx.x == 0) {
        out[0] = (float)(t.a + t.k);
        out[1] = (float)t.a;
        out[2] = (float)t.k;
    }
}

__global__ __launch_bounds__(256) void reparam_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                      const float* __restrict__ eps, float* __restrict__ z, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) z[i] = mu[i] + eps[i] * expf(0.5f * logvar[i]);
}

// grad_mu = gz (+ existing), grad_logvar = gz*eps*0.5*exp(0.5*logvar)
__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* __restrict__ logvar, const float* __restrict__ eps,
                                                          const float* __restrict__ grad_z, float* __restrict__ grad_mu,
                                                          float* __restrict__ grad_logvar, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float gz = grad_z[i];
        grad_mu[i] = gz;
        grad_logvar[i] = gz * eps[i] * 0.5f * expf(0.5f * logvar[i]);
    }
}

static int loss_grid(long n) {
    const long tiles = ((n >> 2) + (long)LOSS_NT * LOSS_UNROLL - 1) / ((long)LOSS_NT * LOSS_UNROLL);
    long g = tiles < 1 ? 1 : tiles;
    if (g > LOSS_MAX_BLOCKS) g = LOSS_MAX_BLOCKS;
    return (int)g;
}

template <bool WANT_LOSS, bool WANT_GRAD>
static int launch_loss(int mode, const float* recon, const float* x, float* grad_recon, long n, const float* mu,
                       const float* logvar, const float* eps, float* z, const float* grad_z, float* grad_mu,
                       float* grad_logvar, long nl, const float* gscale, double* partials, float* out,
                       hipStream_t st) {
    const int grid = loss_grid(n);
#define MV_LOSS_CASE(M)                                                                                        \
    case M:                                                                                                    \
        hipLaunchKernelGGL((loss_stream_kernel<M, WANT_LOSS, WANT_GRAD>), dim3(grid), dim3(LOSS_NT), 0, st,   \
                           recon, x, grad_recon, n, mu, logvar, eps, z, grad_z, grad_mu, grad_logvar, nl,     \
                           gscale, partials);                                                                  \
        break;
    switch (mode) {
        MV_LOSS_CASE(MDNVAE_RECON_POISSON_RATE)
        MV_LOSS_CASE(MDNVAE_RECON_POISSON_LOG)
        MV_LOSS_CASE(MDNVAE_RECON_MSE)
    }
#undef MV_LOSS_CASE
    MV_LAUNCH_CHECK("loss_stream_kernel");
    if (WANT_LOSS) {
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(PAIR_NT), 0, st, partials, grid, out);
        MV_LAUNCH_CHECK("loss_finalize_kernel");
    }
    return 0;
}

static int check_common(const char* fn, int mode, const void* recon, const void* x, long B, long D, long L,
                        const void* mu, const void* logvar) {
    MV_REQUIRE(mode >= 0 && mode <= MDNVAE_RECON_MSE, E_ARG, "%s: unknown reconstruction mode %d", fn, mode);
    MV_REQUIRE(B >= 0 && D >= 0 && L >= 0, E_ARG, "%s: negative size", fn);
    MV_REQUIRE(B * D == 0 || (recon && x), E_ARG, "%s: recon/x is null", fn);
    MV_REQUIRE(B * L == 0 || (mu && logvar), E_ARG, "%s: mu/logvar is null", fn);
    MV_REQUIRE(aligned16(recon) && aligned16(x), E_ALIGN, "%s: recon and x must be 16-byte aligned", fn);
    return 0;
}

}  // namespace mdnvae

using namespace mdnvae;

extern "C" {

size_t mdnvae_loss_workspace_bytes(void) { return sizeof(double) * 2 * LOSS_MAX_BLOCKS; }

int mdnvae_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long n, void* stream) {
    MV_REQUIRE(n >= 0, E_ARG, "mdnvae_reparam_fwd: negative size");
    if (n == 0) return 0;
    MV_REQUIRE(mu && logvar && eps && z, E_ARG, "mdnvae_reparam_fwd: null pointer");
    hipLaunchKernelGGL(reparam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu, logvar,
                       eps, z, n);
    MV_LAUNCH_CHECK("reparam_kernel");
    return 0;
}

int mdnvae_reparam_bwd(const float* logvar, const float* eps, const float* grad_z, float* grad_mu,
                       float* grad_logvar, long n, void* stream) {
    MV_REQUIRE(n >= 0, E_ARG, "mdnvae_reparam_bwd: negative size");
    if (n == 0) return 0;
    MV_REQUIRE(logvar && eps && grad_z && grad_mu && grad_logvar, E_ARG, "mdnvae_reparam_bwd: null pointer");
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       logvar, eps, grad_z, grad_mu, grad_logvar, n);
    MV_LAUNCH_CHECK("reparam_bwd_kernel");
    return 0;
}

int mdnvae_vae_loss_fwd(int mode, const float* recon, const float* x, long B, long D, const float* mu,
                        const float* logvar, long L, const float* eps, float* z, void* workspace, float* loss_out,
                        void* stream) {
    if (int rc = check_common("mdnvae_vae_loss_fwd", mode, recon, x, B, D, L, mu, logvar)) return rc;
    MV_REQUIRE(workspace && loss_out, E_ARG, "mdnvae_vae_loss_fwd: workspace/loss_out is null");
    MV_REQUIRE(!z || eps, E_ARG, "mdnvae_vae_loss_fwd: z requested without eps");
    return launch_loss<true, false>(mode, recon, x, nullptr, B * D, mu, logvar, eps, z, nullptr, nullptr, nullptr,
                                    B * L, nullptr, (double*)workspace, loss_out, (hipStream_t)stream);
}

int mdnvae_vae_loss_bwd(int mode, const float* recon, const float* x, long B, long D, const float* mu,
                        const float* logvar, long L, const float* eps, const float* grad_z,
                        const float* grad_loss, float* grad_recon, float* grad_mu, float* grad_logvar,
                        void* stream) {
    if (int rc = check_common("mdnvae_vae_loss_bwd", mode, recon, x, B, D, L, mu, logvar)) return rc;
    MV_REQUIRE(B * D == 0 || grad_recon, E_ARG, "mdnvae_vae_loss_bwd: grad_recon is null");
    MV_REQUIRE(aligned16(grad_recon), E_ALIGN, "mdnvae_vae_loss_bwd: grad_recon must be 16-byte aligned");
    MV_REQUIRE(B * L == 0 || (grad_mu && grad_logvar), E_ARG, "mdnvae_vae_loss_bwd: grad_mu/grad_logvar is null");
    MV_REQUIRE(!grad_z || eps, E_ARG, "mdnvae_vae_loss_bwd: grad_z given without eps");
    return launch_loss<false, true>(mode, recon, x, grad_recon, B * D, mu, logvar, eps, nullptr, grad_z, grad_mu,
                                    grad_logvar, B * L, grad_loss, nullptr, nullptr, (hipStream_t)stream);
}

int mdnvae_vae_loss_fwdbwd(int mode, const float* recon, const float* x, long B, long D, const float* mu,
                           const float* logvar, long L, float* grad_recon, float* grad_mu, float* grad_logvar,
                           void* workspace, float* loss_out, void* stream) {
    if (int rc = check_common("mdnvae_vae_loss_fwdbwd", mode, recon, x, B, D, L, mu, logvar)) return rc;
    MV_REQUIRE(workspace && loss_out, E_ARG, "mdnvae_vae_loss_fwdbwd: workspace/loss_out is null");
    MV_REQUIRE(B * D == 0 || grad_recon, E_ARG, "mdnvae_vae_loss_fwdbwd: grad_recon is null");
    MV_REQUIRE(aligned16(grad_recon), E_ALIGN, "mdnvae_vae_loss_fwdbwd: grad_recon must be 16-byte aligned");
    MV_REQUIRE(B * L == 0 || (grad_mu && grad_logvar), E_ARG, "mdnvae_vae_loss_fwdbwd: grad_mu/grad_logvar is null");
    return launch_loss<true, true>(mode, recon, x, grad_recon, B * D, mu, logvar, nullptr, nullptr, nullptr, grad_mu,
                                   grad_logvar, B * L, nullptr, (double*)workspace, loss_out, (hipStream_t)stream);
}

}  // extern "C"
