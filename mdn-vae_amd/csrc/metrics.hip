// Batched evaluation metrics on square connectome matrices held one per workgroup in LDS.
//   spectrum: eigenvalues of a symmetric n x n matrix by cyclic Jacobi rotations in the parallel
//     (round-robin) ordering, sorted in descending order, top k kept
//       reference: compute_spectral_features, function/VAE_CNN_NN.py:302-308 (np.linalg.eigvals, real
//       part, sort descending, [:k]) -- equal to it for symmetric input, which is what the scripts feed
//       for real matrices and for the symmetrised decoders;
//   ssim: mean structural similarity, 7x7 uniform window, sample covariance, K1 = 0.01, K2 = 0.03,
//     data_range given (the defaults of the skimage call at function/VAE_CNN_NN.py:291-297).
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

constexpr int EIG_MAXN = 96;
constexpr int EIG_NT = 256;
constexpr int EIG_SWEEPS = 16;

__global__ __launch_bounds__(EIG_NT) void jacobi_spectrum_kernel(const float* __restrict__ mats, int n, int k,
                                                                 float* __restrict__ out) {
    extern __shared__ float smem[];
    const int ld = n + 1;
    float* A = smem;                         // [n][ld]
    float* cs = A + n * ld;                  // [m/2][2] rotation of each pair of the round
    __shared__ int pp[EIG_MAXN / 2], qq[EIG_MAXN / 2];
    __shared__ float red[EIG_NT / 64];
    __shared__ float off_sq, frob_sq;
    const int t = threadIdx.x;
    const float* src = mats + (long)blockIdx.x * n * n;
    for (int e = t; e < n * n; e += EIG_NT) {
        const int i = e / n, j = e - i * n;
        A[i * ld + j] = 0.5f * (src[e] + src[j * n + i]);       // exact for symmetric input
    }
    const int m = (n + 1) & ~1;              // players of the tournament (one dummy when n is odd)
    const int half = m / 2;
    __syncthreads();
    {
        float s = 0.0f;
        for (int e = t; e < n * n; e += EIG_NT) {
            const float v = A[(e / n) * ld + e % n];
            s += v * v;
        }
        s = block_sum<EIG_NT, float>(s, red);
        if (t == 0) frob_sq = s;
        __syncthreads();
    }
    for (int sweep = 0; sweep < EIG_SWEEPS; ++sweep) {
        for (int r = 0; r < m - 1; ++r) {
            // round r of the round-robin: (m-1, r) and ((r+i) mod (m-1), (r-i) mod (m-1)), i = 1..half-1
            if (t < half) {
                int p, q;
                if (t == 0) {
                    p = m - 1;
                    q = r;
                } else {
                    p = (r + t) % (m - 1);
                    q = (r - t + (m - 1)) % (m - 1);
                }
                if (p > q) { const int x = p; p = q; q = x; }
                float c = 1.0f, s = 0.0f;
                if (q < n) {                                     // a pair with the dummy player rests
                    const float apq = A[p * ld + q];
                    if (fabsf(apq) > 1e-30f) {
                        const float tau = (A[q * ld + q] - A[p * ld + p]) / (2.0f * apq);
                        const float tt = (tau >= 0.0f ? 1.0f : -1.0f) / (fabsf(tau) + sqrtf(1.0f + tau * tau));
                        c = 1.0f / sqrtf(1.0f + tt * tt);
                        s = tt * c;
                    }
                } else {
                    q = p;                                       // identity on (p, p)
                }
                pp[t] = p;
                qq[t] = q;
                cs[2 * t] = c;
                cs[2 * t + 1] = s;
            }
            __syncthreads();
            // A <- A J : columns p, q of every row
            for (int w = t; w < n * half; w += EIG_NT) {
                const int i = w / half, pr = w - i * half;
                const int p = pp[pr], q = qq[pr];
                if (p != q) {
                    const float c = cs[2 * pr], s = cs[2 * pr + 1];
                    const float ap = A[i * ld + p], aq = A[i * ld + q];
                    A[i * ld + p] = c * ap - s * aq;
                    A[i * ld + q] = s * ap + c * aq;
                }
            }
            __syncthreads();
            // A <- J^T A : rows p, q of every column
            for (int w = t; w < n * half; w += EIG_NT) {
                const int pr = w / n, j = w - pr * n;
                const int p = pp[pr], q = qq[pr];
                if (p != q) {
                    const float c = cs[2 * pr], s = cs[2 * pr + 1];
                    const float ap = A[p * ld + j], aq = A[q * ld + j];
                    A[p * ld + j] = c * ap - s * aq;
                    A[q * ld + j] = s * ap + c * aq;
                }
            }
            __syncthreads();
        }
        float s = 0.0f;
        for (int e = t; e < n * n; e += EIG_NT) {
            const int i = e / n, j = e - i * n;
            if (i != j) s += A[i * ld + j] * A[i * ld + j];
        }
        s = block_sum<EIG_NT, float>(s, red);
        if (t == 0) off_sq = s;
        __syncthreads();
        if (off_sq <= 1e-14f * frob_sq) break;                   // uniform over the workgroup
    }
    // descending rank of each diagonal entry (ties broken by index)
    for (int i = t; i < n; i += EIG_NT) {
        const float v = A[i * ld + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float u = A[j * ld + j];
            rank += (u > v) || (u == v && j < i);
        }
        if (rank < k) out[(long)blockIdx.x * k + rank] = v;
    }
}

// one workgroup per image pair; both images in LDS; each thread walks valid window positions
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                   float data_range, float* __restrict__ out) {
    extern __shared__ float smem[];
    __shared__ double red[4];
    constexpr int WIN = 7;
    float* ia = smem;
    float* ib = smem + n * n;
    const int t = threadIdx.x;
    for (int e = t; e < n * n; e += 256) {
        ia[e] = a[(long)blockIdx.x * n * n + e];
        ib[e] = b[(long)blockIdx.x * n * n + e];
    }
    __syncthreads();
    const int v = n - WIN + 1;                                   // valid positions per axis
    const float np_ = (float)(WIN * WIN), cov_norm = np_ / (np_ - 1.0f);
    const float c1 = (0.01f * data_range) * (0.01f * data_range), c2 = (0.03f * data_range) * (0.03f * data_range);
    double acc = 0.0;
    for (int e = t; e < v * v; e += 256) {
        const int y0 = e / v, x0 = e - y0 * v;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
        for (int dy = 0; dy < WIN; ++dy)
#pragma unroll
            for (int dx = 0; dx < WIN; ++dx) {
                const float p = ia[(y0 + dy) * n + x0 + dx], q = ib[(y0 + dy) * n + x0 + dx];
                sx += p; sy += q; sxx += p * p; syy += q * q; sxy += p * q;
            }
        const float ux = sx / np_, uy = sy / np_;
        const float vx = cov_norm * (sxx / np_ - ux * ux), vy = cov_norm * (syy / np_ - uy * uy);
        const float vxy = cov_norm * (sxy / np_ - ux * uy);
        const float s = ((2.0f * ux * uy + c1) * (2.0f * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
        acc += (double)s;
    }
    acc = block_sum<256, double>(acc, red);
    if (t == 0) out[blockIdx.x] = (float)(acc / ((double)v * v));
}

// Graph features of one weighted graph per workgroup (n <= 128 nodes):
//   [weighted-degree mean, std, weighted clustering mean, std, average shortest path length]
//   reference: compute_graph_features, validation/FID.py:6-58 (weighted = 1: edge distance
//   1/(w + 1e-5)) and function/VAE_CNN_NN.py:341-379 (weighted = 0: the script attaches no distance,
//   so paths count hops).  The networkx calls behind it are restated: an edge wherever either
//   triangle entry of matrix/max is non-zero (weight = the lower entry if non-zero, else the upper),
//   a self-loop counts twice in the degree, Onnela clustering on weight / max edge weight,
//   all-pairs shortest paths (Floyd-Warshall in LDS), per-component averages when disconnected.
constexpr int GF_MAXN = 128;
__global__ __launch_bounds__(256) void graph_features_kernel(const float* __restrict__ mats, int n, int weighted,
                                                             float* __restrict__ out) {
    extern __shared__ float smem[];
    const int ld = n + 1;
    float* W = smem;                         // [n][ld] symmetric edge weights (0 = no edge)
    float* D = W + n * ld;                   // [n][ld] distances
    __shared__ float deg[GF_MAXN], clu[GF_MAXN], part[GF_MAXN][4];
    __shared__ int comp[GF_MAXN];
    __shared__ float redf[4];
    __shared__ double redd[4];
    __shared__ float s_max, s_wmax;
    const int t = threadIdx.x;
    const float* A = mats + (long)blockIdx.x * n * n;
    // matrix.max()
    {
        float m = -INFINITY;
        for (int e = t; e < n * n; e += 256) m = fmaxf(m, A[e]);
        m = wave_max_all(m);
        if ((t & 63) == 0) redf[t >> 6] = m;
        __syncthreads();
        if (t == 0) s_max = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
        __syncthreads();
    }
    const float mx = s_max;
    float wm = -INFINITY;
    for (int e = t; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        const int hi = max(i, j), lo = min(i, j);
        float w = 0.0f, up = 0.0f;
        if (mx > 0.0f) {
            const float lower = A[hi * n + lo] / mx;
            up = A[lo * n + hi] / mx;
            w = (lower != 0.0f) ? lower : up;
        }
        W[i * ld + j] = w;
        if (w != 0.0f) wm = fmaxf(wm, w);
        float dist = INFINITY;
        if (i == j) dist = 0.0f;
        else if (w != 0.0f) dist = weighted ? 1.0f / (up + 1e-5f) : 1.0f;
        D[i * ld + j] = dist;
    }
    wm = wave_max_all(wm);
    if ((t & 63) == 0) redf[t >> 6] = wm;
    __syncthreads();
    if (t == 0) {
        const float v = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
        s_wmax = (v == -INFINITY) ? 1.0f : v;
    }
    __syncthreads();
    const float wmax = s_wmax;
    // weighted degree
    if (t < n) {
        float s = 0.0f;
        for (int j = 0; j < n; ++j)
            if (j != t) s += W[t * ld + j];
        deg[t] = s + 2.0f * W[t * ld + t];
    }
    // clustering: thread (i, p) takes the neighbours j = p, p + P, ... of node i; pairs j < k
    const int P = (256 / n) < 4 ? (256 / n) : 4;          // 1..4 threads per node
    {
        const int i = t / P, p = t - i * P;
        if (i < n) {
            float s = 0.0f;
            for (int j = p; j < n; j += P) {
                const float wij = W[i * ld + j];
                if (j == i || wij == 0.0f) continue;
                const float hij = wij / wmax;
                for (int k = j + 1; k < n; ++k) {
                    const float wik = W[i * ld + k], wjk = W[j * ld + k];
                    if (k == i || wik == 0.0f || wjk == 0.0f) continue;
                    s += cbrtf(hij * (wjk / wmax) * (wik / wmax));
                }
            }
            part[i][p] = s;
        }
    }
    __syncthreads();
    if (t < n) {
        float tri = 0.0f;
        for (int p = 0; p < P; ++p) tri += part[t][p];
        int d = 0;
        for (int j = 0; j < n; ++j) d += (j != t && W[t * ld + j] != 0.0f);
        tri *= 2.0f;
        clu[t] = (tri == 0.0f) ? 0.0f : tri / ((float)d * (float)(d - 1));
    }
    // Floyd-Warshall: row k and column k do not change in step k, so the update is done in place
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        for (int e = t; e < n * n; e += 256) {
            const int i = e / n, j = e - i * n;
            const float via = D[i * ld + k] + D[k * ld + j];
            if (via < D[i * ld + j]) D[i * ld + j] = via;
        }
    }
    __syncthreads();
    if (t < n) {
        int c = t;
        for (int j = 0; j < n; ++j)
            if (D[t * ld + j] < INFINITY) { c = j; break; }
        comp[t] = c;
    }
    __syncthreads();
    // moments of degree and clustering (population std), in double
    double s1 = 0.0, s2 = 0.0, c1 = 0.0, c2 = 0.0;
    if (t < n) {
        s1 = deg[t]; s2 = (double)deg[t] * deg[t];
        c1 = clu[t]; c2 = (double)clu[t] * clu[t];
    }
    s1 = block_sum<256, double>(s1, redd);
    s2 = block_sum<256, double>(s2, redd);
    c1 = block_sum<256, double>(c1, redd);
    c2 = block_sum<256, double>(c2, redd);
    // per-node sum of distances inside its component (deg[] is free now)
    if (t < n) {
        float s = 0.0f;
        for (int j = 0; j < n; ++j)
            if (comp[j] == comp[t]) s += D[t * ld + j];
        deg[t] = s;
    }
    __syncthreads();
    if (t == 0) {
        float* o = out + (long)blockIdx.x * 5;
        const double dm = s1 / n, cm = c1 / n;
        o[0] = (float)dm;
        o[1] = (float)sqrt(fmax(s2 / n - dm * dm, 0.0));
        o[2] = (float)cm;
        o[3] = (float)sqrt(fmax(c2 / n - cm * cm, 0.0));
        double total = 0.0;
        int ncomp = 0, counted = 0;
        for (int r = 0; r < n; ++r) {
            if (comp[r] != r) continue;
            ++ncomp;
            int size = 0;
            double s = 0.0;
            for (int j = 0; j < n; ++j)
                if (comp[j] == r) { ++size; s += deg[j]; }
            if (size > 1) { total += s / ((double)size * (size - 1)); ++counted; }
        }
        o[4] = counted ? (float)(total / counted) : 0.0f;      // one component: its own average
        (void)ncomp;
    }
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

int mdnvae_spectrum_topk(const float* mats, long count, int n, int k, float* out, void* stream) {
    MV_REQUIRE(count >= 0 && n >= 2 && k >= 1 && k <= n, E_ARG, "mdnvae_spectrum_topk: bad size (n=%d, k=%d)", n, k);
    MV_REQUIRE(n <= EIG_MAXN, E_SHAPE, "mdnvae_spectrum_topk: n=%d exceeds %d", n, EIG_MAXN);
    MV_REQUIRE(count <= 0x7fffffffL, E_SHAPE, "mdnvae_spectrum_topk: too many matrices for one launch");
    if (count == 0) return 0;
    MV_REQUIRE(mats && out, E_ARG, "mdnvae_spectrum_topk: null pointer");
    const size_t lds = sizeof(float) * ((size_t)n * (n + 1) + (size_t)((n + 1) & ~1));
    hipLaunchKernelGGL(jacobi_spectrum_kernel, dim3((unsigned)count), dim3(EIG_NT), lds, (hipStream_t)stream, mats, n, k,
                       out);
    MV_LAUNCH_CHECK("jacobi_spectrum_kernel");
    return 0;
}

int mdnvae_ssim(const float* a, const float* b, long count, int n, float data_range, float* out, void* stream) {
    MV_REQUIRE(count >= 0 && n >= 7, E_ARG, "mdnvae_ssim: images must be at least 7x7 (n=%d)", n);
    MV_REQUIRE(n <= 128, E_SHAPE, "mdnvae_ssim: n=%d exceeds 128", n);
    MV_REQUIRE(data_range > 0.0f, E_ARG, "mdnvae_ssim: data_range must be positive");
    MV_REQUIRE(count <= 0x7fffffffL, E_SHAPE, "mdnvae_ssim: too many pairs for one launch");
    if (count == 0) return 0;
    MV_REQUIRE(a && b && out, E_ARG, "mdnvae_ssim: null pointer");
    {
        // two 128 x 128 images; the kernel's static LDS comes on top of this
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssim_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_ssim: cannot raise the LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(ssim_kernel, dim3((unsigned)count), dim3(256), sizeof(float) * 2 * n * n, (hipStream_t)stream, a, b,
                       n, data_range, out);
    MV_LAUNCH_CHECK("ssim_kernel");
    return 0;
}

int mdnvae_graph_features(const float* mats, long count, int n, int weighted_paths, float* out, void* stream) {
    MV_REQUIRE(count >= 0 && n >= 2, E_ARG, "mdnvae_graph_features: bad size (n=%d)", n);
    MV_REQUIRE(n <= GF_MAXN, E_SHAPE, "mdnvae_graph_features: n=%d exceeds %d", n, GF_MAXN);
    MV_REQUIRE(count <= 0x7fffffffL, E_SHAPE, "mdnvae_graph_features: too many matrices for one launch");
    if (count == 0) return 0;
    MV_REQUIRE(mats && out, E_ARG, "mdnvae_graph_features: null pointer");
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(graph_features_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
        MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_graph_features: cannot raise the LDS limit: %s",
                   hipGetErrorString(e));
    }
    const size_t lds = sizeof(float) * (size_t)2 * n * (n + 1);
    hipLaunchKernelGGL(graph_features_kernel, dim3((unsigned)count), dim3(256), lds, (hipStream_t)stream, mats, n,
                       weighted_paths, out);
    MV_LAUNCH_CHECK("graph_features_kernel");
    return 0;
}

}  // extern "C"
