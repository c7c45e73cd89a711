// FID validation statistics on the GPU.
//   reference: np.mean / np.cov(rowvar=False) / scipy.linalg.sqrtm / np.trace in calculate_fid,
//   function/VAE_CNN_NN.py:310-316, 331-337 and validation/FID.py:61-81.
// Moments: the shifted Gram matrix sum_rows (x-c)(x-c)^T is an MFMA GEMM (fp32 32x32x2 tiles,
// K = rows) over chunks of 1024 rows whose partial tiles are added in float64 in a fixed order;
// only the 128 x 128 tiles on or above the diagonal are computed and kept.  Covariance and mean
// follow in float64.
// The matrix square root is a coupled Newton-Schulz iteration on symmetric PSD matrices, run by
// the host as float64 library GEMMs; this file supplies its elementwise steps
// (T = 1.5 I - 0.5 Z Y, symmetrise, trace).
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int GR_ROWS = 1024;     // rows (K) per chunk: fp32 accumulation inside, float64 across chunks
constexpr int GR_T = 128;         // output tile edge
constexpr int GR_KC = 16;         // rows staged per LDS buffer
constexpr int GR_LD = GR_T + 4;   // LDS row stride (16-byte rows)
constexpr int GR_CAP = 32;        // chunks per launch (bounds the workspace)

// partial[chunk][d][d] (fp32): 128 x 128 tile (ti,tj), ti <= tj, of sum over the chunk's rows of
// (x-c)(x-c)^T.  Four waves as 2 x 2, each a 64 x 64 quadrant (2 x 2 MFMA tiles, two operand reads
// per two MFMAs); rows are staged 16 at a time through double LDS buffers, fetched into registers
// one stage ahead with the shift already subtracted.  VEC: d is a multiple of 4 and x is 16-byte
// aligned, so a staged quad is one load.
template <bool VEC>
__global__ __launch_bounds__(256) void gram_tile_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                        long n, int d, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float ta[2][GR_KC * GR_LD], tb[2][GR_KC * GR_LD];
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (ti > tj) return;
    const long r0 = (long)blockIdx.z * GR_ROWS;
    const long r1 = min(n, r0 + GR_ROWS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l32 = lane & 31;
    const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.0f;
    // this thread stages quads (row q / 32, column 4 (q % 32)) for q = tid, tid + 256: same column both times
    const int scol = (tid & 31) * 4, srow = tid >> 5;
    const int ca = ti * GR_T + scol, cb = tj * GR_T + scol;
    float sa[4], sb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        sa[c] = ca + c < d ? shift[ca + c] : 0.0f;
        sb[c] = cb + c < d ? shift[cb + c] : 0.0f;
    }
    float4 pa[2], pb[2];
    auto quad = [&](long row, int col, const float (&sh)[4]) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < r1) {
            const float* p = x + row * d + col;
            if (VEC) {
                if (col < d) {
                    v = *reinterpret_cast<const float4*>(p);
                    v.x -= sh[0]; v.y -= sh[1]; v.z -= sh[2]; v.w -= sh[3];
                }
            } else {
                if (col < d) v.x = p[0] - sh[0];
                if (col + 1 < d) v.y = p[1] - sh[1];
                if (col + 2 < d) v.z = p[2] - sh[2];
                if (col + 3 < d) v.w = p[3] - sh[3];
            }
        }
        return v;
    };
    auto fetch = [&](long rr) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pa[i] = quad(rr + srow + 8 * i, ca, sa);
            pb[i] = quad(rr + srow + 8 * i, cb, sb);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<float4*>(&ta[buf][(srow + 8 * i) * GR_LD + scol]) = pa[i];
            *reinterpret_cast<float4*>(&tb[buf][(srow + 8 * i) * GR_LD + scol]) = pb[i];
        }
    };
    fetch(r0);
    commit(0);
    __syncthreads();
    int cur = 0;
    for (long rr = r0; rr < r1; rr += GR_KC, cur ^= 1) {
        const bool more = rr + GR_KC < r1;
        if (more) fetch(rr + GR_KC);
        const float* ap = ta[cur] + h * GR_LD + wi + l32;
        const float* bp = tb[cur] + h * GR_LD + wj + l32;
#pragma unroll
        for (int k = 0; k < GR_KC; k += 2) {
            const float a0 = ap[k * GR_LD], a1 = ap[k * GR_LD + 32];
            const float b0 = bp[k * GR_LD], b1 = bp[k * GR_LD + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) commit(cur ^ 1);
        __syncthreads();
    }
    float* dst = partial + (long)blockIdx.z * d * d;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = tj * GR_T + wj + 32 * b + l32;
            if (col < d) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int row = ti * GR_T + wi + 32 * a + (j & 3) + 8 * (j >> 2) + 4 * h;
                    if (row < d) dst[(long)row * d + col] = acc[a][b][j];
                }
            }
        }
}

// Small feature dimensions (the d = 5, 10, 30 of the scripts): the Gram matrix entirely in float64,
// products included.  Their covariances are often degenerate, and float32 products would leave
// eigenvalues of order -1e-7 that the square-root iteration cannot take.
constexpr int G64_MAXD = 128;
constexpr int G64_ROWS = 1024;    // rows per workgroup
constexpr int G64_KT = 32;        // rows staged per LDS tile
__global__ __launch_bounds__(256) void gram_f64_chunk_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                             long n, int d, double* __restrict__ partial) {
    __shared__ double tile[G64_KT * G64_MAXD];
    const long r0 = (long)blockIdx.x * G64_ROWS, r1 = min(n, r0 + G64_ROWS);
    const int tid = threadIdx.x, npairs = d * d;
    constexpr int PER = (G64_MAXD * G64_MAXD + 255) / 256;
    double acc[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) acc[u] = 0.0;
    for (long rr = r0; rr < r1; rr += G64_KT) {
        __syncthreads();
        for (int idx = tid; idx < G64_KT * d; idx += 256) {
            const int k = idx / d, c = idx - k * d;
            tile[k * d + c] = (rr + k < r1) ? (double)x[(rr + k) * d + c] - (double)shift[c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int pr = tid + u * 256;
            if (pr < npairs) {
                const int i = pr / d, j = pr - i * d;
                double s = acc[u];
                for (int k = 0; k < G64_KT; ++k) s += tile[k * d + i] * tile[k * d + j];
                acc[u] = s;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pr = tid + u * 256;
        if (pr < npairs) partial[(long)blockIdx.x * npairs + pr] = acc[u];
    }
}
__global__ __launch_bounds__(256) void gram_f64_reduce_kernel(const double* __restrict__ partial, int nchunks, int d,
                                                              double* __restrict__ gram) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= d * d) return;
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += partial[(long)c * d * d + idx];
    gram[idx] += s;
}

// gram[i][j] += sum_chunks partial (float64) for the elements of the 128 x 128 tiles on or above the
// diagonal; the tiles below it are not maintained (cov_finalize_kernel reads their mirror image)
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ partial, int nchunks, int d,
                                                          double* __restrict__ gram) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)d * d) return;
    const int i = (int)(idx / d), j = (int)(idx - (long)i * d);
    if (i / GR_T > j / GR_T) return;
    double s = 0.0;
    int c = 0;
    for (; c + 4 <= nchunks; c += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = partial[(long)(c + u) * d * d + idx];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (double)v[u];
    }
    for (; c < nchunks; ++c) s += (double)partial[(long)c * d * d + idx];
    gram[idx] += s;
}

// sum[c] += sum_rows (x[row][c] - shift[c]) in float64: per 1024-row chunk one partial per column
// (thread = column, coalesced over columns), then the chunk partials are added in a fixed order
constexpr int CS_ROWS = 1024;
__global__ __launch_bounds__(256) void shifted_colsum_chunk_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ shift, long n, int d,
                                                                   double* __restrict__ partial) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    const long r0 = (long)blockIdx.y * CS_ROWS, r1 = min(n, r0 + CS_ROWS);
    double s = 0.0;
    const float sh = shift[c];
    for (long r = r0; r < r1; ++r) s += (double)(x[r * d + c] - sh);
    partial[(long)blockIdx.y * d + c] = s;
}
__global__ __launch_bounds__(256) void shifted_colsum_reduce_kernel(const double* __restrict__ partial, int nchunks,
                                                                    int d, double* __restrict__ sum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    double s = 0.0;
    for (int k = 0; k < nchunks; ++k) s += partial[(long)k * d + c];
    sum[c] += s;
}

// mean = shift + sum/n ; cov = (gram - sum sum^T / n) / (n - 1)   (np.cov: unbiased)
__global__ __launch_bounds__(256) void cov_finalize_kernel(const double* __restrict__ gram,
                                                           const double* __restrict__ sum,
                                                           const float* __restrict__ shift, double n, int d,
                                                           double* __restrict__ mean, double* __restrict__ cov) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < d) mean[idx] = (double)shift[idx] + sum[idx] / n;
    if (idx < (long)d * d) {
        const int i = (int)(idx / d), j = (int)(idx % d);
        // the accumulator keeps the 128 x 128 tiles on or above the diagonal only
        const long src = (i / GR_T <= j / GR_T) ? idx : (long)j * d + i;
        cov[idx] = (gram[src] - sum[i] * sum[j] / n) / (n - 1.0);
    }
}

// t = 1.5 I - 0.5 zy
__global__ __launch_bounds__(256) void ns_residual_kernel(const double* __restrict__ zy, int d, double* __restrict__ t) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)d * d) return;
    const int i = (int)(idx / d), j = (int)(idx % d);
    t[idx] = (i == j ? 1.5 : 0.0) - 0.5 * zy[idx];
}

// out = 0.5 (a + a^T) * scale
__global__ __launch_bounds__(256) void symmetrize_kernel(const double* __restrict__ a, int d, double scale,
                                                         double* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)d * d) return;
    const int i = (int)(idx / d), j = (int)(idx % d);
    out[idx] = 0.5 * (a[idx] + a[(long)j * d + i]) * scale;
}

// out[0] += trace(a), out[1] += sum of squares (Frobenius norm^2) over this workgroup's share of the rows;
// the caller zeroes out[0..1]
__global__ __launch_bounds__(256) void trace_norm_kernel(const double* __restrict__ a, int d, double* __restrict__ out) {
    __shared__ double red[4];
    double tr = 0.0, sq = 0.0;
    for (int i = blockIdx.x; i < d; i += gridDim.x) {
        const double* row = a + (long)i * d;
        for (int j = threadIdx.x; j < d; j += 256) {
            const double v = row[j];
            sq += v * v;
            if (j == i) tr += v;
        }
    }
    tr = block_sum<256, double>(tr, red);
    sq = block_sum<256, double>(sq, red);
    if (threadIdx.x == 0) {
        atomicAdd(out, tr);
        atomicAdd(out + 1, sq);
    }
}

// out[0] = max over rows of sum |a_ij| (the infinity norm, an upper bound of the largest eigenvalue of a symmetric
// matrix); non-negative doubles order like their bit patterns, so the maximum is an integer atomic.  The caller
// zeroes out[0].
__global__ __launch_bounds__(256) void inf_norm_kernel(const double* __restrict__ a, int d, double* __restrict__ out) {
    __shared__ double red[4];
    double best = 0.0;
    for (int i = blockIdx.x; i < d; i += gridDim.x) {
        const double* row = a + (long)i * d;
        double s = 0.0;
        for (int j = threadIdx.x; j < d; j += 256) s += fabs(row[j]);
        s = block_sum<256, double>(s, red);
        best = s > best ? s : best;                    // NaN rows: s > best is false; the Frobenius norm catches them
    }
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(best));
}

// y = a x for a d x d float64 matrix: one wave per row (power iteration for the scale of the Newton-Schulz iteration)
__global__ __launch_bounds__(256) void matvec_f64_kernel(const double* __restrict__ a, const double* __restrict__ x, int d,
                                                         double* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= d) return;
    const double* r = a + (long)row * d;
    double s = 0.0;
    for (int j = lane; j < d; j += 64) s += r[j] * x[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[row] = s;
}

// t = 1.5 I - 0.25 (w + w^T), and gap[0] += sum of (I - w)^2 over this tile (the caller zeroes gap[0]).  One
// workgroup per 32 x 32 tile, the mirror tile through LDS.
__global__ __launch_bounds__(256) void ns_residual_gap_kernel(const double* __restrict__ w, int d, double* __restrict__ t,
                                                              double* __restrict__ gap) {
    __shared__ double tr[32][33];
    __shared__ double red[4];
    const int bi = blockIdx.y, bj = blockIdx.x;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + 8 * p, i = bj * 32 + r, j = bi * 32 + c;
        tr[r][c] = (i < d && j < d) ? w[(long)i * d + j] : 0.0;
    }
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + 8 * p, i = bi * 32 + r, j = bj * 32 + c;
        if (i < d && j < d) {
            const double x = w[(long)i * d + j], eye = i == j ? 1.0 : 0.0;
            t[(long)i * d + j] = 1.5 * eye - 0.25 * (x + tr[c][r]);
            acc += (eye - x) * (eye - x);
        }
    }
    acc = block_sum<256, double>(acc, red);
    if (threadIdx.x == 0) atomicAdd(gap, acc);
}

// out[0] = trace(a) of a [d][d] float64 matrix
__global__ __launch_bounds__(256) void trace_f64_kernel(const double* __restrict__ a, int d, double* __restrict__ out) {
    __shared__ double red[4];
    double t = 0.0;
    for (int i = threadIdx.x; i < d; i += 256) t += a[(long)i * d + i];
    t = block_sum<256, double>(t, red);
    if (threadIdx.x == 0) out[0] = t;
}

// The end-of-iteration rule of the float64 iteration, kept on the device so that the host can queue several iterations
// and look once.  state: [0] previous trace, [1] previous change, [2] / [3] whether those exist, [4] 0 running /
// 1 settled / 2 diverged, [5] iterations counted, [6] last trace.  One iteration = one call; calls after the end do nothing.
// gap (may be null): ||I - Z Y||_F^2 of the step just taken; at or below 1e-12 the step has landed at rounding level.
__global__ __launch_bounds__(256) void ns_state_step_kernel(const double* __restrict__ y, int d, const double* __restrict__ gap,
                                                            double* __restrict__ state) {
    __shared__ double red[4];
    if (state[4] != 0.0) return;
    double t = 0.0;
    for (int i = threadIdx.x; i < d; i += 256) t += y[(long)i * d + i];
    t = block_sum<256, double>(t, red);
    if (threadIdx.x != 0) return;
    state[5] += 1.0;
    state[6] = t;
    const double gsq = gap ? gap[0] : 1.0;
    const bool finite = fabs(t) <= 1.7976931348623157e308 && fabs(gsq) <= 1.7976931348623157e308;     // false for NaN too
    if (!finite) {
        state[4] = 2.0;
        return;
    }
    if (gap && gsq <= 1e-12) {
        state[4] = 1.0;
        return;
    }
    if (state[2] != 0.0) {
        const double delta = fabs(t - state[0]);
        // the trace rises monotonically while directions still converge: settled at rounding level, or -- once below
        // 1e-9 relative -- no longer shrinking (a singular matrix: from there on Z only amplifies rounding noise)
        if (delta <= 1e-15 * fabs(t) || (state[3] != 0.0 && delta >= state[1] && delta < 1e-9 * fabs(t))) {
            state[4] = 1.0;
            return;
        }
        state[1] = delta;
        state[3] = 1.0;
    }
    state[0] = t;
    state[2] = 1.0;
}

// ---- fp32 steps of the coupled Newton-Schulz iteration on padded matrices [d][ld] (ld a multiple of 32;
// columns d .. ld-1 are kept zero: they are the k padding of the split-bf16 products that form Z Y, Y T, T Z) ----
// kind 0: out = 1.5 I - 0.5 a;  kind 1: out = scale (a + a^T) / 2.  One workgroup per 32 x 32 tile; the
// transposed tile comes through LDS so both reads move whole cache lines.
template <int KIND>
__global__ __launch_bounds__(256) void ns_tile_kernel(const float* __restrict__ a, int d, long ld, float scale,
                                                      float* __restrict__ out) {
    __shared__ float tr[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    if (KIND == 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = r0 + 8 * p, i = bj * 32 + r, j = bi * 32 + c;      // the mirror tile (bj, bi)
            tr[r][c] = (i < d && j < d) ? a[(long)i * ld + j] : 0.0f;
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + 8 * p, i = bi * 32 + r, j = bj * 32 + c;
        if (i >= d) continue;
        float v = 0.0f;
        if (j < d) {
            const float x = a[(long)i * ld + j];
            v = KIND == 0 ? (i == j ? 1.5f : 0.0f) - 0.5f * x : 0.5f * scale * (x + tr[c][r]);
        }
        out[(long)i * ld + j] = v;
    }
}
__global__ __launch_bounds__(256) void trace_f32_kernel(const float* __restrict__ a, int d, long ld, double* __restrict__ out) {
    __shared__ double red[4];
    double t = 0.0;
    for (int i = threadIdx.x; i < d; i += 256) t += (double)a[(long)i * ld + i];
    t = block_sum<256, double>(t, red);
    if (threadIdx.x == 0) out[0] = t;
}
// out [d][ld] fp32 = a [d][d] float64 (padding columns zero), and back
__global__ __launch_bounds__(256) void narrow_pad_kernel(const double* __restrict__ a, int d, long ld, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)d * ld) return;
    const int i = (int)(idx / ld), j = (int)(idx % ld);
    out[idx] = j < d ? (float)a[(long)i * d + j] : 0.0f;
}
__global__ __launch_bounds__(256) void widen_unpad_kernel(const float* __restrict__ a, int d, long ld, double* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)d * d) return;
    const int i = (int)(idx / d), j = (int)(idx % d);
    out[idx] = (double)a[(long)i * ld + j];
}


// ---- float64 products of the Newton-Schulz iteration on the matrix cores -----------------------
// C = A B for d x d matrices of which B is symmetric: C[i][j] = sum_k A[i][k] B[j][k], both operands read along their
// rows.  v_mfma_f64_16x16x4_f64; 128 x 128 tile per workgroup (4 waves of 64 x 64), k steps of 16 staged through two LDS
// buffers (row pitch 17 doubles: the 32 lanes a ds_read_b64 serves together fall on 64 different banks), the next
// step's global loads in flight under the current step's 64 MFMAs per wave.
// UPPER: the product itself is symmetric (all three matrices are polynomials of one matrix): only tiles on or above
// the diagonal are computed, and every value is written to both (i, j) and (j, i) -- half the work, and a result that
// is symmetric to the bit.  EPI 1: C = 1.5 I - 0.5 A B (the iteration's residual step).  CT (general form): the
// transpose of the result is written there as well.
typedef double f64x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));
constexpr int DG_T = 128, DG_K = 16, DG_LD = 17;

template <bool UPPER, int EPI>
__global__ __launch_bounds__(512) void dgemm_nt_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                       double* __restrict__ C, double* __restrict__ CT, int d, int nt,
                                                       const double* __restrict__ skip) {
    __shared__ double sm[2][2][DG_T * DG_LD];          // [buffer][A, B][row][k]
    if (skip && skip[0] != 0.0) return;                // the iteration has ended on the device: the queued rest is void
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;           // 2 x 4 waves of 64 x 32: two waves per SIMD, which a lone
                                                       // workgroup needs to keep the matrix pipe busy
    const int l16 = lane & 15, lq = lane >> 4;
    int ti, tj;
    if (UPPER) {
        int id = blockIdx.x;
        ti = 0;
        while (id >= nt - ti) { id -= nt - ti; ++ti; }
        tj = ti + id;
    } else {
        ti = blockIdx.x / nt;
        tj = blockIdx.x - ti * nt;
    }
    // this thread's piece of a k step: 4 doubles of row (t >> 2) of both tiles (zeros beyond the matrix)
    const int lrow = t >> 2, lk = (t & 3) * 4;
    const long arow = (long)ti * DG_T + lrow, brow = (long)tj * DG_T + lrow;
    const bool aok = arow < d, bok = brow < d;
    const double* ap = A + (aok ? arow : 0) * d + lk;
    const double* bp = B + (bok ? brow : 0) * d + lk;
    f64x2_t ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = *reinterpret_cast<const f64x2_t*>(ap + k0 + 2 * i);
            rb[i] = *reinterpret_cast<const f64x2_t*>(bp + k0 + 2 * i);
        }
    };
    auto lstore = [&](int buf) {
        double* as = &sm[buf][0][lrow * DG_LD + lk];
        double* bs = &sm[buf][1][lrow * DG_LD + lk];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            as[2 * i] = aok ? ra[i][0] : 0.0;
            as[2 * i + 1] = aok ? ra[i][1] : 0.0;
            bs[2 * i] = bok ? rb[i][0] : 0.0;
            bs[2 * i + 1] = bok ? rb[i][1] : 0.0;
        }
    };
    f64x4_t acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const int nk = d / DG_K;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) gload((ks + 1) * DG_K);
        const double* as = &sm[buf][0][(wm * 64 + l16) * DG_LD + lq];
        const double* bs = &sm[buf][1][(wn * 32 + l16) * DG_LD + lq];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double a[4], b[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = as[m * 16 * DG_LD + kk * 4];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = bs[n * 16 * DG_LD + kk * 4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (ks + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    // register r of lane l of a 16 x 16 result holds row 4 r + l / 16, column l % 16
    const bool diag = UPPER && ti == tj;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = (long)ti * DG_T + wm * 64 + m * 16 + 4 * r + lq;
                const long col = (long)tj * DG_T + wn * 32 + n * 16 + l16;
                double v = acc[m][n][r];
                if (EPI == 1) v = (row == col ? 1.5 : 0.0) - 0.5 * v;
                if (row < d && col < d && (!diag || row <= col)) {
                    C[row * d + col] = v;
                    if (diag && row != col) C[col * d + row] = v;
                }
                acc[m][n][r] = v;
            }
    double* mirror = UPPER ? (diag ? nullptr : C) : CT;
    if (mirror) {
        // the mirror image: 16 x 32 pieces go through LDS so that it, too, leaves in 128-byte row pieces
        double* tr = &sm[0][0][0] + wave * (32 * DG_LD);                 // [32 columns][16 rows (+1)]
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) tr[(n * 16 + l16) * DG_LD + 4 * r + lq] = acc[m][n][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const long row0 = (long)ti * DG_T + wm * 64 + m * 16;       // rows of the tile = columns of the mirror image
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = i * 4 + lq;                               // column of the tile = row of the mirror image
                const long col = (long)tj * DG_T + wn * 32 + c;
                if (col < d && row0 + l16 < d) mirror[col * d + row0 + l16] = tr[c * DG_LD + l16];
            }
        }
    }
}

}  // namespace mdnvae
using namespace mdnvae;

static unsigned grid_for(long n) { return (unsigned)((n + 255) / 256); }

// column sums of x - shift over n rows, 64 chunks of 1024 rows per launch; `part` holds 64*d doubles
static int launch_shifted_colsum(const float* x, const float* shift, long n, int d, double* part, double* sum,
                                 hipStream_t st) {
    const long rows_per_launch = 64L * CS_ROWS;
    for (long r0 = 0; r0 < n; r0 += rows_per_launch) {
        const long rows = (n - r0 < rows_per_launch) ? n - r0 : rows_per_launch;
        const int nch = (int)((rows + CS_ROWS - 1) / CS_ROWS);
        hipLaunchKernelGGL(shifted_colsum_chunk_kernel, dim3(grid_for(d), nch), dim3(256), 0, st, x + r0 * d, shift, rows,
                           d, part);
        MV_LAUNCH_CHECK("shifted_colsum_chunk_kernel");
        hipLaunchKernelGGL(shifted_colsum_reduce_kernel, dim3(grid_for(d)), dim3(256), 0, st, part, nch, d, sum);
        MV_LAUNCH_CHECK("shifted_colsum_reduce_kernel");
    }
    return 0;
}

extern "C" {

size_t mdnvae_fid_gram_workspace_floats(long n, int d) {
    const size_t colsum = (size_t)2 * 64 * d;        // 64 chunk partials per column, float64
    if (d <= G64_MAXD) {                             // float64 partials, two floats each
        const long nch = (n + G64_ROWS - 1) / G64_ROWS;
        const size_t gram = (size_t)2 * (nch < 64 ? nch : 64) * d * d;
        return gram > colsum ? gram : colsum;
    }
    const long nch = (n + GR_ROWS - 1) / GR_ROWS;
    const long cap = nch < GR_CAP ? nch : GR_CAP;    // chunks processed per launch
    const size_t gram = (size_t)cap * d * d;
    return gram > colsum ? gram : colsum;
}

int mdnvae_fid_accumulate(const float* x, const float* shift, long n, int d, float* workspace, double* sum,
                          double* gram, void* stream) {
    MV_REQUIRE(n >= 0 && d >= 1, E_ARG, "mdnvae_fid_accumulate: bad size");
    MV_REQUIRE(d <= 8192, E_SHAPE, "mdnvae_fid_accumulate: feature dimension %d exceeds 8192", d);
    if (n == 0) return 0;
    MV_REQUIRE(x && shift && workspace && sum && gram, E_ARG, "mdnvae_fid_accumulate: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (d <= G64_MAXD) {
        MV_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, E_ALIGN,
                   "mdnvae_fid_accumulate: workspace must be 8-byte aligned");
        double* part = reinterpret_cast<double*>(workspace);
        const long rows_per_launch = 64L * G64_ROWS;
        for (long r0 = 0; r0 < n; r0 += rows_per_launch) {
            const long rows = (n - r0 < rows_per_launch) ? n - r0 : rows_per_launch;
            const int nch = (int)((rows + G64_ROWS - 1) / G64_ROWS);
            hipLaunchKernelGGL(gram_f64_chunk_kernel, dim3(nch), dim3(256), 0, st, x + r0 * d, shift, rows, d, part);
            MV_LAUNCH_CHECK("gram_f64_chunk_kernel");
            hipLaunchKernelGGL(gram_f64_reduce_kernel, dim3(grid_for((long)d * d)), dim3(256), 0, st, part, nch, d, gram);
            MV_LAUNCH_CHECK("gram_f64_reduce_kernel");
        }
        return launch_shifted_colsum(x, shift, n, d, part, sum, st);
    }
    const int tiles = (d + GR_T - 1) / GR_T;
    const long rows_per_launch = (long)GR_CAP * GR_ROWS;
    const bool vec = (d & 3) == 0 && aligned16(x);
    for (long r0 = 0; r0 < n; r0 += rows_per_launch) {
        const long rows = (n - r0 < rows_per_launch) ? n - r0 : rows_per_launch;
        const int nch = (int)((rows + GR_ROWS - 1) / GR_ROWS);
        if (vec)
            hipLaunchKernelGGL(gram_tile_kernel<true>, dim3(tiles, tiles, nch), dim3(256), 0, st, x + r0 * d, shift, rows,
                               d, workspace);
        else
            hipLaunchKernelGGL(gram_tile_kernel<false>, dim3(tiles, tiles, nch), dim3(256), 0, st, x + r0 * d, shift, rows,
                               d, workspace);
        MV_LAUNCH_CHECK("gram_tile_kernel");
        hipLaunchKernelGGL(gram_reduce_kernel, dim3(grid_for((long)d * d)), dim3(256), 0, st, workspace, nch, d, gram);
        MV_LAUNCH_CHECK("gram_reduce_kernel");
    }
    MV_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, E_ALIGN,
               "mdnvae_fid_accumulate: workspace must be 8-byte aligned");
    return launch_shifted_colsum(x, shift, n, d, reinterpret_cast<double*>(workspace), sum, st);
}

int mdnvae_fid_finalize(const double* gram, const double* sum, const float* shift, double n, int d, double* mean,
                        double* cov, void* stream) {
    MV_REQUIRE(d >= 1 && n >= 2.0, E_ARG, "mdnvae_fid_finalize: need at least two samples");
    MV_REQUIRE(gram && sum && shift && mean && cov, E_ARG, "mdnvae_fid_finalize: null pointer");
    hipLaunchKernelGGL(cov_finalize_kernel, dim3(grid_for((long)d * d)), dim3(256), 0, (hipStream_t)stream, gram, sum,
                       shift, n, d, mean, cov);
    MV_LAUNCH_CHECK("cov_finalize_kernel");
    return 0;
}

int mdnvae_ns_residual(const double* zy, int d, double* t, void* stream) {
    MV_REQUIRE(d >= 1 && zy && t, E_ARG, "mdnvae_ns_residual: bad argument");
    hipLaunchKernelGGL(ns_residual_kernel, dim3(grid_for((long)d * d)), dim3(256), 0, (hipStream_t)stream, zy, d, t);
    MV_LAUNCH_CHECK("ns_residual_kernel");
    return 0;
}

int mdnvae_symmetrize(const double* a, int d, double scale, double* out, void* stream) {
    MV_REQUIRE(d >= 1 && a && out && a != out, E_ARG, "mdnvae_symmetrize: bad argument (in-place is not supported)");
    hipLaunchKernelGGL(symmetrize_kernel, dim3(grid_for((long)d * d)), dim3(256), 0, (hipStream_t)stream, a, d, scale,
                       out);
    MV_LAUNCH_CHECK("symmetrize_kernel");
    return 0;
}

int mdnvae_trace_norm(const double* a, int d, double* out2, void* stream) {
    MV_REQUIRE(d >= 1 && a && out2, E_ARG, "mdnvae_trace_norm: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const hipError_t e = clear_words(out2, 4, st);
    MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_trace_norm: clearing the result failed: %s", hipGetErrorString(e));
    // one workgroup up to 256 rows: small matrices keep a fixed summation order
    hipLaunchKernelGGL(trace_norm_kernel, dim3(d <= 256 ? 1u : 512u), dim3(256), 0, st, a, d, out2);
    MV_LAUNCH_CHECK("trace_norm_kernel");
    return 0;
}

int mdnvae_inf_norm(const double* a, int d, double* out, void* stream) {
    MV_REQUIRE(d >= 1 && a && out, E_ARG, "mdnvae_inf_norm: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const hipError_t e = clear_words(out, 2, st);
    MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_inf_norm: clearing the result failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(inf_norm_kernel, dim3(d < 512 ? (unsigned)d : 512u), dim3(256), 0, st, a, d, out);
    MV_LAUNCH_CHECK("inf_norm_kernel");
    return 0;
}

int mdnvae_matvec_f64(const double* a, const double* x, int d, double* y, void* stream) {
    MV_REQUIRE(d >= 1 && a && x && y && x != y, E_ARG, "mdnvae_matvec_f64: bad argument");
    hipLaunchKernelGGL(matvec_f64_kernel, dim3((unsigned)((d + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, x, d, y);
    MV_LAUNCH_CHECK("matvec_f64_kernel");
    return 0;
}

int mdnvae_ns_residual_gap(const double* w, int d, double* t, double* gap, void* stream) {
    MV_REQUIRE(d >= 1 && w && t && gap && w != t, E_ARG, "mdnvae_ns_residual_gap: bad argument (in-place is not supported)");
    hipStream_t st = (hipStream_t)stream;
    const hipError_t e = clear_words(gap, 2, st);
    MV_REQUIRE(e == hipSuccess, (int)e, "mdnvae_ns_residual_gap: clearing gap failed: %s", hipGetErrorString(e));
    const unsigned tiles = (unsigned)((d + 31) / 32);
    hipLaunchKernelGGL(ns_residual_gap_kernel, dim3(tiles, tiles), dim3(256), 0, st, w, d, t, gap);
    MV_LAUNCH_CHECK("ns_residual_gap_kernel");
    return 0;
}

int mdnvae_dgemm_sym(const double* a, const double* b, int d, int upper, int epilogue, double* c, double* c_transposed,
                     const double* skip, void* stream) {
    MV_REQUIRE(d >= 0 && d % 16 == 0, E_SHAPE, "mdnvae_dgemm_sym: d = %d is not a multiple of 16", d);
    MV_REQUIRE(epilogue == 0 || epilogue == 1, E_ARG, "mdnvae_dgemm_sym: epilogue %d", epilogue);
    if (d == 0) return 0;
    MV_REQUIRE(a && b && c, E_ARG, "mdnvae_dgemm_sym: null pointer");
    MV_REQUIRE(c != a && c != b && c_transposed != a && c_transposed != b && c_transposed != c, E_ARG,
               "mdnvae_dgemm_sym: the results may not overwrite an operand or each other");
    MV_REQUIRE(!upper || !c_transposed, E_ARG, "mdnvae_dgemm_sym: a symmetric result is its own transpose");
    const int nt = (d + DG_T - 1) / DG_T;
    hipStream_t st = (hipStream_t)stream;
    if (upper) {
        dim3 grid((unsigned)(nt * (nt + 1) / 2));
        if (epilogue) hipLaunchKernelGGL((dgemm_nt_kernel<true, 1>), grid, dim3(512), 0, st, a, b, c, c_transposed, d, nt, skip);
        else hipLaunchKernelGGL((dgemm_nt_kernel<true, 0>), grid, dim3(512), 0, st, a, b, c, c_transposed, d, nt, skip);
    } else {
        dim3 grid((unsigned)(nt * nt));
        if (epilogue) hipLaunchKernelGGL((dgemm_nt_kernel<false, 1>), grid, dim3(512), 0, st, a, b, c, c_transposed, d, nt, skip);
        else hipLaunchKernelGGL((dgemm_nt_kernel<false, 0>), grid, dim3(512), 0, st, a, b, c, c_transposed, d, nt, skip);
    }
    MV_LAUNCH_CHECK("dgemm_nt_kernel");
    return 0;
}

int mdnvae_trace(const double* a, int d, double* out, void* stream) {
    MV_REQUIRE(d >= 1 && a && out, E_ARG, "mdnvae_trace: bad argument");
    hipLaunchKernelGGL(trace_f64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, d, out);
    MV_LAUNCH_CHECK("trace_f64_kernel");
    return 0;
}

int mdnvae_ns_state_step(const double* y, int d, const double* gap, double* state, void* stream) {
    MV_REQUIRE(d >= 1 && y && state, E_ARG, "mdnvae_ns_state_step: bad argument");
    hipLaunchKernelGGL(ns_state_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, y, d, gap, state);
    MV_LAUNCH_CHECK("ns_state_step_kernel");
    return 0;
}

static int ns_f32_args(const char* fn, const void* a, int d, long ld, const void* out) {
    MV_REQUIRE(d >= 1 && ld >= d && ld % 32 == 0, E_ARG, "%s: d = %d, ld = %ld (ld a multiple of 32, at least d)", fn, d, ld);
    MV_REQUIRE(a && out && a != out, E_ARG, "%s: null pointer, or in-place", fn);
    return 0;
}

int mdnvae_ns_residual_f32(const float* zy, int d, long ld, float* t, void* stream) {
    if (int rc = ns_f32_args("mdnvae_ns_residual_f32", zy, d, ld, t)) return rc;
    hipLaunchKernelGGL(ns_tile_kernel<0>, dim3((unsigned)(ld / 32), (unsigned)((d + 31) / 32)), dim3(256), 0,
                       (hipStream_t)stream, zy, d, ld, 1.0f, t);
    MV_LAUNCH_CHECK("ns_tile_kernel");
    return 0;
}

int mdnvae_symmetrize_f32(const float* a, int d, long ld, float scale, float* out, void* stream) {
    if (int rc = ns_f32_args("mdnvae_symmetrize_f32", a, d, ld, out)) return rc;
    hipLaunchKernelGGL(ns_tile_kernel<1>, dim3((unsigned)(ld / 32), (unsigned)((d + 31) / 32)), dim3(256), 0,
                       (hipStream_t)stream, a, d, ld, scale, out);
    MV_LAUNCH_CHECK("ns_tile_kernel");
    return 0;
}

int mdnvae_trace_f32(const float* a, int d, long ld, double* out, void* stream) {
    if (int rc = ns_f32_args("mdnvae_trace_f32", a, d, ld, out)) return rc;
    hipLaunchKernelGGL(trace_f32_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, d, ld, out);
    MV_LAUNCH_CHECK("trace_f32_kernel");
    return 0;
}

int mdnvae_f64_to_f32_padded(const double* a, int d, long ld, float* out, void* stream) {
    if (int rc = ns_f32_args("mdnvae_f64_to_f32_padded", a, d, ld, out)) return rc;
    hipLaunchKernelGGL(narrow_pad_kernel, dim3(grid_for((long)d * ld)), dim3(256), 0, (hipStream_t)stream, a, d, ld, out);
    MV_LAUNCH_CHECK("narrow_pad_kernel");
    return 0;
}

int mdnvae_f32_padded_to_f64(const float* a, int d, long ld, double* out, void* stream) {
    if (int rc = ns_f32_args("mdnvae_f32_padded_to_f64", a, d, ld, out)) return rc;
    hipLaunchKernelGGL(widen_unpad_kernel, dim3(grid_for((long)d * d)), dim3(256), 0, (hipStream_t)stream, a, d, ld, out);
    MV_LAUNCH_CHECK("widen_unpad_kernel");
    return 0;
}

}  // extern "C"
