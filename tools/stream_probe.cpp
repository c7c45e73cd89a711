// Read-only streaming rates of one [B][7569] fp32 array under the access shapes a per-sample sweep can take
// (every variant reads each sample's 30 KiB run of whole 1 KiB pieces, 128-byte aligned, once):
//   flat    : waves walk consecutive pieces of the whole array (grid-stride), registers
//   wave    : one wave per sample, samples strided by the number of waves, registers
//   group   : one workgroup (WAVES waves) per sample, wave w takes pieces p = w (mod WAVES), registers
//   block   : one wave per sample, a workgroup takes WAVES consecutive samples, registers
// AHEAD pieces (16 bytes per lane each) are in flight per wave.   usage: stream_probe [B] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int D = 7569, PIECES = 30;

__device__ __forceinline__ const char* run_of(const float* x, long b) {
    return reinterpret_cast<const char*>(reinterpret_cast<unsigned long>(x + b * D) & ~127ul);
}

template <int AHEAD>
__device__ __forceinline__ float sweep(const char* base, int first, int step, int lane) {
    // pieces first, first + step, ... < PIECES of one run
    v4 r[AHEAD];
    float acc = 0.0f;
    int p = first, q = first;
#pragma unroll
    for (int i = 0; i < AHEAD; ++i, q += step)
        if (q < PIECES) r[i] = __builtin_nontemporal_load(reinterpret_cast<const v4*>(base + q * 1024 + lane * 16));
    while (p < PIECES) {
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) {
            if (p < PIECES) {
                acc += r[i].x + r[i].y + r[i].z + r[i].w;
                if (q < PIECES) r[i] = __builtin_nontemporal_load(reinterpret_cast<const v4*>(base + q * 1024 + lane * 16));
                p += step;
                q += step;
            }
        }
    }
    return acc;
}

template <int MODE, int AHEAD, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(const float* __restrict__ x, int B, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc = 0.0f;
    if (MODE == 0) {
        const long total = ((long)B * D * 4 + 1023) / 1024;
        const long nw = (long)gridDim.x * WAVES, w = (long)blockIdx.x * WAVES + wid;
        v4 r[AHEAD];
        long q = w;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i, q += nw)
            if (q < total) r[i] = __builtin_nontemporal_load(reinterpret_cast<const v4*>(reinterpret_cast<const char*>(x) + q * 1024 + lane * 16));
        for (long p = w; p < total;) {
#pragma unroll
            for (int i = 0; i < AHEAD; ++i) {
                if (p < total) {
                    acc += r[i].x + r[i].y + r[i].z + r[i].w;
                    if (q < total) r[i] = __builtin_nontemporal_load(reinterpret_cast<const v4*>(reinterpret_cast<const char*>(x) + q * 1024 + lane * 16));
                    p += nw;
                    q += nw;
                }
            }
        }
    } else if (MODE == 1) {
        const int nw = gridDim.x * WAVES;
        for (int b = blockIdx.x * WAVES + wid; b < B; b += nw) acc += sweep<AHEAD>(run_of(x, b), 0, 1, lane);
    } else if (MODE == 2) {
        for (int b = blockIdx.x; b < B; b += gridDim.x) acc += sweep<AHEAD>(run_of(x, b), wid, WAVES, lane);
    } else {
        for (int b0 = blockIdx.x * WAVES; b0 < B; b0 += gridDim.x * WAVES)
            if (b0 + wid < B) acc += sweep<AHEAD>(run_of(x, b0 + wid), 0, 1, lane);
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int AHEAD, int WAVES>
static void run(const char* name, const float* x, int B, float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ts;
    for (int it = 0; it < iters + 2; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<MODE, AHEAD, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, x, B, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)B * D * 4;
    printf("%-6s ahead %2d waves/wg %d blocks %5d : %.4f ms  %.0f GB/s\n", name, AHEAD, WAVES, blocks, ts[ts.size() / 2],
           bytes / ts[ts.size() / 2] / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16384, iters = argc > 2 ? atoi(argv[2]) : 10;
    float *x, *out;
    hipMalloc(&x, (size_t)B * D * 4 + 4096);
    hipMalloc(&out, 16);
    hipMemset(x, 0, (size_t)B * D * 4 + 4096);
    x += 1;
    for (int blocks : {512, 1024, 2048, 4096}) {
        run<0, 4, 4>("flat", x, B, out, blocks, iters);
        run<0, 8, 4>("flat", x, B, out, blocks, iters);
        run<1, 4, 4>("wave", x, B, out, blocks, iters);
        run<1, 8, 4>("wave", x, B, out, blocks, iters);
        run<1, 16, 4>("wave", x, B, out, blocks, iters);
        run<2, 2, 4>("group", x, B, out, blocks, iters);
        run<2, 4, 4>("group", x, B, out, blocks, iters);
        run<2, 8, 4>("group", x, B, out, blocks, iters);
        run<3, 4, 4>("block", x, B, out, blocks, iters);
        run<3, 8, 4>("block", x, B, out, blocks, iters);
    }
    run<2, 4, 8>("group", x, B, out, 1024, iters);
    run<2, 4, 8>("group", x, B, out, 2048, iters);
    run<2, 2, 8>("group", x, B, out, 2048, iters);
    run<1, 8, 8>("wave", x, B, out, 512, iters);
    run<1, 8, 8>("wave", x, B, out, 1024, iters);
    return 0;
}
