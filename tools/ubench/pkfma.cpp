// Issue rate of packed fp32 FMA on one SIMD: all-VGPR operands, a scalar-broadcast operand, and the
// unpacked form.  hipcc --offload-arch=gfx950 -O3 -o pkfma pkfma.cpp && ./pkfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int CHAINS = 12, ITER = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    v2 acc[CHAINS];
    const float s0 = in[0], s1 = in[1];            // wave-uniform (SGPR after the scalar load)
    const v2 m{in[threadIdx.x & 7], in[(threadIdx.x + 1) & 7]};
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) acc[j] = v2{(float)threadIdx.x + j, (float)j};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < CHAINS; ++j) {
            if (MODE == 0) acc[j] = acc[j] * m + m;                       // v_pk_fma, VGPR operands
            if (MODE == 1) acc[j] = acc[j] * v2{s0, s0} + v2{s1, s1};     // scalar broadcast operands
            if (MODE == 2) { acc[j].x = __builtin_fmaf(acc[j].x, m.x, m.y); acc[j].y = __builtin_fmaf(acc[j].y, m.y, m.x); }
            if (MODE == 3) acc[j] = acc[j] * v2{s0, s0} + m;              // one scalar broadcast operand
        }
    }
    v2 t{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < CHAINS; ++j) t += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;
}

template <int MODE>
void run(const char* name, float* out, float* in, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;       // 4 waves per block -> one per SIMD per block
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, ITER);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, ITER);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)waves_per_simd * ITER * CHAINS * (MODE == 2 ? 2 : 1);
    printf("%-28s waves/SIMD %d: %.3f ms, %.2f cycles per instruction per SIMD at 2.4 GHz, %.1f TFLOP/s\n", name,
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd,
           (double)blocks * 256 * ITER * CHAINS * 4 / (ms * 1e-3) / 1e12);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&in, 64);
    float h[16];
    for (int i = 0; i < 16; ++i) h[i] = 0.5f + 0.01f * i;
    hipMemcpy(in, h, 64, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        run<0>("pk_fma vgpr", out, in, w);
        run<1>("pk_fma 2 scalar operands", out, in, w);
        run<3>("pk_fma 1 scalar operand", out, in, w);
        run<2>("2 x v_fma", out, in, w);
    }
    return 0;
}
