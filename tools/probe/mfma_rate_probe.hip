// Issue rate of matrix instructions: every wave of a full grid runs a loop of independent MFMAs (8 accumulators);
// prints cycles per instruction per SIMD at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void rate(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f4 c[8];
    f16v c32[4];
    d4 cd[8];
    for (int i = 0; i < 8; ++i) { c[i] = f4{0, 0, 0, 0}; cd[i] = d4{0, 0, 0, 0}; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c32[i][j] = 0.f;
    h8 ha, hb;
    for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)a; hb[j] = (_Float16)b; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
            if (KIND == 1) c32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c32[i & 3], 0, 0, 0);
            if (KIND == 2) cd[i] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a, (double)b, cd[i], 0, 0, 0);
            if (KIND == 3) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[i], 0, 0, 0);
            if (KIND == 4) c32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c32[i & 3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + (float)cd[i][0];
    for (int i = 0; i < 4; ++i) s += c32[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, double flop) {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4 * 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        rate<KIND><<<256, 256 * wps>>>(d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        rate<KIND><<<256, 256 * wps>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 8 * wps;            // instructions per SIMD
        printf("%-28s %d wave(s)/SIMD: %6.1f ns per instruction per SIMD, %7.1f TFLOP/s\n", name, wps, ms * 1e6 / n,
               flop * n * 1024 / (ms * 1e-3) / 1e12);
    }
    hipFree(d);
}
int main() {
    run<0>("v_mfma_f32_16x16x4_f32", 2048);
    run<1>("v_mfma_f32_32x32x2_f32", 4096);
    run<2>("v_mfma_f64_16x16x4_f64", 2048);
    run<3>("v_mfma_f32_16x16x32_f16", 16384);
    run<4>("v_mfma_f32_32x32x16_f16", 32768);
    return 0;
}
