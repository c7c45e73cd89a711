// fp32 MFMAs whose A operands come from LDS: one dword per instruction (as 24 scalar reads per 24 MFMAs) against one
// 16-byte read per four instructions.  Prints ns per MFMA per SIMD at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int WIDE>
__global__ void rate(float* out, int iters) {
    __shared__ float pan[96 * 100];
    for (int i = threadIdx.x; i < 96 * 100; i += blockDim.x) pan[i] = i * 1e-4f;
    __syncthreads();
    const int lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4;
    float x[4] = {1.f + lane, 2.f, 3.f, 4.f};
    f4 c[6];
    for (int i = 0; i < 6; ++i) c[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kq = 0; kq < 6; ++kq) {
            float w[24];
            if (WIDE) {
#pragma unroll
                for (int ct = 0; ct < 6; ++ct) {
                    const f4 v = *reinterpret_cast<const f4*>(pan + (16 * ct + c16) * 100 + 16 * kq + 4 * g);
                    w[ct] = v[0]; w[6 + ct] = v[1]; w[12 + ct] = v[2]; w[18 + ct] = v[3];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ct = 0; ct < 6; ++ct) w[r * 6 + ct] = pan[(16 * kq + 4 * g + r) * 100 + ct * 16 + c16];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < 6; ++ct) c[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r * 6 + ct], x[r], c[ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 6; ++i) s += c[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WIDE>
void run(const char* name) {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        rate<WIDE><<<256, 256 * wps>>>(d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        rate<WIDE><<<256, 256 * wps>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 144 * wps;
        printf("%-34s %d wave(s)/SIMD: %6.1f ns per MFMA per SIMD\n", name, wps, ms * 1e6 / n);
    }
    hipFree(d);
}
int main() {
    run<0>("A operand: one dword per MFMA");
    run<1>("A operand: 16 bytes per four MFMAs");
    return 0;
}
