// Prints where v_mfma_f64_16x16x4_f64 puts its result: D = A B with A[i][k] = i + 1 (k = 0 only), B[0][j] = 100 (j + 1).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    const double a = (l / 16 == 0) ? (double)(l % 16 + 1) : 0.0;      // A[i = l % 16][k = l / 16]
    const double b = (l / 16 == 0) ? 100.0 * (l % 16 + 1) : 0.0;      // B[k = l / 16][j = l % 16]
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    double* d;
    hipMalloc(&d, 256 * 8);
    probe<<<1, 64>>>(d);
    double h[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) printf("  i=%2d j=%2d", (int)(h[l * 4 + r]) % 100 - 1, (int)(h[l * 4 + r] / 100 + 0.5) / 1 - 0 > 0 ? (int)((h[l * 4 + r] / ((int)(h[l*4+r]) % 100 == 0 ? 1 : 1))) : -1);
        printf("   raw %g %g %g %g\n", h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
