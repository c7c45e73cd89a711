// Agreement and timing harness for the split-bf16 GEMM of the first layers.
//   sg_tune [B] : forward  Y[B,2048] = relu(X[B,3840] . W[2048,3840]^T + b)
//                 gradient G[3840,2048] = X^T . P   (P = grad_pre [B,2048])
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../include/mdnvae_hip.h"

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define RC(e) do { int rc_ = (e); if (rc_) { printf("call failed %d at line %d: %s\n", rc_, __LINE__, mdnvae_last_error()); return 1; } } while (0)

template <typename F>
static float time_ms(F&& f, int iters = 10) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int it = 0; it < iters + 2; ++it) {
        hipEventRecord(e0, 0);
        f();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    const long B = argc > 1 ? atol(argv[1]) : 16384, T = 3840, H = 2048;
    const long Bp = (B + 255) / 256 * 256;
    std::mt19937 rng(3);
    std::normal_distribution<float> G(0.0f, 1.0f);
    std::vector<float> hx(B * T), hw(H * T), hb(H), hp(B * H);
    for (auto& v : hx) v = std::fabs(G(rng)) * (rng() % 8 == 0 ? 20.0f : 1.0f);
    for (auto& v : hw) v = 0.011f * G(rng);
    for (auto& v : hb) v = 0.01f * G(rng);
    for (auto& v : hp) v = (rng() % 2) ? 1e-3f * G(rng) : 0.0f;
    float *x, *w, *bias, *p, *y0, *y1, *g0, *g1;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&bias, H * 4)); CK(hipMalloc(&p, hp.size() * 4));
    CK(hipMalloc(&y0, B * 1024 * 4)); CK(hipMalloc(&y1, B * 1024 * 4)); CK(hipMalloc(&g0, T * 1024 * 4)); CK(hipMalloc(&g1, T * 1024 * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), H * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    void *xa, *xb, *wb, *pb;
    CK(hipMalloc(&xa, mdnvae_split_operand_bytes(Bp, T))); CK(hipMalloc(&xb, mdnvae_split_operand_bytes(T, Bp)));
    CK(hipMalloc(&wb, mdnvae_split_operand_bytes(H, T))); CK(hipMalloc(&pb, mdnvae_split_operand_bytes(H, Bp)));
    CK(hipMemset(xa, 0, mdnvae_split_operand_bytes(Bp, T)));
    RC(mdnvae_split_rows(x, B, T, T, 0, xa, 0));
    RC(mdnvae_split_rows(w, H, T, T, 0, wb, 0));
    RC(mdnvae_split_cols(x, B, T, T, 0, xb, 0));
    RC(mdnvae_split_cols(p, B, H, H, 0, pb, 0));
    RC(mdnvae_split_gemm(xa, wb, Bp, H, T, B, 1024, bias, bias + 1024, 1, 0, y0, y1, 1024, 1024, 256, 6, 0));
    const long Kw = (B + 15) / 16 * 16;
    RC(mdnvae_split_gemm(xb, pb, T, H, Kw, T, 1024, nullptr, nullptr, 0, 0, g0, g1, 1024, 1024, 128, 6, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> hy0(B * 1024), hy1(B * 1024), hg0(T * 1024), hg1(T * 1024);
    CK(hipMemcpy(hy0.data(), y0, hy0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy1.data(), y1, hy1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hg0.data(), g0, hg0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hg1.data(), g1, hg1.size() * 4, hipMemcpyDeviceToHost));
    // sampled entries against float64, error relative to sum |a b| (what fp32 accumulation is measured in)
    double worst_f = 0, worst_g = 0, worst_f32 = 0;
    for (int s = 0; s < 3000; ++s) {
        const long m = rng() % B, n = rng() % H;
        double acc = hb[n], mag = std::fabs(hb[n]);
        float acc32 = 0.0f;
        for (long k = 0; k < T; ++k) { const double t = (double)hx[m * T + k] * hw[n * T + k]; acc += t; mag += std::fabs(t); acc32 = fmaf(hx[m * T + k], hw[n * T + k], acc32); }
        const double want = acc > 0 ? acc : 0.0;
        const float got = n < 1024 ? hy0[m * 1024 + n] : hy1[m * 1024 + n - 1024];
        worst_f = std::fmax(worst_f, std::fabs(got - want) / mag);
        worst_f32 = std::fmax(worst_f32, std::fabs((double)acc32 + hb[n] - acc) / mag);
    }
    for (int s = 0; s < 600; ++s) {
        const long t = rng() % T, n = rng() % H;
        double acc = 0, mag = 0;
        for (long b = 0; b < B; ++b) { const double v = (double)hx[b * T + t] * hp[b * H + n]; acc += v; mag += std::fabs(v); }
        const float got = n < 1024 ? hg0[t * 1024 + n] : hg1[t * 1024 + n - 1024];
        worst_g = std::fmax(worst_g, std::fabs(got - acc) / (mag + 1e-300));
    }
    printf("B %ld: forward max |err| / sum|ab| = %.3e (a plain fp32 fma chain on the host: %.3e), gradient %.3e\n", B, worst_f, worst_f32, worst_g);
    const double fl = 2.0 * B * H * T;
    float ms;
    ms = time_ms([&] { mdnvae_split_rows(x, B, T, T, 0, xa, 0); });
    printf("split_rows x      : %.4f ms  (%.0f GB/s)\n", ms, (B * T * 4.0 + B * T * 6.0) / ms / 1e6);
    ms = time_ms([&] { mdnvae_split_cols(x, B, T, T, 0, xb, 0); });
    printf("split_cols x      : %.4f ms  (%.0f GB/s)\n", ms, (B * T * 4.0 + B * T * 6.0) / ms / 1e6);
    ms = time_ms([&] { mdnvae_split_rows(w, H, T, T, 0, wb, 0); });
    printf("split_rows w      : %.4f ms\n", ms);
    ms = time_ms([&] { mdnvae_split_cols(p, B, H, H, 0, pb, 0); });
    printf("split_cols p      : %.4f ms  (%.0f GB/s)\n", ms, (B * H * 10.0) / ms / 1e6);
    for (int tile : {256, 128}) {
        ms = time_ms([&] { mdnvae_split_gemm(xa, wb, Bp, H, T, B, 1024, bias, bias + 1024, 1, 0, y0, y1, 1024, 1024, tile, 6, 0); });
        printf("forward  tile %d : %.4f ms  fp32-equivalent %.1f TFLOP/s, bf16 MFMA %.0f TFLOP/s\n", tile, ms, fl / ms / 1e9, 6 * fl / ms / 1e9);
        ms = time_ms([&] { mdnvae_split_gemm(xb, pb, T, H, Kw, T, 1024, nullptr, nullptr, 0, 0, g0, g1, 1024, 1024, tile, 6, 0); });
        printf("gradient tile %d : %.4f ms  fp32-equivalent %.1f TFLOP/s, bf16 MFMA %.0f TFLOP/s\n", tile, ms, fl / ms / 1e9, 6 * fl / ms / 1e9);
    }
    return 0;
}
