// Timing / agreement harness for the rank-one mixture likelihood pass (mdnvae_mdn_outer_fwdbwd):
// the symmetric-target sweep against the general sweep on the same symmetric input, then the
// symmetric sweep timed with HIP events.   usage: mdn_tune [B] [N] [iters]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <algorithm>
#include "../include/mdnvae_hip.h"

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 87, iters = argc > 3 ? atoi(argv[3]) : 20;
    const int K = 5, L = 68, NP = MDNVAE_DEC_NP;
    const size_t D = (size_t)N * N;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::normal_distribution<float> G(0.0f, 1.0f);
    std::vector<float> hx(B * D), hn((size_t)B * K * NP, 0.0f), hl((size_t)B * K), hs(K), hmu((size_t)B * L), hlv((size_t)B * L);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i)
            for (int j = i; j < N; ++j) {
                const float v = U(rng) * (b % 3 + 1);
                hx[b * D + (size_t)i * N + j] = hx[b * D + (size_t)j * N + i] = v;
            }
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < N; ++c) hn[((size_t)b * K + k) * NP + c] = U(rng);
    for (auto& v : hl) v = G(rng);
    for (auto& v : hs) v = 1.0f + 0.1f * G(rng);
    for (auto& v : hmu) v = G(rng);
    for (auto& v : hlv) v = 0.3f * G(rng);
    float *x, *nodes, *logits, *ls, *mu, *lv, *out;
    struct Out { float *gn, *gl, *gs, *gmu, *glv, *resp, *loss; } o[2];
    CK(hipMalloc(&x, hx.size() * 4 + 64));
    x += 1;                                   // an odd 4-byte phase for the whole array on purpose when N is odd anyway
    CK(hipMalloc(&nodes, hn.size() * 4)); CK(hipMalloc(&logits, hl.size() * 4)); CK(hipMalloc(&ls, K * 4));
    CK(hipMalloc(&mu, hmu.size() * 4)); CK(hipMalloc(&lv, hlv.size() * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(nodes, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(logits, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ls, hs.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(mu, hmu.data(), hmu.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lv, hlv.data(), hlv.size() * 4, hipMemcpyHostToDevice));
    void* ws;
    CK(hipMalloc(&ws, mdnvae_mdn_workspace_bytes(B)));
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&o[v].gn, hn.size() * 4)); CK(hipMalloc(&o[v].gl, hl.size() * 4)); CK(hipMalloc(&o[v].gs, hl.size() * 4));
        CK(hipMalloc(&o[v].gmu, hmu.size() * 4)); CK(hipMalloc(&o[v].glv, hmu.size() * 4)); CK(hipMalloc(&o[v].resp, hl.size() * 4));
        CK(hipMalloc(&o[v].loss, 16));
        CK(hipMemset(o[v].gn, 0xff, hn.size() * 4));
        int rc = mdnvae_mdn_outer_fwdbwd(nodes, x, logits, ls, B, K, N, mu, lv, L, o[v].gn, o[v].gl, o[v].gs, o[v].gmu, o[v].glv,
                                         o[v].resp, v == 0, ws, o[v].loss, 0);
        if (rc) { printf("call failed %d: %s\n", rc, mdnvae_last_error()); return 1; }
        CK(hipDeviceSynchronize());
    }
    (void)out;
    auto cmp = [&](const char* name, float* a, float* b, size_t n) {
        std::vector<float> ha(n), hb(n);
        hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0; size_t bad = 0;
        for (size_t i = 0; i < n; ++i) {
            if (!std::isfinite(ha[i])) ++bad;
            md = std::fmax(md, std::fabs((double)ha[i] - hb[i]));
            mx = std::fmax(mx, std::fabs((double)hb[i]));
        }
        printf("  %-8s max|sym-general| %.3e  of max %.3e  (rel %.2e)  nonfinite %zu\n", name, md, mx, md / (mx + 1e-300), bad);
    };
    printf("B %d N %d K %d\n", B, N, K);
    cmp("loss", o[0].loss, o[1].loss, 3);
    cmp("gnodes", o[0].gn, o[1].gn, hn.size());
    cmp("glogits", o[0].gl, o[1].gl, hl.size());
    cmp("glsig", o[0].gs, o[1].gs, hl.size());
    cmp("resp", o[0].resp, o[1].resp, hl.size());
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int sym = 1; sym >= 0; --sym) {
        std::vector<float> ts;
        for (int it = 0; it < iters + 3; ++it) {
            hipEventRecord(e0, 0);
            mdnvae_mdn_outer_fwdbwd(nodes, x, logits, ls, B, K, N, mu, lv, L, o[0].gn, o[0].gl, o[0].gs, o[0].gmu, o[0].glv, o[0].resp,
                                    sym, ws, o[0].loss, 0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 3) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        const double bytes = 4.0 * B * (D + 2.0 * K * NP + 4.0 * L + 4.0 * K);
        printf("%s: median %.4f ms  min %.4f ms  -> %.0f GB/s median (%.3f of 8 TB/s), incl. the finalize launch\n",
               sym ? "symmetric sweep" : "general sweep  ", ts[ts.size() / 2], ts[0], bytes / ts[ts.size() / 2] / 1e6,
               bytes / ts[ts.size() / 2] / 1e6 / 8000.0);
    }
    return 0;
}
