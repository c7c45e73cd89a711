// Tuning harness: times mdnvae_vae_loss_fwd / _bwd / mdn loss through the C ABI with hipEvents.
// Built against loss.hip compiled with different -D settings (see tools/loss_tune.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/mdnvae_hip.h"

int main(int argc, char** argv) {
    long B = argc > 1 ? atol(argv[1]) : 8192;
    long N = argc > 2 ? atol(argv[2]) : 87;
    long D = N * N, L = 68;
    float *r, *x, *g, *mu, *lv, *gm, *gl, *out;
    void* ws;
    hipMalloc(&r, B * D * 4); hipMalloc(&x, B * D * 4); hipMalloc(&g, B * D * 4);
    hipMalloc(&mu, B * L * 4); hipMalloc(&lv, B * L * 4); hipMalloc(&gm, B * L * 4); hipMalloc(&gl, B * L * 4);
    hipMalloc(&out, 64); hipMalloc(&ws, mdnvae_loss_workspace_bytes()); hipMemset(ws, 0, mdnvae_loss_workspace_bytes());
    hipMemset(r, 0, B * D * 4); hipMemset(x, 0, B * D * 4); hipMemset(mu, 0, B * L * 4); hipMemset(lv, 0, B * L * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 3; ++which) {
        float best = 1e9, sum = 0; int iters = 20;
        for (int it = -3; it < iters; ++it) {
            hipEventRecord(e0, 0);
            int rc = 0;
            if (which == 0) rc = mdnvae_vae_loss_fwd(1, r, x, B, D, mu, lv, L, nullptr, nullptr, ws, out, 0);
            if (which == 1) rc = mdnvae_vae_loss_bwd(1, r, x, B, D, mu, lv, L, nullptr, nullptr, nullptr, g, gm, gl, 0);
            if (which == 2) rc = mdnvae_vae_loss_fwdbwd(1, r, x, B, D, mu, lv, L, g, gm, gl, ws, out, 0);
            if (rc) { printf("rc=%d %s\n", rc, mdnvae_last_error()); return 1; }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 0) { sum += ms; if (ms < best) best = ms; }
        }
        double bytes = which == 0 ? 4.0 * B * (2 * D + 2 * L) : 4.0 * B * (3 * D + 4 * L);
        printf("%s B=%ld N=%ld avg %.4f ms %.0f GB/s best %.0f GB/s\n", which == 0 ? "fwd" : which == 1 ? "bwd" : "fwdbwd", B, N,
               sum / iters, bytes / (sum / iters) / 1e6, bytes / best / 1e6);
    }
    // trait-gated mixture loss, K components
    const int K = argc > 3 ? atoi(argv[3]) : 5;
    float *comp, *gcomp, *logits, *lsig, *resp, *ss, *glog, *glsr; void* mws;
    long pad = argc > 4 ? atol(argv[4]) : 0; hipMalloc(&comp, B * K * D * 4); hipMalloc(&gcomp, B * K * D * 4 + pad); gcomp = (float*)((char*)gcomp + pad); hipMalloc(&logits, B * K * 4);
    hipMalloc(&lsig, K * 4); hipMalloc(&resp, B * K * 4); hipMalloc(&ss, B * K * 4); hipMalloc(&glog, B * K * 4);
    hipMalloc(&glsr, B * K * 4); hipMalloc(&mws, mdnvae_mdn_workspace_bytes((int)B));
    hipMemset(comp, 0, B * K * D * 4); hipMemset(logits, 0, B * K * 4); hipMemset(lsig, 0, K * 4);
    for (int which = 0; which < 2; ++which) {
        float best = 1e9, sum = 0; int iters = 20;
        for (int it = -3; it < iters; ++it) {
            hipEventRecord(e0, 0);
            int rc = 0;
            if (which == 0) rc = mdnvae_mdn_loss_fwd(comp, x, logits, lsig, (int)B, K, (int)D, mu, lv, (int)L, nullptr, nullptr, resp, ss, mws, out, 0);
            if (which == 1) rc = mdnvae_mdn_loss_bwd(comp, x, logits, lsig, resp, ss, (int)B, K, (int)D, mu, lv, (int)L, nullptr, gcomp, glog, glsr, gm, gl, mws, 0);
            if (rc) { printf("rc=%d %s\n", rc, mdnvae_last_error()); return 1; }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 0) { sum += ms; if (ms < best) best = ms; }
        }
        double bytes = which == 0 ? 4.0 * B * ((K + 1) * D + 2 * L + 3 * K) : 4.0 * B * ((2 * K + 1) * D + 4 * L + 5 * K);
        printf("mdn_%s B=%ld N=%ld K=%d avg %.4f ms %.0f GB/s best %.0f GB/s\n", which == 0 ? "fwd" : "bwd", B, N, K,
               sum / iters, bytes / (sum / iters) / 1e6, bytes / best / 1e6);
    }
    return 0;
}
