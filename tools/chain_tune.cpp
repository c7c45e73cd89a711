// Tuning harness: times the chain stage of the decoder (node vectors only, no matrix stage) through
// the C ABI with hipEvents.  Built against decoder.hip compiled with different -D settings.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/mdnvae_hip.h"

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8192;
    const int N = argc > 2 ? atoi(argv[2]) : 87;
    const int saved = argc > 3 ? atoi(argv[3]) : 1;          // forward keeps the layer inputs, backward does not recompute
    const int L = 68, R = 5, NL = 4, KP = MDNVAE_DEC_KP, NP = MDNVAE_DEC_NP;
    const int acts = (1 << 2) | (1 << 4) | (1 << 6);          // linear, then three sigmoid layers
    const size_t panel = (size_t)R * NL * KP * NP;
    std::vector<float> h(panel);
    for (size_t i = 0; i < panel; ++i) h[i] = 0.02f * (float)((i * 2654435761u >> 8) % 201) / 100.0f - 0.02f;
    float *z, *wt, *wn, *nodes, *gnodes, *gz, *hin, *dp;
    hipMalloc(&z, (size_t)B * L * 4); hipMalloc(&wt, panel * 4); hipMalloc(&wn, panel * 4);
    hipMalloc(&nodes, (size_t)B * R * NP * 4); hipMalloc(&gnodes, (size_t)B * R * NP * 4); hipMalloc(&gz, (size_t)B * L * 4);
    const size_t rows = mdnvae_decoder_bwd_rows_floats(B, R, NL);
    hipMalloc(&hin, rows * 4); hipMalloc(&dp, (size_t)R * NL * B * NP * 4);
    hipMemcpy(wt, h.data(), panel * 4, hipMemcpyHostToDevice); hipMemcpy(wn, h.data(), panel * 4, hipMemcpyHostToDevice);
    hipMemset(z, 0, (size_t)B * L * 4); hipMemset(gnodes, 0, (size_t)B * R * NP * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        float best = 1e9, sum = 0; const int iters = 20;
        for (int it = -3; it < iters; ++it) {
            hipEventRecord(e0, 0);
            int rc;
            if (which == 0)
                rc = mdnvae_decoder_fwd(z, wt, B, N, L, R, NL, acts, 0, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, nullptr,
                                        nodes, saved ? hin : nullptr, 0);
            else
                rc = mdnvae_decoder_bwd(z, wt, wn, B, N, L, R, NL, acts, 0, nullptr, nullptr, 0, nullptr, 0, 0, nullptr,
                                        nullptr, gz, hin, dp, nullptr, nodes, gnodes, saved, nullptr, 0);
            if (rc) { printf("rc=%d %s\n", rc, mdnvae_last_error()); return 1; }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 0) { sum += ms; if (ms < best) best = ms; }
        }
        const double flop = 2.0 * B * R * NL * KP * NP * (which ? (saved ? 1 : 2) : 1);
        printf("chain_%s B=%d N=%d avg %.1f us best %.1f us  %.1f TFLOP/s\n", which ? "bwd" : "fwd", B, N, sum / iters * 1e3,
               best * 1e3, flop / (sum / iters) / 1e9);
    }
    return 0;
}
