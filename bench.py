#!/usr/bin/env python3
"""Training-throughput benchmark of the MI355X connectome-VAE path.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = zero_grad + forward + loss + backward + SUM all-reduce + Adam on one batch of
synthetic 87x87 connectomes that is already resident in HBM.  Weak scaling: every rank trains
on its own shard with the same per-GPU batch.  Rank 0 prints ONE JSON line.

The default workload is the trait-gated mixture VAE (FNN encoder, GCN decoder, fused MDN/KL loss;
`--model mdn_gate`).  On one GPU the line also carries, under "also", a short run of the conv-encoder
VAE (`--model cnn_gcn`), whose step is dominated by the hand-written MFMA convolutions.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import mdnvae_amd as M  # noqa: E402
from mdnvae_amd import ops  # noqa: E402

PEAK_FP32_MATRIX_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16, dense
PEAK_HBM_GBPS = 8000.0
# fabric-side bytes / algorithmic bytes from the rocprofv3 TCC counters (FETCH_SIZE x2 + WRITE_SIZE,
# separate passes; profiles/r01_c_loss_pmc_traffic.md, profiles/r01_h_pmc_traffic.md); kernels without a collected pass report null
TRAFFIC_RATIO = {"loss_stream_kernel": 1.00, "mdn_fwd_kernel": 1.10, "mdn_bwd_stream_kernel": 1.27,
                 "mdn_outer_kernel": 1.03, "tri_pack_kernel": 1.01, "adam_kernel": 1.00}

# fabric-side bytes per launch of the matrix-core kernels, collected the same way at the default workload's shapes
# (profiles/r02_e_mdn_gate_B16384_counters.csv, r02_g_...); keyed by the launch's executed flops so that another shape reports null
TRAFFIC_BYTES = {("split_gemm_kernel", 12.0 * 16384 * 2048 * 3840): (1207.6 + 208.7) * 1e6,
                 # profiles/r02_g_mdn_gate_B16384_counters.csv (the four-stage ring)
                 ("split_gemm_kernel<3 fp16 terms>", 6.0 * 16384 * 2048 * 3840): (755.7 + 134.2) * 1e6,
                 ("split_gemm_kernel<3 terms>", 6.0 * 3829 * 2048 * 16384): (1443.1 + 62.7) * 1e6}

REHEARSE_EXCHANGE = False

MODELS = {
    # name: (constructor kwargs, loss mode, data kind)
    "cnn_gcn": (dict(decoder="gcn", intercept="diag", final="none"), "poisson_log", "sc"),   # function/VAE-CNN-GCN.py
    "cnn_nn": (dict(decoder="nn", intercept="dense", final="exp"), "poisson_rate", "sc"),    # function/VAE_CNN_NN.py
    "cnn_gcn_fc": (dict(decoder="gcn", intercept="diag", final="none"), "mse", "fc"),        # function_FC/VAE_CNN_GCN.py
    "mm_gate": (None, "poisson_log", "sc"),                                                  # function/MM_Gate.py
    "mdn_gate": (None, "mdn", "sc"),          # MM_Gate with the trait-gated mixture head (build-defined)
}


def build(model_name, n_nodes, device, symmetric=True, precision="fp32"):
    """`symmetric`: the synthetic connectomes are symmetric like the prepared data of the scripts,
    so the FNN-encoder models run their first layers on the packed triangle (same sums, half the
    GEMM work; the symmetry is verified on the device every step and checked after the run)."""
    kw, mode, kind = MODELS[model_name]
    torch.manual_seed(11111)
    if model_name == "mm_gate":
        model = M.GCNVAE(R=5, n_nodes=n_nodes, k=32, n_layers=3, latent_dim=68, symmetric_input=symmetric)
    elif model_name == "mdn_gate":
        model = M.MDNGateVAE(R=5, n_nodes=n_nodes, k=32, n_layers=3, latent_dim=68, n_traits=4,
                             symmetric_input=symmetric)
    else:
        model = M.ConvVAE(n_nodes=n_nodes, latent_dim=68, precision=precision, **kw)
    return model.to(device), mode, kind


def rank_mask_from(x, n_nodes, n_size=32):
    """The scripts' mask recipe on the synthetic group-mean matrix (function/VAE-CNN-GCN.py:209)."""
    a = x.mean(dim=0).reshape(n_nodes, n_nodes).double().cpu()
    order = torch.argsort(torch.argsort(a, dim=-1), dim=-1)
    return (order < n_size + 1).float()


def knn_adjacency_from(x, n_nodes, k=32):
    """function/MM_Gate.py:55-65 on the synthetic group-mean matrix."""
    a = x.mean(dim=0).reshape(n_nodes, n_nodes).double().cpu()
    a = a + a.t()
    adj = []
    for u in range(n_nodes):
        desc = torch.argsort(a[u], descending=True)[:k].tolist()
        desc = [v for v in desc if v != u][:k - 1]
        adj.append([u] + desc)
    return adj


def cpu_baseline(model_name, n_nodes, seconds_budget=20.0):
    """The eager-PyTorch restatement of the reference training step (oracle/ref_eager.py),
    timed on the host cores on a bounded sample of the same workload."""
    from oracle import ref_eager as R
    torch.manual_seed(11111)
    kw, mode, kind = MODELS[model_name]
    B = 32
    gen = M.data.synthetic_sc if kind == "sc" else M.data.synthetic_fc
    x = gen(B, n_nodes, seed=1234, device="cuda").cpu()      # the data preparation itself is a device kernel
    if model_name == "mm_gate":
        ref = M.GCNVAE(R=5, n_nodes=n_nodes, k=32, n_layers=3, latent_dim=68)
        adjacency = knn_adjacency_from(x, n_nodes)
    elif model_name == "mdn_gate":
        ref = M.MDNGateVAE(R=5, n_nodes=n_nodes, k=32, n_layers=3, latent_dim=68, n_traits=4)
        adjacency = knn_adjacency_from(x, n_nodes)
        traits = torch.randn(B, 4)
    else:
        ref = M.ConvVAE(n_nodes=n_nodes, latent_dim=68, **kw)
        mask = rank_mask_from(x, n_nodes) if kw["decoder"] == "gcn" else None
        intercept = {"diag": "diagvec", "dense": "dense", "none": "none"}[kw["intercept"]]
    p = {k: v.detach().clone().requires_grad_(True) for k, v in ref.state_dict().items()}
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()}

    def loss_of(eps):
        if model_name == "mm_gate":
            recon, mu, lv, _, _ = R.vae_forward("mm_gate", p, x, eps, adjacency=adjacency)
        elif model_name == "mdn_gate":
            mu, lv = R.fnn_encode(p, x)
            z = R.reparameterize(mu, lv, eps)
            comp = R.gcn_components(p, z, adjacency)
            logits = R.mdn_gate_logits(traits, p["gate.weight"], p["gate.bias"])
            return R.mdn_nll(comp, p["log_sigma"], logits, x) + R.kl_term(mu, lv)
        else:
            mu, lv = R.conv_encode(p, x, n_nodes)
            z = R.reparameterize(mu, lv, eps)
            recon, _ = R.rank1_decode(p, z, mask=mask, intercept=intercept, final=kw["final"], n_nodes=n_nodes)
        return R.vae_loss(recon, x, mu, lv, mode)

    steps, t_used = 0, 0.0
    while steps < 2 or (t_used < seconds_budget and steps < 50):
        eps = torch.randn(B, 68)
        t0 = time.perf_counter()
        for v in p.values():
            v.grad = None
        loss_of(eps).backward()
        with torch.no_grad():
            R.adam_step(p, {k: v.grad for k, v in p.items()}, state, steps + 1)
        dt = time.perf_counter() - t0
        if steps >= 1:                      # the first step warms the allocator and the thread pool
            t_used += dt
        steps += 1
    timed = steps - 1
    return dict(value=B * timed / t_used, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{timed} steps of batch {B} (eager-PyTorch restatement of the reference step, fp32)")


class KernelClock:
    """HIP-event bracket (torch's current stream = the stream the launches go to) around every tagged
    call during the timed region."""

    def __init__(self):
        self.records = {}

    def bracket(self, tag, work, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.records.setdefault(tag, []).append((a, b, work))
        return out

    def summary(self):
        out = {}
        for tag, recs in self.records.items():
            t = sum(a.elapsed_time(b) for a, b, _ in recs) * 1e-3
            w = sum(wk for _, _, wk in recs)
            out[tag] = dict(launches=len(recs), seconds=t, work=w)
        return out


def _decoder_work(args, backward):
    """Algorithmic HBM bytes of one chain-decoder call: the [B,(R,)N*N] matrix it writes (forward) or
    reads (backward) - or, when the caller works on the node vectors themselves, those [B,R,NP] -
    plus the per-layer rows the backward emits."""
    B, N, L, R, NL = args[2 + backward], args[3 + backward], args[4 + backward], args[5 + backward], args[6 + backward]
    components = args[14 + backward]
    matrix = args[15 + backward]                       # out (forward) / grad_out (backward)
    if matrix is None:
        # chain stage alone: matrix-core work, one 96x96 panel product per (sample, channel, layer)
        # forward; recompute + dh.W backward, plus the weight-gradient operands it emits
        saved = backward and bool(args[-3])               # layer_in_saved: no recomputation
        return 2.0 * B * R * NL * ops.DEC_KP * ops.DEC_NP * (2 if backward and not saved else 1)
    rows = R if components else 1
    extra = R * NL * (ops.DEC_KP + ops.DEC_NP) + L if backward else L
    return 4.0 * B * (rows * N * N + R * ops.DEC_NP + extra)


def _mdn_outer_work(a):
    """(nodes, x, logits, log_sigma, B, K, N, mu, logvar, L, ...): x once, node vectors in, their
    gradient out, the latent rows in and their gradients out."""
    B, K, N, L = a[4], a[5], a[6], a[9]
    return 4.0 * B * (N * N + 2 * K * ops.DEC_NP + 4 * L + 4 * K)


CHAIN_ROWS_MIN_B = 8192       # from this batch size the chain decoder runs its wave-per-16-samples kernels

# C-ABI entry point -> (tag = the kernel(s) it launches, possibly depending on the arguments; work of one call)
ABI_TAGS = {
    "mdnvae_vae_loss_fwdbwd": ("loss_stream_kernel", lambda a: 4.0 * a[3] * (3 * a[4] + 4 * a[7])),
    "mdnvae_mdn_loss_fwd": ("mdn_fwd_kernel", lambda a: 4.0 * a[4] * ((a[5] + 1) * a[6] + 2 * a[9] + 3 * a[5])),
    "mdnvae_mdn_loss_bwd": ("mdn_bwd_stream_kernel", lambda a: 4.0 * a[6] * ((2 * a[7] + 1) * a[8] + 4 * a[11] + 5 * a[7])),
    "mdnvae_mdn_outer_fwdbwd": ("mdn_outer_kernel", _mdn_outer_work),
    # (workspace, B, loss_out, rows, K, ...): the per-sample pairs and the [B,K] log-sigma rows, one read each
    "mdnvae_mdn_finalize": ("mdn_finalize_kernel", lambda a: a[1] * (16.0 + 4.0 * (a[4] if a[3] is not None else 0))),
    # (mode, nodes, x, B, R, N, mu, logvar, L, ...): x once, node vectors in, their gradient out, latent rows
    "mdnvae_outer_loss_fwdbwd": ("outer_loss_kernel",
                                 lambda a: 4.0 * a[3] * (a[5] * a[5] + 2 * a[4] * ops.DEC_NP + 4 * a[8])),
    "mdnvae_decoder_fwd": (lambda a: "chain_fwd_mfma_kernel+outer_epilogue_kernel" if a[15] is not None
                           else (("chain_fwd_regs_kernel" if a[6] <= 4 and a[12] is None and a[13] == 0 else "chain_fwd_rows_kernel")
                                 if a[2] >= CHAIN_ROWS_MIN_B else "chain_fwd_mfma_kernel"),
                           lambda a: _decoder_work(a, 0)),
    "mdnvae_decoder_bwd": (lambda a: "outer_grad_kernel+chain_bwd_mfma_kernel" if a[16] is not None
                           else ((("chain_bwd_regs_kernel" if a[7] <= 4 and a[13] is None and a[14] == 0 and a[-2] is not None
                                   else "chain_bwd_rows_kernel") if a[3] >= CHAIN_ROWS_MIN_B else "chain_bwd_saved_kernel")
                                 if a[-3] else "chain_bwd_mfma_kernel"),
                           lambda a: _decoder_work(a, 1)),
    "mdnvae_adam_step": ("adam_kernel", lambda a: 28.0 * a[4]),
    # (param, grad_t, m_t, v_t, H, N, ...): the weight in and out, the folded gradient in, both folded moments in and out
    "mdnvae_adam_step_folded": ("adam_folded_moments_kernel+apply_folded_step_kernel",
                                lambda a: 4.0 * a[4] * (2 * a[5] * a[5] + 7 * (a[5] * (a[5] + 1) // 2))),
    # args: (x, B, N, ld, out, asym, stream): read N*N, write the triangle, per matrix
    "mdnvae_tri_pack": ("tri_pack_kernel", lambda a: 4.0 * a[1] * (a[2] * a[2] + a[2] * (a[2] + 1) // 2)),
}
HBM_TAGS = {v[0] for v in ABI_TAGS.values() if isinstance(v[0], str)} | {
    "chain_fwd_mfma_kernel+outer_epilogue_kernel", "outer_grad_kernel+chain_bwd_mfma_kernel"}
ABI_TAGS.update({
    # (src, R, K, ld, ...) / (src, Kd, R, ld, ...): the fp32 source in, three bf16 parts out
    "mdnvae_split_rows": ("split_rows_kernel", lambda a: 10.0 * a[1] * a[2]),
    "mdnvae_split_cols": ("split_cols_kernel", lambda a: 10.0 * a[1] * a[2]),
    # (src, gate, Kd, R, ld, ones_row, parts, ...): the source (and its gate) in, `parts` bf16 parts out
    # (gpre, w, dense, gate, Kd, R, J, parts, ...): grad_pre and the gate in (+ the addend), `parts` bf16 parts out
    "mdnvae_split_cols_dgrad": ("split_cols_dgrad_kernel",
                                lambda a: a[4] * (4.0 * a[6] + (4.0 + (4.0 if a[2] is not None else 0.0) + 2.0 * a[7]) * a[5])),
    "mdnvae_split_cols_gated": ("split_cols_kernel", lambda a: (4.0 + (4.0 if a[1] is not None else 0.0) + 2.0 * a[6]) * a[2] * a[3]),
    # (y, grad_y, grad_pre, n, act): two reads, one write
    "mdnvae_act_bwd": ("act_bwd_kernel", lambda a: 12.0 * a[3]),
    "mdnvae_tri_fold": ("tri_pack_kernel<fold>", lambda a: 4.0 * a[1] * (a[2] * a[2] + a[2] * (a[2] + 1) // 2)),
})
HBM_TAGS |= {"split_rows_kernel", "split_cols_kernel", "split_cols_dgrad_kernel", "act_bwd_kernel", "tri_pack_kernel<fold>"}
# bf16 matrix-core work: (a, b, M, N, K, m_valid, ...) executes six bf16 MFMA products per fp32 multiply-add
BF16_TAGS = {"split_gemm_kernel", "split_gemm_kernel<3 terms>", "conv_bf16_kernel", "conv_wgrad_bf16_kernel"}
# kernels that reach fp32 accuracy with several bf16 products per multiply: rated in fp32-equivalent
# flops against the bf16 peak divided by their number of products
SPLIT_PRODUCTS = {"conv_bf16x2_kernel": 3, "conv_bf16x3_kernel": 6, "conv_wgrad_bf16x2_kernel": 3, "conv_wgrad_bf16x3_kernel": 6}
# (a, b, M, N, K, m_valid, n_valid, ..., n_split at 13, ld_out, tile_m, terms at 16): bf16 matrix-core flops executed,
# 2 * terms per fp32 multiply-add; the three-term form (weight gradients) is rated as a kernel of its own
ABI_TAGS["mdnvae_split_gemm"] = (lambda a: "split_gemm_kernel" if a[16] == 6 else "split_gemm_kernel<3 terms>",
                                 lambda a: 2.0 * a[16] * a[5] * min(a[3], a[6] * (a[3] // a[13])) * a[4])
# (a, b, absmax_a, absmax_b, M, N, K, m_valid, n_valid, ..., n_split at 15): three fp16 products per fp32 multiply-add
ABI_TAGS["mdnvae_split_gemm_f16"] = (lambda a: "split_gemm_kernel<3 fp16 terms>" if a[18] < 0
                                     else "split_gemm_kernel<3 fp16 terms, weight gradient>",
                                     lambda a: 6.0 * a[7] * min(a[5], a[8] * (a[5] // a[15])) * a[6])
BF16_TAGS |= {"split_gemm_kernel<3 fp16 terms>", "split_gemm_kernel<3 fp16 terms, weight gradient>"}   # fp16 rate = bf16 rate
ABI_TAGS.update({
    # (src, R, K, ld, ...): the fp32 source in, two fp16 parts out; (src, gate, Kd, R, ld, ...) likewise
    "mdnvae_split_rows_f16": ("split_rows_kernel<fp16>", lambda a: 8.0 * a[1] * a[2]),
    "mdnvae_split_cols_f16": ("split_cols_kernel<fp16>", lambda a: (8.0 + (4.0 if a[1] is not None else 0.0)) * a[2] * a[3]),
    # (src, Kd, R, ld, ones_row, parts, ...): the batch in once, two fp16 parts and `parts` bf16 parts out
    "mdnvae_split_rows_f16_cols": ("split_both_kernel", lambda a: (8.0 + 2.0 * (a[5] or 2)) * a[1] * a[3]),
    # (gpre, w, dense, gate, Kd, R, J, ...): grad_pre and the gate in (+ the addend), two fp16 parts out
    "mdnvae_split_cols_dgrad_f16": ("split_cols_dgrad_kernel",
                                    lambda a: a[4] * (4.0 * a[6] + (8.0 + (4.0 if a[2] is not None else 0.0)) * a[5])),
    "mdnvae_absmax": ("absmax_kernel", lambda a: 4.0 * a[1] * a[2]),
    "mdnvae_tri_pack_absmax": ("tri_pack_kernel", lambda a: 4.0 * a[1] * (a[2] * a[2] + a[2] * (a[2] + 1) // 2)),
    "mdnvae_tri_fold_absmax": ("tri_pack_kernel<fold>", lambda a: 4.0 * a[1] * (a[2] * a[2] + a[2] * (a[2] + 1) // 2)),
})
HBM_TAGS |= {"split_rows_kernel<fp16>", "split_cols_kernel<fp16>", "absmax_kernel", "split_both_kernel"}
# matrix-core entry points with a fixed tag (work in flops; kept out of HBM_TAGS)
ABI_TAGS.update({
    # (layer_in, layer_dpre, B, count, ...): one 96 x 96 x B product per (channel, layer)
    "mdnvae_decoder_wgrad": ("panel_wgrad_partial_kernel+tsgemm_reduce_kernel",
                             lambda a: 2.0 * a[2] * a[3] * ops.DEC_KP * ops.DEC_NP),
    # (grad_pre, x, B, N, M, ...)
    "mdnvae_linear_wgrad": ("tsgemm_partial_kernel+tsgemm_reduce_kernel", lambda a: 2.0 * a[2] * a[3] * a[4]),
    "mdnvae_linear_wgrad_bias": ("tsgemm_partial_kernel+tsgemm_reduce_kernel", lambda a: 2.0 * a[2] * a[3] * a[4]),
})


# ---- channel-blocked conv encoder (bf16 operands): flops of the convolution the launch stands for ----
# mdnvae_cb_conv(act, part_stride, w, bias, out, out16, B, Cout, Cin, kind, rows, pitch, ...): square planes, `rows` output
# rows per side (the padded grid, two more, for the stride-1 data gradient)
_CB_KIND = {0: "cb_conv_kernel<forward>", 1: "cb_conv_kernel<forward, stride 2>", 2: "cb_conv_kernel<data gradient>",
            3: "cb_conv_kernel<data gradient, stride 2>"}
# parts (argument 13 of both products) = 2: two fp16 parts per operand, three matrix-core products per multiply-add; such a
# launch is rated like the other split products: fp32-equivalent flops against a third of the 16-bit peak
_TWO = ", two fp16 parts"


def _cb_tag(base, parts):
    return base if parts == 1 else base[:-1] + _TWO + ">" if base.endswith(">") else base + "<" + _TWO[2:] + ">"


# (x, grad_out, workspace, grad_w, B, H, W, Cout, KS, pad): the stem's weight gradient on bf16 operands
ABI_TAGS["mdnvae_conv_stem_wgrad_bf16"] = ("conv_stem_wgrad_kernel<bf16>", lambda a: 2.0 * a[8] * a[8] * a[7] * a[5] * a[6] * a[4])
BF16_TAGS.add("conv_stem_wgrad_kernel<bf16>")
ABI_TAGS.update({
    "mdnvae_cb_conv": (lambda a: _cb_tag(_CB_KIND[a[9]], a[13]),
                       lambda a: 18.0 * a[6] * a[7] * a[8] * (a[10] - (2 if a[9] == 2 else 0)) ** 2),
    # (g, act, workspace, grad_w, B, Cout, Cin, planes, PPg, PPa, kbeg, kend, pitch, parts, ...)
    "mdnvae_cb_wgrad": (lambda a: _cb_tag("cb_wgrad_kernel" if a[7] == 1 else "cb_wgrad_kernel<stride 2>", a[13]),
                        lambda a: 18.0 * a[4] * a[5] * a[6] * (a[12] - (2 if a[7] == 1 else 1)) ** 2),
    # (y, y16, stats, compute, skip, stream_out, act, part_stride, B, C, H, W, pitch_y, Qy, relu, mode, PP, parts, ...):
    # every buffer once, as laid out (padding included)
    "mdnvae_cb_norm_fwd": ("cb_norm_fwd_kernel",
                           lambda a: 1.0 * a[8] * a[9] * (a[13] * (2 if a[1] else 4) + (4 * a[10] * a[11] if a[4] is not None else 0)
                                                          + (4 * a[10] * a[11] if a[5] is not None else 0)
                                                          + (2 * a[17] * a[16] * (4 if a[15] == 1 else 1) if a[6] is not None else 0))),
    # (g, gmode, pitch_g, Qg, base, sum_out, y, y16, stats, pitch_y, Qy, relu, B, C, H, W, omode, dc, part_stride, PP, parts,
    #  scale, gsum, absmax_dc, absmax_probe, g16, ...); a probe run reads its inputs and writes a record only
    "mdnvae_cb_norm_bwd": (lambda a: "cb_norm_bwd_kernel" if a[24] is None else "cb_norm_bwd_kernel<probe of the largest entry>",
                           lambda a: 1.0 * a[12] * a[13] * (((2 if a[25] else 4) * a[3] * (4 if a[1] == 2 else 1) if a[0] is not None else 0)
                                                            + (4 * a[14] * a[15] if a[4] is not None else 0)
                                                            + (0 if a[24] is not None else 4 * a[14] * a[15] if a[5] is not None else 0)
                                                            + a[10] * (2 if a[7] else 4)
                                                            + (0 if a[24] is not None else 4 * a[14] * a[15] if a[16] == 2
                                                               else 2 * a[20] * a[19]))),
})
_CB_PRODUCTS = set(_CB_KIND.values()) | {"cb_wgrad_kernel", "cb_wgrad_kernel<stride 2>"}
BF16_TAGS |= _CB_PRODUCTS | {_cb_tag(t, 2) for t in _CB_PRODUCTS}
SPLIT_PRODUCTS.update({_cb_tag(t, 2): 3 for t in _CB_PRODUCTS})
ELEMENTWISE_TAGS = {"cb_norm_fwd_kernel", "cb_norm_bwd_kernel", "cb_norm_bwd_kernel<probe of the largest entry>"}
HBM_TAGS |= ELEMENTWISE_TAGS


def install_clock(clock):
    """Time every conv launch, every tagged C-ABI call and the library GEMMs of the dense layers by
    their own event pairs."""
    L = ops._ConvLayer
    orig_fwd, orig_dgrad, orig_wgrad, orig_fwd_stats = L.fwd, L.dgrad, L.wgrad, L.fwd_stats

    def flops(ly, B):
        return 2.0 * ly.ks * ly.ks * ly.cin * ly.cout * ly.oh * ly.oh * B

    split_tag = {1: "conv_bf16_kernel", 2: "conv_bf16x2_kernel", 3: "conv_bf16x3_kernel"}
    wgrad_tag = {1: "conv_wgrad_bf16_kernel", 2: "conv_wgrad_bf16x2_kernel", 3: "conv_wgrad_bf16x3_kernel"}

    def fwd(self, x, stats, relu_in, wf, bias, B, parts=0):
        tag = "conv_stem_kernel" if self.cin == 1 else split_tag.get(parts, "conv_mfma_kernel")
        return clock.bracket(tag, flops(self, B), lambda: orig_fwd(self, x, stats, relu_in, wf, bias, B, parts))

    def dgrad(self, gout, wb, B, parts=0):
        return clock.bracket(split_tag.get(parts, "conv_mfma_kernel"), flops(self, B),
                             lambda: orig_dgrad(self, gout, wb, B, parts))

    def wgrad(self, x, stats, relu_in, gout, B, parts=0):
        tag = "conv_stem_wgrad_kernel" if self.cin == 1 else wgrad_tag.get(parts, "conv_wgrad_kernel")
        return clock.bracket(tag, flops(self, B), lambda: orig_wgrad(self, x, stats, relu_in, gout, B, parts))

    def fwd_stats(self, x, wf, bias, B):
        return clock.bracket("conv_stem_kernel", flops(self, B), lambda: orig_fwd_stats(self, x, wf, bias, B))

    L.fwd, L.dgrad, L.wgrad, L.fwd_stats = fwd, dgrad, wgrad, fwd_stats

    orig_call = ops.call

    def call(fname, *args):
        if fname in ABI_TAGS:
            tag, work = ABI_TAGS[fname]
            if callable(tag):
                tag = tag(args)
            return clock.bracket(tag, work(args), lambda: orig_call(fname, *args))
        return orig_call(fname, *args)

    patched = []
    for mod in (ops, M.trainer):
        if getattr(mod, "call", None) is orig_call:
            mod.call = call
            patched.append(mod)

    orig_gemm = ops._library_gemm

    def gemm(a, b):
        return clock.bracket("library_sgemm(dense layers)", 2.0 * a.shape[0] * a.shape[1] * b.shape[1],
                             lambda: orig_gemm(a, b))

    ops._library_gemm = gemm
    orig_fused = ops._library_fused_linear

    def fused(x, w_t, bias, relu):
        return clock.bracket("library_sgemm(dense layers)", 2.0 * x.shape[0] * x.shape[1] * w_t.shape[1],
                             lambda: orig_fused(x, w_t, bias, relu))

    ops._library_fused_linear = fused

    def restore():
        L.fwd, L.dgrad, L.wgrad, L.fwd_stats = orig_fwd, orig_dgrad, orig_wgrad, orig_fwd_stats
        for mod in patched:
            mod.call = orig_call
        ops._library_gemm = orig_gemm
        ops._library_fused_linear = orig_fused
    return restore


def default_batch(model_name):
    return 16384 if model_name in ("mm_gate", "mdn_gate") else 1024


def kernel_table(clock, elapsed):
    """Per tagged kernel: achieved rate from the HIP-event brackets of the instrumented timed pass
    (`elapsed` = that pass's wall time)."""
    kernels = {}
    for tag, r in clock.summary().items():
        per_launch = r["seconds"] / max(r["launches"], 1)
        hbm = tag in HBM_TAGS
        rate = r["work"] / r["seconds"] / (1e9 if hbm else 1e12)
        peak = PEAK_HBM_GBPS if hbm else (PEAK_BF16_MATRIX_TFLOPS if tag in BF16_TAGS else PEAK_FP32_MATRIX_TFLOPS)
        if tag in SPLIT_PRODUCTS:
            peak = PEAK_BF16_MATRIX_TFLOPS / SPLIT_PRODUCTS[tag]
        kernels[tag] = dict(bound="hbm" if hbm else "mfma", achieved=rate, peak=peak, unit="GB/s" if hbm else "TFLOP/s",
                            frac=rate / peak, launches=r["launches"], avg_ms=per_launch * 1e3,
                            share_of_step=r["seconds"] / elapsed,
                            traffic=(TRAFFIC_RATIO[tag] * r["work"] / max(r["launches"], 1)) if tag in TRAFFIC_RATIO
                            else TRAFFIC_BYTES.get((tag, r["work"] / max(r["launches"], 1))))
    return kernels


def mixture_stream_rates(batch, nodes, device, reps=5):
    """The streaming form of the mixture likelihood (component means given as [B,K,N*N] matrices:
    what a caller with arbitrary, not rank-one, component means runs), timed on its own at the
    workload's shape: forward reads the matrices and x once, backward reads them and writes the
    gradient.  The training step above does not run these kernels - it uses the rank-one form that
    never materialises the matrices - so their HBM rates are reported from this separate pass."""
    K, D, L = 5, nodes * nodes, 68
    g = torch.Generator(device=device).manual_seed(5)
    comp = torch.rand(batch, K, D, device=device, generator=g).requires_grad_(True)
    x = torch.rand(batch, D, device=device, generator=g)
    logits = torch.randn(batch, K, device=device, generator=g)
    ls = torch.full((K,), 1.0, device=device)
    mu, lv = (torch.randn(batch, L, device=device, generator=g) * 0.1 for _ in range(2))
    clock = KernelClock()
    restore = install_clock(clock)
    try:
        for i in range(reps + 1):
            comp.grad = None
            ops.mdn_loss(comp, x, logits, ls, mu, lv).backward()
            if i == 0:
                torch.cuda.synchronize()
                clock.records.clear()
        torch.cuda.synchronize()
    finally:
        restore()
    total = sum(r["seconds"] for r in clock.summary().values())
    table = kernel_table(clock, total)
    return {k: {f: v[f] for f in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_ms")}
            for k, v in table.items()}


def fused_loss_rates(nodes, device, batches=(16384, 1024), reps=20):
    """mdnvae_vae_loss_fwd / _bwd / _fwdbwd (the reference's own loss_function, function/VAE_CNN_NN.py:197-200 and
    function/MM_Gate.py:179-182, and its gradient) for the three reconstruction modes, each timed as `reps`
    back-to-back calls between one pair of HIP events on the launch stream.  Algorithmic bytes per call: forward
    reads recon and x [B,N*N] and the latent rows; backward and the fused pass also write grad_recon and the two
    latent gradients.  The calls rotate over enough buffer sets (about 1 GB in all) that no call finds its
    operands in the 256 MB memory-side cache."""
    from mdnvae_amd._lib import call, ptr, stream
    out = {}
    D, L = nodes * nodes, 68
    for B in batches:
        per_set = 4.0 * B * (3 * D + 4 * L)
        nsets = max(1, min(16, int(1.0e9 // per_set)))
        g = torch.Generator(device=device).manual_seed(B)
        sets = []
        for _ in range(nsets):
            sets.append(dict(recon=torch.rand(B, D, device=device, generator=g) + 0.1, x=torch.rand(B, D, device=device, generator=g),
                             mu=torch.randn(B, L, device=device, generator=g) * 0.1,
                             lv=torch.randn(B, L, device=device, generator=g) * 0.1,
                             gr=torch.empty(B, D, device=device), gm=torch.empty(B, L, device=device),
                             gl=torch.empty(B, L, device=device)))
        ws = torch.empty(int(M._lib.load().mdnvae_loss_workspace_bytes()), dtype=torch.uint8, device=device)
        loss = torch.empty(3, device=device)
        entry = {}
        for mode_name, mode in ops.RECON_MODES.items():
            def fwd(t):
                call("mdnvae_vae_loss_fwd", mode, ptr(t["recon"]), ptr(t["x"]), B, D, ptr(t["mu"]), ptr(t["lv"]), L, None, None,
                     ws.data_ptr(), ptr(loss), stream())

            def bwd(t):
                call("mdnvae_vae_loss_bwd", mode, ptr(t["recon"]), ptr(t["x"]), B, D, ptr(t["mu"]), ptr(t["lv"]), L, None, None,
                     None, ptr(t["gr"]), ptr(t["gm"]), ptr(t["gl"]), stream())

            def fwdbwd(t):
                call("mdnvae_vae_loss_fwdbwd", mode, ptr(t["recon"]), ptr(t["x"]), B, D, ptr(t["mu"]), ptr(t["lv"]), L,
                     ptr(t["gr"]), ptr(t["gm"]), ptr(t["gl"]), ws.data_ptr(), ptr(loss), stream())

            for name, fn, nbytes in (("fwd", fwd, 4.0 * B * (2 * D + 2 * L)), ("bwd", bwd, 4.0 * B * (3 * D + 4 * L)),
                                     ("fwdbwd", fwdbwd, 4.0 * B * (3 * D + 4 * L))):
                for i in range(2):
                    fn(sets[i % nsets])
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(reps):
                    fn(sets[i % nsets])
                b.record()
                torch.cuda.synchronize()
                sec = a.elapsed_time(b) * 1e-3 / reps
                entry[f"{mode_name}_{name}"] = dict(bound="hbm", achieved=nbytes / sec / 1e9, peak=PEAK_HBM_GBPS, unit="GB/s",
                                                    frac=nbytes / sec / 1e9 / PEAK_HBM_GBPS, avg_ms=sec * 1e3,
                                                    algorithmic_bytes=nbytes)
        out[f"B={B}"] = entry
        del sets
    out["note"] = (f"N = {nodes}, L = {L}; {reps} back-to-back calls per figure between one HIP-event pair (each call = the "
                   "streaming kernel + the one-workgroup finalize where a loss is returned); buffer sets rotated past the "
                   "memory-side cache")
    return out


def run_workload(model_name, batch, nodes, steps, warmup, pool_batches, device, rank, world, use_clock,
                 overlap=True, symmetric=True, graph=False, precision="fp32", fast_wgrad=False, arith=None):
    with ops.arithmetic(**(arith or {})):
        return _run_workload(model_name, batch, nodes, steps, warmup, pool_batches, device, rank, world, use_clock,
                             overlap, symmetric, graph, precision, fast_wgrad)


def _run_workload(model_name, batch, nodes, steps, warmup, pool_batches, device, rank, world, use_clock,
                  overlap, symmetric, graph, precision, fast_wgrad):
    """Warm up, then time exactly `steps` steps between barrier + synchronize pairs; returns
    (elapsed seconds: max over ranks, final loss, kernel clock, wall time of the instrumented pass
    or None, workload description)."""
    model, mode, kind = build(model_name, nodes, device, symmetric, precision)
    gen = M.data.synthetic_sc if kind == "sc" else M.data.synthetic_fc
    # each rank owns its shard of the synthetic subjects (seed differs per rank)
    pool = gen(batch * pool_batches, nodes, seed=1234 + rank, device=device)
    extra_inputs = ()
    if model_name in ("cnn_gcn", "cnn_gcn_fc"):
        mask = rank_mask_from(gen(256, nodes, seed=1234, device=device), nodes)
        model.set_mask([mask] * 5)
    if model_name in ("mm_gate", "mdn_gate"):
        extra_inputs = (knn_adjacency_from(gen(256, nodes, seed=1234, device=device), nodes),)
    if model_name == "mdn_gate":
        extra_inputs += (torch.randn(batch, 4, device=device, generator=torch.Generator(device=device).manual_seed(7)),)
    trainer = M.Trainer(model, mode, fast_wgrad=fast_wgrad, **({} if overlap else {"overlap_floats": 0}))
    eps_gen = torch.Generator(device=device).manual_seed(99 + rank)

    if graph:                                    # the whole step recorded once, replayed as one launch
        x0 = pool[:batch]
        trainer.capture((x0,) + extra_inputs, x0, torch.randn(batch, 68, device=device, generator=eps_gen))

    def one_step(i):
        x = pool[(i % pool_batches) * batch:(i % pool_batches + 1) * batch]
        eps = torch.randn(batch, 68, device=device, generator=eps_gen)
        if graph:
            return trainer.replay((x,) + extra_inputs, x, eps)
        return trainer.step((x,) + extra_inputs, x, eps=eps)

    for i in range(warmup):
        one_step(i)

    def timed(nsteps, first):
        if world > 1 or REHEARSE_EXCHANGE:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for i in range(nsteps):
            out = one_step(first + i)
        if world > 1 or REHEARSE_EXCHANGE:
            dist.barrier()
        torch.cuda.synchronize()
        staged = (world > 1 or REHEARSE_EXCHANGE) and dist.get_backend() == "gloo"
        el = torch.tensor([time.perf_counter() - t0], device="cpu" if staged else device, dtype=torch.float64)
        if world > 1 or REHEARSE_EXCHANGE:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), out

    # pass 1: exactly `steps` steps, nothing but the work itself -> the reported throughput
    elapsed, loss = timed(steps, warmup)
    # pass 2: the same steps again with an event pair around every tagged launch -> kernel rates.
    # (Timing events serialise neighbouring kernels; left in pass 1 they cost about 10 % of the step.)
    clock, clocked = KernelClock(), None
    if use_clock:
        restore = install_clock(clock)
        one_step(warmup + steps)                 # first patched calls, event pool
        torch.cuda.synchronize()
        clock.records.clear()
        clocked, _ = timed(steps, warmup + steps + 1)
        restore()
    if hasattr(model, "check_inputs"):
        model.check_inputs()                     # raises if a batch broke the symmetric-input promise
    what = f"synthetic {nodes}x{nodes} {kind.upper()} connectomes, {model_name} VAE ({mode}+KL), fwd+bwd+Adam, DP over subjects"
    return elapsed, float(loss.item()), clock, clocked, what


def headline_dtype(args):
    """The arithmetic of the step's products, as short as it can be said; config["arithmetic"] spells it out."""
    if args.model in ("mm_gate", "mdn_gate"):
        wg = "bf16x3" if args.fast_wgrad else ops.ARITH_WGRAD
        return f"f32 (first-layer products: {ops.ARITH_FWD} forward / {wg} weight gradients on the 16-bit MFMA, f32 accumulate)"
    return {"fp32": "f32 (3x3 convolutions: two fp16 parts per operand, three MFMA products, f32 accumulate)", "fp32_mfma": "f32",
            "fp32_planar": "f32 (3x3 convolutions: bf16x3 / bf16x2 split MFMA, f32 accumulate)",
            "bf16x2": "f32 (3x3 convolutions: bf16x2 split MFMA, f32 accumulate)"}.get(
                args.precision, "bf16 operands, f32 accumulate (3x3 convolutions); f32 elsewhere")


ARITHMETIC_NOTE = (
    "every tensor is f32.  The two big first-layer products run on the 16-bit matrix cores from split operands: "
    "f16x2 = two fp16 parts per operand (22 significand bits, per-tensor power-of-two scale), three products, f32 "
    "accumulate - entry-wise error <= 3 * 2^-22 of sum |a||b|, the level of an f32 dot product; bf16x6 = three bf16 "
    "parts, six products, 2^-24; bf16x3 = two bf16 parts, three products, up to 3 * 2^-16 per product (weight gradients "
    "only, opt-in).  All other products: f32 MFMA or f32 VALU.")


def main():
    global REHEARSE_EXCHANGE
    # the one JSON line is all that may reach stdout: libraries that print there (RCCL's version banner) go to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        line = _main()
    finally:
        os.dup2(json_fd, 1)
        os.close(json_fd)
    if line is not None:
        print(json.dumps(line), flush=True)


def _main():
    global REHEARSE_EXCHANGE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="mdn_gate", choices=list(MODELS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 16384 for the FNN-encoder models, "
                                                          "1024 for the conv-encoder models)")
    ap.add_argument("--also", default="cnn_gcn", help="second workload reported under 'also' on one GPU ('' = none)")
    ap.add_argument("--nodes", type=int, default=87)
    ap.add_argument("--pool", type=int, default=2, help="resident synthetic batches per GPU")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="one GPU only: run the collective calls of the data-parallel path on a one-rank RCCL group")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-script-batch", action="store_true",
                    help="skip the batch-256 runs (eager and graph replay) reported under 'script_batch' on one GPU")
    ap.add_argument("--graph", action="store_true", help="run the main workload's steps as replays of a captured HIP graph "
                                                          "(single GPU)")
    ap.add_argument("--no-n68", action="store_true", help="skip the short 68 x 68 run reported under 'n68' on one GPU")
    ap.add_argument("--no-full-variant", action="store_true",
                    help="skip the short run without the symmetric-input promise reported under 'full_first_layer'")
    ap.add_argument("--no-kernel-clock", action="store_true")
    ap.add_argument("--fast-wgrad", action="store_true",
                    help="Trainer(fast_wgrad=True): first-layer weight gradients from two bf16 parts (3 products, about 1e-5 of "
                         "the gradient's norm) instead of two fp16 parts")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the 'strict' (six bf16 products everywhere) and 'fast_wgrad' runs of the headline workload, the "
                         "'fused_loss_kernels' block and the 'exchange_rehearsal' runs reported on one GPU")
    ap.add_argument("--no-overlap", action="store_true", help="all-reduce every gradient after backward")
    ap.add_argument("--full-first-layer", action="store_true",
                    help="FNN-encoder models: run the first layers on all N*N entries instead of the packed triangle")
    ap.add_argument("--precision", default="fp32", choices=sorted(M.modules.CONV_PRECISIONS),
                    help="conv-encoder models: how the 3x3 convolutions multiply (see ConvVAE)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI).  gloo is a rehearsal path (ranks may share "
                         "a GPU): the trainer stages device buffers through pinned host memory for it, because gloo's "
                         "own handling of device tensors aborted / stalled in this image")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()          # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.rehearse_exchange and world == 1:
        # one GPU: a one-rank RCCL group whose collectives are identities, so that the calls of the
        # multi-GPU path (barriers, early and bucketed gradient all-reduce, MAX of the timings) run
        REHEARSE_EXCHANGE = True
        M.trainer.MIN_RANKS_FOR_EXCHANGE = 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)

    batch = args.batch or default_batch(args.model)
    elapsed, final_loss, clock, clocked, what = run_workload(args.model, batch, args.nodes, args.steps, args.warmup,
                                                             args.pool, device, rank, world,
                                                             not args.no_kernel_clock and not args.graph,
                                                             not args.no_overlap, not args.full_first_layer,
                                                             graph=args.graph and world == 1, precision=args.precision,
                                                             fast_wgrad=args.fast_wgrad)
    if rank == 0:
        line = {
            "metric": "train samples/sec (fwd+bwd, 87x87 connectome)" if args.nodes == 87 else
                      f"train samples/sec (fwd+bwd, {args.nodes}x{args.nodes} connectome)",
            "value": world * batch * args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": headline_dtype(args), "data": "synthetic",
            "config": {"workload": what, "per_gpu_batch": batch, "global_batch": batch * world,
                       "n_nodes": args.nodes, "parallelism": f"dp{world}", "final_loss_per_sample": final_loss / batch,
                       "arithmetic": ARITHMETIC_NOTE,
                       # FNN-encoder models: first layers on the packed upper triangle of the (device-verified)
                       # symmetric input, half the reference's multiply-adds; the like-for-like number without
                       # that shortcut is full_first_layer_value
                       "symmetric_first_layer": bool(args.model in ("mm_gate", "mdn_gate") and not args.full_first_layer),
                       # arithmetic of the two hand-written first-layer products (see ops.ARITH_FWD / ARITH_WGRAD)
                       "first_layer_forward": ops.ARITH_FWD if args.model in ("mm_gate", "mdn_gate") else "library fp32",
                       "first_layer_weight_gradient": ("bf16x3" if args.fast_wgrad else ops.ARITH_WGRAD)
                       if args.model in ("mm_gate", "mdn_gate") else "library fp32"},
        }
        kernels = kernel_table(clock, clocked) if clocked else {}
        if kernels:
            line["kernel_clock"] = {"pass": "second timed pass of the same steps with HIP-event brackets",
                                    "ms_per_step": clocked / args.steps * 1e3}
            # the dominant hand-written kernel of the step (library GEMMs of the small layers are listed
            # under "kernels" like everything else)
            own = [k for k in kernels if not k.startswith("library_")] or list(kernels)
            dominant = max(own, key=lambda k: kernels[k]["share_of_step"])
            line["roofline"] = dict(kernels[dominant], kernel=dominant)
            # the fused mixture-likelihood / KL pass: the kernel the HBM target of this path is stated for
            if "mdn_outer_kernel" in kernels:
                line["roofline_fused_loss"] = dict(kernels["mdn_outer_kernel"], kernel="mdn_outer_sym_kernel")
            line["kernels"] = kernels
        if kernels and args.model == "mdn_gate" and world == 1:
            line["mixture_stream_kernels"] = dict(
                mixture_stream_rates(batch, args.nodes, device),
                note="streaming likelihood over materialised [B,K,N*N] component means, timed on its own; the "
                     "training step uses the rank-one form (mdn_outer_kernel)")
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.model, args.nodes)
    fnn = args.model in ("mm_gate", "mdn_gate")
    if world == 1 and fnn and not args.no_variants and not args.full_first_layer and not args.fast_wgrad and not args.graph:
        # the same workload with every hand-written product at six bf16 terms (2^-24 of sum |a||b|, the finest this
        # library has), and with the faster three-term bf16 weight gradients
        vsteps = max(2, args.steps)
        for key, kw, note in (
                ("strict", dict(arith=dict(fwd="bf16x6", wgrad="bf16x6")),
                 "first-layer products (forward and weight gradients) from three bf16 parts, six products, f32 accumulate"),
                ("fast_wgrad", dict(fast_wgrad=True),
                 "Trainer(fast_wgrad=True): first-layer weight gradients from two bf16 parts, three products")):
            elv, lossv, _, _, whatv = run_workload(args.model, batch, args.nodes, vsteps, 3, args.pool, device, rank, world,
                                                   False, not args.no_overlap, True, **kw)
            line[key] = {"workload": whatv, "arithmetic": note, "value": batch * vsteps / elv, "unit": "samples/s",
                         "per_gpu_batch": batch, "steps": vsteps, "warmup": 3, "ms_per_step": elv / vsteps * 1e3,
                         "final_loss_per_sample": lossv / batch}
    if world == 1 and not args.no_variants:
        line["fused_loss_kernels"] = fused_loss_rates(args.nodes, device)
    if world == 1 and args.nodes != 68 and not args.no_n68:
        # the reference-parity shape (68 x 68 crop) next to the 87 x 87 headline shape
        steps68 = max(2, args.steps)             # the same step count as the headline: a 4-step window read 15 % low
        el68, loss68, _, _, what68 = run_workload(args.model, batch, 68, steps68, 3, args.pool, device, rank, world, False,
                                                  not args.no_overlap, not args.full_first_layer, precision=args.precision)
        line["n68"] = {"workload": what68, "value": batch * steps68 / el68, "unit": "samples/s", "per_gpu_batch": batch,
                       "steps": steps68, "warmup": 3, "ms_per_step": el68 / steps68 * 1e3,
                       "final_loss_per_sample": loss68 / batch}
    if world == 1 and not args.no_full_variant and not args.full_first_layer and args.model in ("mm_gate", "mdn_gate"):
        # the same workload without the promise of symmetric inputs: first layers over all N*N entries,
        # likelihood pass with row and column sums (what a model built with symmetric_input=False runs)
        fsteps = max(2, args.steps)
        elf, lossf, _, _, whatf = run_workload(args.model, batch, args.nodes, fsteps, 3, args.pool, device, rank, world,
                                               False, not args.no_overlap, False)
        line["config"]["full_first_layer_value"] = batch * fsteps / elf
        line["full_first_layer"] = {"workload": whatf, "value": batch * fsteps / elf, "unit": "samples/s",
                                    "per_gpu_batch": batch, "steps": fsteps, "warmup": 3,
                                    "ms_per_step": elf / fsteps * 1e3, "final_loss_per_sample": lossf / batch}
    if world == 1 and not args.no_script_batch and args.model in ("mm_gate", "mdn_gate"):
        # the scripts' own batch size (function/VAE_CNN_NN.py:53): launch-bound when issued call by
        # call, so the step is also timed as one replayed HIP graph
        sb, ssteps = 256, 100
        entry = {"per_gpu_batch": sb, "steps": ssteps, "warmup": 10, "unit": "samples/s"}
        for key, use_graph in (("eager", False), ("graph", True)):
            el, ls, _, _, _ = run_workload(args.model, sb, args.nodes, ssteps, 10, args.pool, device, rank, world, False,
                                           not args.no_overlap, not args.full_first_layer, graph=use_graph)
            entry[key] = {"value": sb * ssteps / el, "ms_per_step": el / ssteps * 1e3, "final_loss_per_sample": ls / sb}
        line["script_batch"] = entry
    if world == 1 and fnn and not args.no_variants and not args.rehearse_exchange and not args.graph:
        # What the gradient exchange costs on the host side, as far as one GPU can show it: the same steps with the
        # collective calls of the data-parallel path issued on a one-rank RCCL group (every sum is an identity), next to
        # the same steps without; and the scripts' batch size replayed as two graphs split at the exchange.
        try:
            rsteps = max(2, args.steps)
            el0, _, _, _, _ = run_workload(args.model, batch, args.nodes, rsteps, 3, args.pool, device, rank, world, False,
                                           not args.no_overlap, not args.full_first_layer)
            REHEARSE_EXCHANGE = True
            M.trainer.MIN_RANKS_FOR_EXCHANGE = 1
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=device)
            try:
                el1, _, _, _, _ = run_workload(args.model, batch, args.nodes, rsteps, 3, args.pool, device, rank, world, False,
                                               not args.no_overlap, not args.full_first_layer)
                elg, _, _, _, _ = run_workload(args.model, 256, args.nodes, 100, 10, args.pool, device, rank, world, False,
                                               not args.no_overlap, not args.full_first_layer, graph=True)
            finally:
                dist.destroy_process_group()
                REHEARSE_EXCHANGE = False
                M.trainer.MIN_RANKS_FOR_EXCHANGE = 2
            line["exchange_rehearsal"] = {
                "what": "one-rank RCCL group on this GPU: early all-reduce of the folded first-layer gradients from inside "
                        "backward, bucketed all-reduce of the rest, barriers; sums are identities, so this is the cost of "
                        "issuing and waiting for the collectives, not of moving bytes over xGMI",
                "per_gpu_batch": batch, "steps": rsteps, "ms_per_step_without": el0 / rsteps * 1e3,
                "ms_per_step_with": el1 / rsteps * 1e3, "overhead_ms_per_step": (el1 - el0) / rsteps * 1e3,
                "script_batch_split_graph": {"per_gpu_batch": 256, "steps": 100, "ms_per_step": elg / 100 * 1e3,
                                             "value": 256 * 100 / elg, "unit": "samples/s",
                                             "what": "captured step replayed as backward graph -> all-reduce -> Adam graph"}}
        except Exception as e:      # no RCCL on this box: the headline does not depend on it
            line["exchange_rehearsal"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if world == 1 and args.also and args.also != args.model:
        ab = default_batch(args.also)
        steps = args.steps
        el2, loss2, clock2, clocked2, what2 = run_workload(args.also, ab, args.nodes, steps, 2, args.pool, device, rank,
                                                           world, not args.no_kernel_clock)
        k2 = kernel_table(clock2, clocked2) if clocked2 else {}
        also = {"workload": what2, "value": ab * steps / el2, "unit": "samples/s", "per_gpu_batch": ab, "steps": steps,
                "warmup": 2, "ms_per_step": el2 / steps * 1e3, "dtype": "f32", "final_loss_per_sample": loss2 / ab,
                "kernels": k2}
        if k2:
            dom2 = max(k2, key=lambda k: k2[k]["share_of_step"])
            also["roofline"] = dict(k2[dom2], kernel=dom2)
            also["elementwise_share_of_step"] = sum(v["share_of_step"] for k, v in k2.items() if k in ELEMENTWISE_TAGS)
        if not args.no_cpu_baseline:
            also["cpu_baseline"] = cpu_baseline(args.also, args.nodes, seconds_budget=12.0)
        line["also"] = also
        # the conv-encoder VAE of function_FC/VAE_CNN_GCN.py with its 3x3 convolutions on the bf16 matrix cores
        el3, loss3, clock3, clocked3, what3 = run_workload("cnn_gcn_fc", ab, args.nodes, steps, 2, args.pool, device, rank,
                                                           world, not args.no_kernel_clock, precision="bf16")
        k3 = kernel_table(clock3, clocked3) if clocked3 else {}
        bf = {"workload": what3, "value": ab * steps / el3, "unit": "samples/s", "per_gpu_batch": ab, "steps": steps,
              "warmup": 2, "ms_per_step": el3 / steps * 1e3, "dtype": "bf16 operands, f32 accumulation (3x3 convolutions); f32 elsewhere",
              "final_loss_per_sample": loss3 / ab, "kernels": k3}
        if k3:
            dom3 = max(k3, key=lambda k: k3[k]["share_of_step"])
            bf["roofline"] = dict(k3[dom3], kernel=dom3)
            bf["elementwise_share_of_step"] = sum(v["share_of_step"] for k, v in k3.items() if k in ELEMENTWISE_TAGS)
        line["also_bf16"] = bf
    if world > 1 or REHEARSE_EXCHANGE:
        dist.destroy_process_group()
    return line if rank == 0 else None


if __name__ == "__main__":
    main()
