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
PEAK_HBM_GBPS = 8000.0
HBM_KERNELS = ("loss_stream_kernel", "mdn_fwd_kernel", "mdn_bwd_stream_kernel", "chain_decoder_fwd_kernel",
               "chain_decoder_bwd_kernel")
# fabric-side bytes / algorithmic bytes from the rocprofv3 TCC counters (FETCH_SIZE x2 + WRITE_SIZE,
# separate passes; profiles/r01_c_loss_pmc_traffic.md); kernels without a collected pass report null
TRAFFIC_RATIO = {"loss_stream_kernel": 1.00, "mdn_fwd_kernel": 1.10, "mdn_bwd_stream_kernel": 1.27}

MODELS = {
    # name: (constructor kwargs, loss mode, data kind)
    "cnn_gcn": (dict(decoder="gcn", intercept="diag", final="none"), "poisson_log", "sc"),   # function/VAE-CNN-GCN.py
    "cnn_nn": (dict(decoder="nn", intercept="dense", final="exp"), "poisson_rate", "sc"),    # function/VAE_CNN_NN.py
    "cnn_gcn_fc": (dict(decoder="gcn", intercept="diag", final="none"), "mse", "fc"),        # function_FC/VAE_CNN_GCN.py
    "mm_gate": (None, "poisson_log", "sc"),                                                  # function/MM_Gate.py
    "mdn_gate": (None, "mdn", "sc"),          # MM_Gate with the trait-gated mixture head (build-defined)
}


def build(model_name, n_nodes, device):
    kw, mode, kind = MODELS[model_name]
    torch.manual_seed(11111)
    if model_name == "mm_gate":
        model = M.GCNVAE(R=5, n_nodes=n_nodes, k=32, n_layers=3, latent_dim=68)
    elif model_name == "mdn_gate":
        model = M.MDNGateVAE(R=5, n_nodes=n_nodes, k=32, n_layers=3, latent_dim=68, n_traits=4)
    else:
        model = M.ConvVAE(n_nodes=n_nodes, latent_dim=68, **kw)
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
    x = gen(B, n_nodes, seed=1234, device="cpu")
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
    """HIP-event bracket around every launch of the tagged kernels during the timed region."""

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


def install_clock(clock):
    """Wrap the conv launches and the fused loss so that each is timed by its own event pair."""
    L = ops._ConvLayer
    orig_fwd, orig_dgrad, orig_wgrad = L.fwd, L.dgrad, L.wgrad

    def flops(ly, B):
        return 2.0 * ly.ks * ly.ks * ly.cin * ly.cout * ly.oh * ly.oh * B

    def fwd(self, x, stats, relu_in, wf, bias, B):
        tag = "conv_stem_kernel" if self.cin == 1 else "conv_mfma_kernel"
        return clock.bracket(tag, flops(self, B), lambda: orig_fwd(self, x, stats, relu_in, wf, bias, B))

    def dgrad(self, gout, wb, B):
        return clock.bracket("conv_mfma_kernel", flops(self, B), lambda: orig_dgrad(self, gout, wb, B))

    def wgrad(self, x, stats, relu_in, gout, B):
        tag = "conv_stem_wgrad_kernel" if self.cin == 1 else "conv_wgrad_kernel"
        return clock.bracket(tag, flops(self, B), lambda: orig_wgrad(self, x, stats, relu_in, gout, B))

    L.fwd, L.dgrad, L.wgrad = fwd, dgrad, wgrad
    orig_loss = ops.vae_loss_and_grads

    def loss(recon, x, mu, logvar, mode):
        B, D = recon.shape
        nbytes = 4.0 * B * (3 * D + 4 * mu.shape[1])
        return clock.bracket("loss_stream_kernel", nbytes, lambda: orig_loss(recon, x, mu, logvar, mode))

    ops.vae_loss_and_grads = loss
    # trait-gated mixture loss: forward and backward launches are bracketed separately
    F = ops._MdnLoss
    orig_mf, orig_mb = F.forward, F.backward

    def mdn_fwd(ctx, comp, x, logits, log_sigma, mu, logvar):
        B, K, D = comp.shape
        nbytes = 4.0 * B * ((K + 1) * D + 2 * mu.shape[1] + 3 * K)
        return clock.bracket("mdn_fwd_kernel", nbytes, lambda: orig_mf(ctx, comp, x, logits, log_sigma, mu, logvar))

    def mdn_bwd(ctx, gout):
        B, K, D = ctx.saved_tensors[0].shape
        nbytes = 4.0 * B * ((2 * K + 1) * D + 4 * ctx.saved_tensors[4].shape[1] + 5 * K)
        return clock.bracket("mdn_bwd_stream_kernel", nbytes, lambda: orig_mb(ctx, gout))

    F.forward, F.backward = staticmethod(mdn_fwd), staticmethod(mdn_bwd)
    # chain decoder: what it must move is the [B,(R,)N*N] output (forward) / its gradient (backward)
    C = ops._ChainDecoder
    orig_cf, orig_cb = C.forward, C.backward

    def dec_fwd(ctx, z, spec, icpt_w, icpt_b, drop, *params):
        rows = spec.R if spec.components else 1
        nbytes = 4.0 * z.shape[0] * (rows * spec.N * spec.N + spec.L)
        return clock.bracket("chain_decoder_fwd_kernel", nbytes,
                             lambda: orig_cf(ctx, z, spec, icpt_w, icpt_b, drop, *params))

    def dec_bwd(ctx, gout, gxlat):
        spec = ctx.spec
        rows = spec.R if spec.components else 1
        nbytes = 4.0 * gout.shape[0] * (rows * spec.N * spec.N + spec.R * spec.NL * (ops.DEC_KP + ops.DEC_NP))
        return clock.bracket("chain_decoder_bwd_kernel", nbytes, lambda: orig_cb(ctx, gout, gxlat))

    C.forward, C.backward = staticmethod(dec_fwd), staticmethod(dec_bwd)

    def restore():
        L.fwd, L.dgrad, L.wgrad = orig_fwd, orig_dgrad, orig_wgrad
        ops.vae_loss_and_grads = orig_loss
        F.forward, F.backward = staticmethod(orig_mf), staticmethod(orig_mb)
        C.forward, C.backward = staticmethod(orig_cf), staticmethod(orig_cb)
    return restore


def default_batch(model_name):
    return 8192 if model_name in ("mm_gate", "mdn_gate") else 1024


def kernel_table(clock, elapsed):
    """Per tagged kernel: achieved rate from the HIP-event brackets inside the timed region."""
    kernels = {}
    for tag, r in clock.summary().items():
        per_launch = r["seconds"] / max(r["launches"], 1)
        hbm = tag in HBM_KERNELS
        rate = r["work"] / r["seconds"] / (1e9 if hbm else 1e12)
        peak = PEAK_HBM_GBPS if hbm else PEAK_FP32_MATRIX_TFLOPS
        kernels[tag] = dict(bound="hbm" if hbm else "mfma", achieved=rate, peak=peak, unit="GB/s" if hbm else "TFLOP/s",
                            frac=rate / peak, launches=r["launches"], avg_ms=per_launch * 1e3,
                            share_of_step=r["seconds"] / elapsed,
                            traffic=(TRAFFIC_RATIO[tag] * r["work"] / max(r["launches"], 1)) if tag in TRAFFIC_RATIO
                            else None)
    return kernels


def run_workload(model_name, batch, nodes, steps, warmup, pool_batches, device, rank, world, use_clock,
                 overlap=True):
    """Warm up, then time exactly `steps` steps between barrier + synchronize pairs; returns
    (elapsed seconds: max over ranks, final loss, kernel clock, workload description)."""
    model, mode, kind = build(model_name, nodes, device)
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
    trainer = M.Trainer(model, mode) if overlap else M.Trainer(model, mode, overlap_floats=0)
    eps_gen = torch.Generator(device=device).manual_seed(99 + rank)

    def one_step(i):
        x = pool[(i % pool_batches) * batch:(i % pool_batches + 1) * batch]
        eps = torch.randn(batch, 68, device=device, generator=eps_gen)
        return trainer.step((x,) + extra_inputs, x, eps=eps)

    for i in range(warmup):
        one_step(i)
    clock = KernelClock()
    restore = install_clock(clock) if use_clock else (lambda: None)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = None
    for i in range(steps):
        loss = one_step(warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    restore()
    el = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    what = (f"synthetic {nodes}x{nodes} {kind.upper()} connectomes, {model_name} VAE ({mode} + KL), fwd+bwd+Adam, "
            "data-parallel over subjects")
    return float(el.item()), float(loss.item()), clock, what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="mdn_gate", choices=list(MODELS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 8192 for the FNN-encoder models, "
                                                          "1024 for the conv-encoder models)")
    ap.add_argument("--also", default="cnn_gcn", help="second workload reported under 'also' on one GPU ('' = none)")
    ap.add_argument("--nodes", type=int, default=87)
    ap.add_argument("--pool", type=int, default=2, help="resident synthetic batches per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-clock", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="all-reduce every gradient after backward")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL).  gloo on device tensors "
                                                      "hung in this image when two ranks shared one GPU, so the "
                                                      "multi-rank path is rehearsed on the CPU by tests/test_trainer_cpu.py")
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)

    batch = args.batch or default_batch(args.model)
    elapsed, final_loss, clock, what = run_workload(args.model, batch, args.nodes, args.steps, args.warmup, args.pool,
                                                    device, rank, world, not args.no_kernel_clock,
                                                    not args.no_overlap)
    if rank == 0:
        line = {
            "metric": "train samples/sec (fwd+bwd, 87x87 connectome)" if args.nodes == 87 else
                      f"train samples/sec (fwd+bwd, {args.nodes}x{args.nodes} connectome)",
            "value": world * batch * args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": what, "per_gpu_batch": batch, "global_batch": batch * world,
                       "n_nodes": args.nodes, "parallelism": f"dp{world}", "final_loss_per_sample": final_loss / batch},
        }
        kernels = kernel_table(clock, elapsed)
        if kernels:
            dominant = max(kernels, key=lambda k: kernels[k]["share_of_step"])
            line["roofline"] = dict(kernels[dominant], kernel=dominant)
            line["kernels"] = kernels
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.model, args.nodes)
    if world == 1 and args.also and args.also != args.model:
        ab = default_batch(args.also)
        steps = max(2, min(args.steps, 4))
        el2, loss2, clock2, what2 = run_workload(args.also, ab, args.nodes, steps, 2, args.pool, device, rank, world,
                                                 not args.no_kernel_clock)
        k2 = kernel_table(clock2, el2)
        also = {"workload": what2, "value": ab * steps / el2, "unit": "samples/s", "per_gpu_batch": ab, "steps": steps,
                "warmup": 2, "ms_per_step": el2 / steps * 1e3, "dtype": "f32", "final_loss_per_sample": loss2 / ab,
                "kernels": k2}
        if k2:
            dom2 = max(k2, key=lambda k: k2[k]["share_of_step"])
            also["roofline"] = dict(k2[dom2], kernel=dom2)
        if not args.no_cpu_baseline:
            also["cpu_baseline"] = cpu_baseline(args.also, args.nodes, seconds_budget=12.0)
        line["also"] = also
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
