#!/usr/bin/env python3
"""Micro-benchmarks of single kernels (algorithmic bytes or flops / HIP-event time)."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdnvae_amd as M  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def bench_loss(B, N, L=68):
    D = N * N
    dev = "cuda:0"
    r = torch.randn(B, D, device=dev) * 0.3
    x = torch.rand(B, D, device=dev)
    mu = torch.randn(B, L, device=dev)
    lv = torch.randn(B, L, device=dev) * 0.3
    out = []
    for mode in ("poisson_log", "poisson_rate", "mse"):
        rr = r.abs() + 0.1 if mode == "poisson_rate" else r
        med, best = timeit(lambda: M.vae_loss_terms(rr, x, mu, lv, mode))
        fb = 4 * B * (2 * D + 2 * L)
        out.append(dict(kernel="loss_fwd", mode=mode, B=B, N=N, ms=med * 1e3, GBps=fb / med / 1e9, best_GBps=fb / best / 1e9))
        med, best = timeit(lambda: M.vae_loss_and_grads(rr, x, mu, lv, mode))
        bb = 4 * B * (3 * D + 4 * L)
        out.append(dict(kernel="loss_fwdbwd", mode=mode, B=B, N=N, ms=med * 1e3, GBps=bb / med / 1e9, best_GBps=bb / best / 1e9))
    return out


def bench_conv(B):
    from mdnvae_amd import ops
    dev = "cuda:0"
    geoms = [(1, 48, 7, 1, 3, True, 87), (48, 96, 3, 2, 1, False, 87), (96, 192, 3, 2, 1, False, 44),
             (192, 192, 3, 1, 1, True, 22), (192, 192, 3, 1, 1, True, 17)]
    if os.environ.get("KB_RES_ONLY"):
        geoms = geoms[3:4]
    if os.environ.get("KB_GEOMS"):              # "cin,cout,ks,stride,pad,reflect,h;..."
        geoms = [tuple(int(v) for v in g.split(",")) for g in os.environ["KB_GEOMS"].split(";")]
        geoms = [(a, b, c, d, e, bool(f), h) for a, b, c, d, e, f, h in geoms]
    plain = not os.environ.get("KB_NORM")       # the encoder feeds materialised activations (no norm on load)
    parts = int(os.environ.get("KB_PARTS", "0"))   # 0 = fp32 matrix instruction, 1..3 = bf16 parts
    out = []
    for cin, cout, ks, stride, pad, reflect, h in geoms:
        ly = ops._ConvLayer(cin, cout, ks, stride, pad, reflect, h)
        x = torch.randn(B, cin, h, h, device=dev)
        w = torch.randn(cout, cin, ks, ks, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        wf, wb = ly.pack(w, dev, True, parts)
        st = ops._plane_stats(x) if (cin > 1 and not plain) else None
        g = torch.randn(B, cout, ly.oh, ly.oh, device=dev)
        fl = 2.0 * ks * ks * cin * cout * ly.oh * ly.oh * B
        med, best = timeit(lambda: ly.fwd(x, st, cin > 1 and not plain, wf, b, B, parts), iters=10)
        out.append(dict(kernel="conv_fwd", parts=parts, norm=not plain, geom=[cin, cout, ks, stride, h], B=B, ms=med * 1e3, TFLOPs=fl / med / 1e12))
        if cin > 1:
            med, best = timeit(lambda: ly.dgrad(g, wb, B, parts), iters=10)
            out.append(dict(kernel="conv_dgrad", geom=[cin, cout, ks, stride, h], B=B, ms=med * 1e3, TFLOPs=fl / med / 1e12))
        med, best = timeit(lambda: ly.wgrad(x, st, cin > 1 and not plain, g, B, parts), iters=10)
        out.append(dict(kernel="conv_wgrad", geom=[cin, cout, ks, stride, h], B=B, ms=med * 1e3, TFLOPs=fl / med / 1e12))
    return out


def bench_metrics(count, n=68):
    """Validation-side kernels: matrices (or pairs) per second."""
    dev = "cuda:0"
    a = torch.rand(count, n, n, device=dev)
    a = (a + a.transpose(1, 2)).contiguous()
    b = (a + 0.05 * torch.rand_like(a)).contiguous()
    E = M.evaluate
    out = []
    for name, fn in (("spectrum_top10", lambda: E.spectral_features(a, 10, symmetry_tol=None)),
                     ("graph_features(hops)", lambda: E.graph_features(a, False)),
                     ("graph_features(weighted)", lambda: E.graph_features(a, True)),
                     ("ssim", lambda: E.matrix_ssim(a, b)), ("matrix_mse", lambda: E.matrix_mse(a, b))):
        med, best = timeit(fn, iters=5, warmup=1)
        out.append(dict(kernel=name, count=count, n=n, ms=med * 1e3, matrices_per_s=count / med))
    raw = torch.rand(count, 87, 87, device=dev)
    med, best = timeit(lambda: M.data.prepare_sc(raw), iters=5, warmup=1)
    out.append(dict(kernel="prep_connectome", count=count, n=87, ms=med * 1e3, matrices_per_s=count / med,
                    GBps=count * (87 * 87 + 68 * 68) * 4 / med / 1e9))
    x = torch.randn(count, 30, device=dev)
    med, best = timeit(lambda: M.feature_stats(x), iters=5, warmup=1)
    out.append(dict(kernel="fid_moments_d30", count=count, ms=med * 1e3, rows_per_s=count / med))
    return out


def bench_fid(n, d):
    """FID at scale: moments of two [n,d] feature sets (MFMA Gram) and the two matrix square roots."""
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(n, d, device=dev, generator=g)
    b = torch.randn(n, d, device=dev, generator=g) * 1.1 + 0.05
    out = []
    med, _ = timeit(lambda: M.feature_stats(a), iters=3, warmup=1)
    out.append(dict(kernel="fid_moments", n=n, d=d, ms=med * 1e3, TFLOPs=2.0 * n * d * d / med / 1e12,
                    note="algorithmic 2*n*d^2 (the kernel computes only tiles on/above the diagonal)"))
    m1, c1 = M.feature_stats(a)
    m2, c2 = M.feature_stats(b)
    import time
    for tag, mixed in (("first call", False), ("second call", False), ("mixed, first call", True), ("mixed, second call", True)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        val = M.fid_from_stats(m1, c1, m2, c2, mixed=mixed)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.append(dict(kernel="fid_from_stats (2 Newton-Schulz square roots), " + tag, d=d, s=dt, fid=val))
    t0 = time.perf_counter()
    root, iters = M.sqrtm_psd(c1)
    torch.cuda.synchronize()
    out.append(dict(kernel="sqrtm_psd", d=d, s=time.perf_counter() - t0, iterations=iters,
                    residual=float((root @ root - c1).norm() / c1.norm())))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="loss")
    ap.add_argument("--B", type=int, default=8192)
    ap.add_argument("--N", type=int, default=87)
    a = ap.parse_args()
    if a.what == "loss":
        for row in bench_loss(a.B, a.N):
            print(json.dumps(row))
    if a.what == "fid":
        for row in bench_fid(a.B, a.N):
            print(json.dumps(row))
    if a.what == "metrics":
        for row in bench_metrics(a.B):
            print(json.dumps(row))
    if a.what == "conv":
        for row in bench_conv(a.B):
            print(json.dumps(row))
