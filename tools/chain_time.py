#!/usr/bin/env python3
"""Device time of the chain decoder's forward kernel at a large batch (torch profiler), with and without
the saved layer inputs a backward pass needs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdnvae_amd as M
from mdnvae_amd import ops

DEV = "cuda:0"
B = int(os.environ.get("CB", "16384"))
n_nodes = 87
dec = M.MultiChannelGCNDecoder(R=5, n_nodes=n_nodes, k=32, n_layers=3, latent_dim=68).to(DEV)
adjacency = [[(u + 3 * j) % n_nodes for j in range(32)] for u in range(n_nodes)]
z = torch.randn(B, 68, device=DEV)


def run(grad):
    zz = z.clone().requires_grad_(grad)
    return dec.node_vectors(zz, adjacency)


def time_kernel(grad, iters=20):
    for _ in range(3):
        run(grad)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            run(grad)
        torch.cuda.synchronize()
    for e in prof.key_averages():
        if "chain_fwd" in e.key:
            return e.key[:40], e.device_time_total / e.count
    return None, None


for grad in (True, False):
    k, us = time_kernel(grad)
    print(f"save_rows={grad} {k} {us:.1f} us", flush=True)
