"""Synthetic connectomes shaped like the reference's inputs, generated on the device.

SC-like: upper-triangular counts ~ round(Gamma(0.3, scale 240)) (heavy tail, mean ~72 like
data/HCP_samples.mat), then the reference's preparation: t + t^T, diagonal := column means,
optional crop [18:86], / 200  (function/VAE_CNN_NN.py:33-38).
FC-like: correlation matrices of a rank-8 factor model, then (x + x^T with the diagonal rule,
crop) mapped to [0,1] by (x+1)/2  (function_FC/VAE_CNN_GCN.py:35-41).
"""
import torch


def _prepare(t, crop, n_nodes):
    t = t + t.transpose(1, 2)
    col_mean = t.mean(dim=1)                        # np.mean(ith, 0): column means
    idx = torch.arange(t.shape[1], device=t.device)
    t[:, idx, idx] = col_mean
    if crop:
        t = t[:, 18:86, 18:86]
    return t.reshape(t.shape[0], -1)


def synthetic_sc(n, n_nodes=87, seed=1234, device="cuda", offset=200.0):
    """[n, n_nodes^2] fp32 structural-connectome rows; n_nodes 87 (raw size) or 68 (cropped)."""
    g = torch.Generator(device=device).manual_seed(seed)
    raw_n = 87
    shape = torch.full((n, raw_n, raw_n), 0.3, device=device)
    counts = torch.round(torch._standard_gamma(shape, generator=g) * 240.0)
    counts = torch.triu(counts, diagonal=1)
    x = _prepare(counts, n_nodes == 68, n_nodes) / offset
    if n_nodes not in (68, 87):
        raise ValueError("n_nodes must be 87 (raw Desikan size) or 68 (cortical crop)")
    return x.contiguous().float()


def synthetic_fc(n, n_nodes=87, seed=1234, device="cuda"):
    g = torch.Generator(device=device).manual_seed(seed)
    raw_n = 87
    u = torch.randn(n, raw_n, 8, device=device, generator=g)
    c = u @ u.transpose(1, 2) + 0.5 * torch.eye(raw_n, device=device)
    d = torch.rsqrt(torch.diagonal(c, dim1=1, dim2=2))
    c = c * d.unsqueeze(1) * d.unsqueeze(2)
    c = torch.triu(c, diagonal=1) * 0.5              # t + t^T then restores the off-diagonal value
    x = (_prepare(c, n_nodes == 68, n_nodes) + 1.0) / 2.0
    return x.contiguous().float()


def shard_range(n, rank, world):
    """Contiguous block of subject indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
