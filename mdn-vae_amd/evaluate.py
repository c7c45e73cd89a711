"""Evaluation loop of the training scripts, on the device.

reference: function/VAE_CNN_NN.py:239-289 — `model.eval()`, reconstruction of the test set in
batches of 64 under `no_grad`, division of real and generated matrices by the maximum of the
real ones, clamp of the generated ones at zero, then per-matrix MSE / PSNR.  The reference never
draws from the prior; `sample_prior` adds that (decode of z ~ N(0, I))."""
import math

import torch

from . import _lib


@torch.no_grad()
def reconstruct(model, data, batch_size=64, extra=(), eps=None):
    """Reconstructions [n, D] of `data` [n, D] (function/VAE_CNN_NN.py:242-260).

    `extra` are further positional arguments of model.forward (the adjacency list of the GCN
    model, function/MM_Gate.py:201).  `eps` [n, L] replaces the draws of reparameterize."""
    was_training = model.training
    model.eval()
    try:
        out = torch.empty_like(data)
        for lo in range(0, data.shape[0], batch_size):
            hi = min(lo + batch_size, data.shape[0])
            kw = {} if eps is None else {"eps": eps[lo:hi].contiguous()}
            res = model(data[lo:hi].contiguous(), *extra, **kw)
            out[lo:hi] = res[0]
    finally:
        model.train(was_training)
    return out


@torch.no_grad()
def sample_prior(model, n, extra=(), generator=None, batch_size=64):
    """decode(z) for z ~ N(0, I): [n, D] generated connectomes (no counterpart in the reference)."""
    latent = model.fc_mean.out_features if hasattr(model, "fc_mean") else model.fc21.out_features
    dev = next(model.parameters()).device
    was_training = model.training
    model.eval()
    try:
        out = []
        for lo in range(0, n, batch_size):
            z = torch.randn(min(batch_size, n - lo), latent, device=dev, generator=generator)
            res = model.decode(z, *extra)
            out.append(res[0] if isinstance(res, tuple) else res)
    finally:
        model.train(was_training)
    return torch.cat(out)


def _max(x):
    ws = torch.empty(_lib.workspace_bytes("mdnvae_max_workspace_floats"), device=x.device)
    out = torch.empty(1, device=x.device)
    _lib.call("mdnvae_max", _lib.ptr(x, "x"), x.numel(), _lib.ptr(ws, "workspace"), _lib.ptr(out, "out"),
              _lib.stream())
    return out


def _div_clamp(x, denom, clamp):
    out = torch.empty_like(x)
    _lib.call("mdnvae_div_clamp", _lib.ptr(x, "x"), _lib.ptr(denom, "denom"), int(clamp), 0.0, x.numel(),
              _lib.ptr(out, "out"), _lib.stream())
    return out


@torch.no_grad()
def normalise_for_metrics(real, generated):
    """(real / max(real), max(generated / max(real), 0)) — function/VAE_CNN_NN.py:268-271."""
    real, generated = real.detach().contiguous(), generated.detach().contiguous()
    m = _max(real)
    return _div_clamp(real, m, False), _div_clamp(generated, m, True)


@torch.no_grad()
def matrix_mse(real, generated):
    """Per-matrix mean squared error [n] (the mse_values of function/VAE_CNN_NN.py:274,282);
    inputs [n, D] or [n, N, N]."""
    n = real.shape[0]
    a = real.detach().reshape(n, -1).contiguous()
    b = generated.detach().reshape(n, -1).contiguous()
    if a.shape != b.shape:
        raise ValueError(f"real {tuple(a.shape)} and generated {tuple(b.shape)} differ in shape")
    out = torch.empty(n, device=a.device)
    _lib.call("mdnvae_row_mse", _lib.ptr(a, "real"), _lib.ptr(b, "generated"), n, a.shape[1], _lib.ptr(out, "out"),
              _lib.stream())
    return out


def compute_mse(real, generated):
    """Average of the per-matrix MSE (function/VAE_CNN_NN.py:273-276)."""
    v = matrix_mse(real, generated).tolist()
    return sum(v) / len(v)


def compute_psnr(real, generated, max_value=1.0):
    """Average PSNR over the matrices with non-zero error (function/VAE_CNN_NN.py:281-286)."""
    v = [20 * math.log10(max_value) - 10 * math.log10(m) for m in matrix_mse(real, generated).tolist() if m > 0]
    return sum(v) / len(v) if v else float("nan")
