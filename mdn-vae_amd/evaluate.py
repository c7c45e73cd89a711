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


@torch.no_grad()
def spectral_features(matrices, k=None, symmetry_tol=1e-4):
    """The k largest eigenvalues of every matrix, descending: compute_spectral_features of
    function/VAE_CNN_NN.py:302-308, batched on the device (Jacobi rotations in LDS).

    matrices [count, n, n] or [count, n*n] must be symmetric (the scripts' real connectomes and the
    outputs of the symmetrising decoders are); an asymmetric batch is refused rather than silently
    given the spectrum of its symmetric part."""
    m = matrices.detach()
    if m.dim() == 2:
        n = int(round(m.shape[1] ** 0.5))
        if n * n != m.shape[1]:
            raise ValueError(f"rows of {m.shape[1]} entries are not square matrices")
        m = m.reshape(m.shape[0], n, n)
    m = m.contiguous()
    n = m.shape[1]
    if m.shape[0] and symmetry_tol is not None:
        asym = _max((m - m.transpose(1, 2)).abs().contiguous())          # one scalar check per batch
        scale = _max(m.abs().contiguous())
        if float(asym) > symmetry_tol * max(float(scale), 1e-30):
            raise ValueError("spectral_features needs symmetric matrices (max |A - A^T| = "
                             f"{float(asym):.3g} at scale {float(scale):.3g})")
    k = n if k is None else int(k)
    out = torch.empty(m.shape[0], k, device=m.device)
    _lib.call("mdnvae_spectrum_topk", _lib.ptr(m, "matrices"), m.shape[0], n, k, _lib.ptr(out, "out"), _lib.stream())
    return out


@torch.no_grad()
def matrix_ssim(real, generated, data_range=1.0):
    """Mean structural similarity of every (real, generated) pair [count] with the skimage defaults
    the scripts rely on (function/VAE_CNN_NN.py:291-297): 7x7 uniform window, sample covariance."""
    a, b = real.detach(), generated.detach()
    if a.dim() == 2:
        n = int(round(a.shape[1] ** 0.5))
        a, b = a.reshape(-1, n, n), b.reshape(-1, n, n)
    if a.shape != b.shape:
        raise ValueError(f"real {tuple(a.shape)} and generated {tuple(b.shape)} differ in shape")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty(a.shape[0], device=a.device)
    _lib.call("mdnvae_ssim", _lib.ptr(a, "real"), _lib.ptr(b, "generated"), a.shape[0], a.shape[1], float(data_range),
              _lib.ptr(out, "out"), _lib.stream())
    return out


def compute_ssim_average(real, generated):
    """function/VAE_CNN_NN.py:291-297 (data_range 1.0)."""
    v = matrix_ssim(real, generated, 1.0).tolist()
    return sum(v) / len(v)


def spectral_fid(real, generated, k=10, group=None):
    """FID between the top-k spectra of real and generated matrices (function/VAE_CNN_NN.py:318-337)."""
    from . import fid
    return fid.calculate_fid(spectral_features(real, k), spectral_features(generated, k), group=group)


@torch.no_grad()
def graph_features(matrices, weighted_paths=True):
    """[count, 5] graph features {weighted-degree mean, std, weighted clustering mean, std, average
    shortest path length} of every matrix read as a weighted graph: compute_graph_features of
    validation/FID.py:6-58 (weighted_paths=True) or of function/VAE_CNN_NN.py:341-379
    (weighted_paths=False: that script never attaches the edge distances, so paths count hops).
    Weights are expected non-negative (as the scripts' normalised, clamped matrices are)."""
    m = matrices.detach()
    if m.dim() == 2:
        n = int(round(m.shape[1] ** 0.5))
        if n * n != m.shape[1]:
            raise ValueError(f"rows of {m.shape[1]} entries are not square matrices")
        m = m.reshape(m.shape[0], n, n)
    m = m.to(torch.float32).contiguous()
    out = torch.empty(m.shape[0], 5, device=m.device)
    _lib.call("mdnvae_graph_features", _lib.ptr(m, "matrices"), m.shape[0], m.shape[1], int(bool(weighted_paths)),
              _lib.ptr(out, "out"), _lib.stream())
    return out


def graph_fid(real, generated, weighted_paths=False, group=None):
    """calculate_fid_graph of function/VAE_CNN_NN.py:381-396 on the five graph features."""
    from . import fid
    return fid.calculate_fid(graph_features(real, weighted_paths), graph_features(generated, weighted_paths), group=group)


@torch.no_grad()
def extract_latent(model, data, batch_size=256, extra=()):
    """Posterior means [n, L] and log-variances of `data`, batch by batch in eval mode -- the loop of
    tools/extract.py:29-45 (encode only: no reconstruction is kept)."""
    was_training = model.training
    model.eval()
    try:
        mus, lvs = [], []
        for lo in range(0, data.shape[0], batch_size):
            mu, logvar = model.encode(data[lo:lo + batch_size].contiguous(), *extra)
            mus.append(mu)
            lvs.append(logvar)
    finally:
        model.train(was_training)
    return torch.cat(mus), torch.cat(lvs)


def save_latent(mu, prefix):
    """Writes `<prefix>.npy` and `<prefix>-un.mat` ({'mu': [n, L]}) as tools/extract.py:47-51 does."""
    import numpy as np
    import scipy.io
    arr = mu.detach().cpu().numpy()
    np.save(prefix + ".npy", arr)
    scipy.io.savemat(prefix + "-un.mat", mdict={"mu": arr})
    return prefix + ".npy", prefix + "-un.mat"
