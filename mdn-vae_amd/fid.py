"""FID between two feature sets, computed on the GPU.

Mirrors the reference's validation step: `calculate_fid(mu1, sigma1, mu2, sigma2)`
(function/VAE_CNN_NN.py:310-316, called at 331-337 on flattened networks) and
`calculate_fid(features1, features2)` (validation/FID.py:61-81).

Moments come from one MFMA Gram kernel per feature set (float64 accumulators, so statistics can be
summed across batches and across data-parallel ranks before they are finalised).  The reference's
scipy `sqrtm(sigma1 @ sigma2)` is replaced by the trace identity
    tr sqrtm(S1 S2) = tr sqrtm(S1^(1/2) S2 S1^(1/2)),
valid for symmetric positive semi-definite covariances, with both square roots taken by a coupled
Newton-Schulz iteration in float64 (library GEMMs + the elementwise kernels of the C ABI).
"""
import numpy as np
import torch

from . import _lib

_MAX_ITERS = 100
_RIDGE = 1e-14          # relative to the Frobenius norm of the matrix


def _dev_f64(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=device).contiguous()


class FeatureStats:
    """Running (n, sum, Gram) of feature rows around a fixed shift vector.

    `update` may be called once per batch; `allreduce` sums the accumulators over a process group
    (every rank must have been created with the same `shift`); `finalize` returns (mean, cov) in
    float64 with np.cov's n-1 normalisation."""

    def __init__(self, dim, shift=None, device="cuda"):
        self.dim = int(dim)
        self.device = torch.device(device)
        if shift is None:
            shift = torch.zeros(self.dim, device=self.device)
        self.shift = shift.to(device=self.device, dtype=torch.float32).contiguous()
        if self.shift.numel() != self.dim:
            raise ValueError(f"shift has {self.shift.numel()} entries, expected {self.dim}")
        self.n = 0
        self.sum = torch.zeros(self.dim, dtype=torch.float64, device=self.device)
        self.gram = torch.zeros(self.dim, self.dim, dtype=torch.float64, device=self.device)
        self._ws = None

    def update(self, x):
        if x.dim() != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected feature rows [n,{self.dim}], got {tuple(x.shape)}")
        x = x.detach().to(torch.float32).contiguous()
        n = x.shape[0]
        need = _lib.workspace_bytes_f("mdnvae_fid_gram_workspace_floats", n, self.dim)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1), device=self.device)
        _lib.call("mdnvae_fid_accumulate", _lib.ptr(x, "x"), _lib.ptr(self.shift, "shift"), n, self.dim,
                  _lib.ptr(self._ws, "workspace"), _lib.ptr(self.sum, "sum", torch.float64),
                  _lib.ptr(self.gram, "gram", torch.float64), _lib.stream())
        self.n += n
        return self

    def allreduce(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            flat = torch.cat([self.sum, self.gram.reshape(-1),
                              torch.tensor([float(self.n)], dtype=torch.float64, device=self.device)])
            from .trainer import all_reduce_sum_
            all_reduce_sum_(flat, group)       # device memory under RCCL, staged through the host under gloo
            self.sum = flat[: self.dim].contiguous()
            self.gram = flat[self.dim: self.dim + self.dim * self.dim].reshape(self.dim, self.dim).contiguous()
            self.n = int(round(float(flat[-1])))
        return self

    def finalize(self):
        if self.n < 2:
            raise ValueError("covariance needs at least two feature rows")
        mean = torch.empty(self.dim, dtype=torch.float64, device=self.device)
        cov = torch.empty(self.dim, self.dim, dtype=torch.float64, device=self.device)
        _lib.call("mdnvae_fid_finalize", _lib.ptr(self.gram, "gram", torch.float64),
                  _lib.ptr(self.sum, "sum", torch.float64), _lib.ptr(self.shift, "shift"), float(self.n), self.dim,
                  _lib.ptr(mean, "mean", torch.float64), _lib.ptr(cov, "cov", torch.float64), _lib.stream())
        return mean, cov


def feature_stats(x, group=None):
    """(mean, cov) of feature rows x [n,d]; the first row serves as the shift (broadcast from rank 0
    when a process group is active, so every rank centres on the same vector)."""
    x = x.detach().to(torch.float32)
    shift = x[0].clone()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(shift, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return FeatureStats(x.shape[1], shift, x.device).update(x).allreduce(group).finalize()


def _trace_norm(a, out2):
    _lib.call("mdnvae_trace_norm", _lib.ptr(a, "a", torch.float64), a.shape[0],
              _lib.ptr(out2, "out2", torch.float64), _lib.stream())
    tr, sq = out2.tolist()
    return tr, sq


def _symmetrize(a, scale, out):
    _lib.call("mdnvae_symmetrize", _lib.ptr(a, "a", torch.float64), a.shape[0], float(scale),
              _lib.ptr(out, "out", torch.float64), _lib.stream())
    return out


def sqrtm_psd(a):
    """Principal square root of a symmetric PSD matrix a [d,d] (float64, device).

    Coupled Newton-Schulz: with A scaled by its Frobenius norm, Y0 = A/||A||, Z0 = I,
    T = (3I - Z Y)/2, Y <- Y T, Z <- T Z; Y -> sqrt(A/||A||).  Iterates are symmetrised every
    step (rounding otherwise feeds the unstable antisymmetric mode).  The trace of Y rises
    monotonically while eigen-directions are still converging; the loop ends when its change
    drops to rounding level (1e-15 relative) or, once below 1e-9 relative, stops shrinking:
    from there on Z keeps growing in the null space of a singular A and would amplify rounding
    noise, while directions with eigenvalue < 1e-18 ||A|| contribute < 1e-9 to the trace.
    Returns (sqrt(A), iterations)."""
    d = a.shape[0]
    out2 = torch.empty(2, dtype=torch.float64, device=a.device)
    _, sq = _trace_norm(a, out2)
    norm = sq ** 0.5
    if not np.isfinite(norm):
        raise FloatingPointError("sqrtm_psd: matrix has non-finite entries")
    if norm == 0.0:
        return torch.zeros_like(a), 0
    y = _symmetrize(a, 1.0 / norm, torch.empty_like(a))
    # ridge: the iteration's Z = Y^-1 is bounded by 1/sqrt(smallest eigenvalue), so on a singular
    # covariance (fewer samples than features, degenerate features) it amplifies rounding noise
    # without limit; 1e-14 keeps Z below 1e7 and moves the trace by at most 1e-7 per null direction
    y.diagonal().add_(_RIDGE)
    z = torch.eye(d, dtype=torch.float64, device=a.device)
    t = torch.empty_like(a)
    tmp = torch.empty_like(a)
    prev, last_delta, iters = None, None, 0
    for iters in range(1, _MAX_ITERS + 1):
        torch.matmul(z, y, out=tmp)
        _lib.call("mdnvae_ns_residual", _lib.ptr(tmp, "zy", torch.float64), d, _lib.ptr(t, "t", torch.float64),
                  _lib.stream())
        torch.matmul(y, t, out=tmp)
        y_new = _symmetrize(tmp, 1.0, torch.empty_like(a))
        torch.matmul(t, z, out=tmp)
        z = _symmetrize(tmp, 1.0, torch.empty_like(a))
        y = y_new
        tr, _ = _trace_norm(y, out2)
        if not np.isfinite(tr):
            raise FloatingPointError("sqrtm_psd: Newton-Schulz iteration diverged")
        if prev is not None:
            delta = abs(tr - prev)
            if delta <= 1e-15 * abs(tr) or (last_delta is not None and delta >= last_delta
                                            and delta < 1e-9 * abs(tr)):
                break
            last_delta = delta
        prev = tr
    return _symmetrize(y, norm ** 0.5, torch.empty_like(a)), iters


def fid_from_stats(mu1, sigma1, mu2, sigma2, device="cuda"):
    """||mu1-mu2||^2 + tr(sigma1 + sigma2 - 2 sqrtm(sigma1 sigma2)); inputs numpy or tensors."""
    mu1, mu2 = _dev_f64(mu1, device).reshape(-1), _dev_f64(mu2, device).reshape(-1)
    s1, s2 = _dev_f64(sigma1, device), _dev_f64(sigma2, device)
    d = mu1.numel()
    if s1.shape != (d, d) or s2.shape != (d, d) or mu2.numel() != d:
        raise ValueError("fid_from_stats: mean/covariance shapes disagree")
    root1, _ = sqrtm_psd(s1)
    inner = torch.matmul(torch.matmul(root1, s2), root1)
    root, _ = sqrtm_psd(inner)
    out2 = torch.empty(2, dtype=torch.float64, device=s1.device)
    tr1, _ = _trace_norm(s1, out2)
    tr2, _ = _trace_norm(s2, out2)
    trc, _ = _trace_norm(root, out2)
    diff = (mu1 - mu2).tolist()
    return float(sum(v * v for v in diff) + tr1 + tr2 - 2.0 * trc)


def calculate_fid(*args, device="cuda", group=None):
    """Both reference call forms: calculate_fid(mu1, sigma1, mu2, sigma2) and
    calculate_fid(features1, features2) with feature rows [n,d]."""
    if len(args) == 4:
        return fid_from_stats(*args, device=device)
    if len(args) != 2:
        raise TypeError("calculate_fid takes (mu1, sigma1, mu2, sigma2) or (features1, features2)")
    stats = []
    for f in args:
        if not isinstance(f, torch.Tensor):
            f = torch.as_tensor(np.asarray(f, dtype=np.float32))
        stats.append(feature_stats(f.to(device), group))
    (m1, c1), (m2, c2) = stats
    return fid_from_stats(m1, c1, m2, c2, device=device)
