"""FID between two feature sets, computed on the GPU.

Mirrors the reference's validation step: `calculate_fid(mu1, sigma1, mu2, sigma2)`
(function/VAE_CNN_NN.py:310-316, called at 331-337 on flattened networks) and
`calculate_fid(features1, features2)` (validation/FID.py:61-81).

Moments come from one MFMA Gram kernel per feature set (float64 accumulators, so statistics can be
summed across batches and across data-parallel ranks before they are finalised).  The reference's
scipy `sqrtm(sigma1 @ sigma2)` is replaced by the trace identity
    tr sqrtm(S1 S2) = tr sqrtm(S1^(1/2) S2 S1^(1/2)),
valid for symmetric positive semi-definite covariances (and with any factor F F^T = S1 in place of
S1^(1/2)), with both roots taken by a coupled Newton-Schulz iteration in float64 (library GEMMs + the
elementwise kernels of the C ABI); on request (mixed=True) the iteration starts in split-bf16 products
on the matrix cores (sqrt_factor).
"""
import numpy as np
import torch

from . import _lib

_MAX_ITERS = 100
_RIDGE = 1e-14          # relative to the bound of the largest eigenvalue the matrix is scaled by


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


def _trace(a, out):
    _lib.call("mdnvae_trace", _lib.ptr(a, "a", torch.float64), a.shape[0], _lib.ptr(out, "out", torch.float64),
              _lib.stream())
    return float(out[0])


def _symmetrize(a, scale, out):
    _lib.call("mdnvae_symmetrize", _lib.ptr(a, "a", torch.float64), a.shape[0], float(scale),
              _lib.ptr(out, "out", torch.float64), _lib.stream())
    return out


_POWER_STEPS = 10
_POWER_MARGIN = 1.25


def _largest_eigenvalue_estimate(a):
    """Rayleigh quotient of a symmetric PSD matrix after a few power-iteration steps from a fixed, generic start: a
    lower bound of the largest eigenvalue that is within a factor (1/d)^(1/(2 steps)) of it at worst (0.65 at
    d = 4624) and within a per cent on covariances."""
    d = a.shape[0]
    i = torch.arange(d, dtype=torch.float64, device=a.device)
    v = 1.0 + 0.5 * torch.sin(0.7 * i)
    v /= v.norm()
    w = torch.empty_like(v)
    for _ in range(_POWER_STEPS):
        _lib.call("mdnvae_matvec_f64", _lib.ptr(a, "a", torch.float64), _lib.ptr(v, "v", torch.float64), d,
                  _lib.ptr(w, "w", torch.float64), _lib.stream())
        n = w.norm()
        v, w = w / torch.clamp(n, min=1e-300), v
    _lib.call("mdnvae_matvec_f64", _lib.ptr(a, "a", torch.float64), _lib.ptr(v, "v", torch.float64), d,
              _lib.ptr(w, "w", torch.float64), _lib.stream())
    return float(torch.dot(v, w))


def _scaled(a, who, tight=True):
    """(a / s symmetrised with the ridge on its diagonal, s).  s bounds the largest eigenvalue, so that the spectrum
    of a / s lies where the iteration converges (it needs eigenvalues below 2; every factor 2.25 of looseness costs
    an iteration): min(Frobenius norm, infinity norm) are rigorous bounds; tight=True takes 1.25 times a
    power-iteration estimate when that is smaller -- on the covariance of 4624 weakly correlated features the
    infinity norm is 8 times the largest eigenvalue, 2.6 iterations per root.  (None, 0.0) for a zero matrix."""
    d = a.shape[0]
    out3 = torch.empty(3, dtype=torch.float64, device=a.device)
    _lib.call("mdnvae_trace_norm", _lib.ptr(a, "a", torch.float64), d, _lib.ptr(out3, "out", torch.float64), _lib.stream())
    _lib.call("mdnvae_inf_norm", _lib.ptr(a, "a", torch.float64), d, out3[2:].data_ptr(), _lib.stream())
    _, sq, inf = out3.tolist()
    if not (np.isfinite(sq) and np.isfinite(inf)):
        raise FloatingPointError(f"{who}: matrix has non-finite entries")
    s = min(sq ** 0.5, inf)
    if s == 0.0:
        return None, 0.0
    if tight and d >= 64:
        rho = _largest_eigenvalue_estimate(a)
        if np.isfinite(rho) and rho > 0.0:
            s = min(s, _POWER_MARGIN * rho)
    y = _symmetrize(a, 1.0 / s, torch.empty_like(a))
    # ridge: the iteration's Z = Y^-1 is bounded by 1/sqrt(smallest eigenvalue), so on a singular
    # covariance (fewer samples than features, degenerate features) it amplifies rounding noise
    # without limit; 1e-14 keeps Z below 1e7 and moves the trace by at most 1e-7 per null direction
    y.diagonal().add_(_RIDGE)
    return y, s


def _pad16(a, fill=None):
    """a [d,d] as the leading block of a matrix whose size is the next multiple of 16 (what the float64 matrix-core
    kernel takes).  fill=None puts the mean of a's diagonal on the added diagonal: the matrix stays block diagonal,
    so the leading block of its root is the root of a, and the added directions neither slow the iteration down (as
    null directions would) nor change the conditioning.  fill=0.0 for a factor of a product."""
    d = a.shape[0]
    dp = (d + 15) // 16 * 16
    if dp == d:
        return a
    out = torch.zeros(dp, dp, dtype=a.dtype, device=a.device)
    out[:d, :d] = a
    if fill is None:
        fill = float(a.diagonal().sum()) / d
        fill = fill if np.isfinite(fill) and fill > 0.0 else 0.0
    if fill:
        out.diagonal()[d:] = fill
    return out


def sqrtm_psd(a):
    """Principal square root of a symmetric PSD matrix a [d,d] (float64, device).

    Coupled Newton-Schulz: with A scaled by a bound ||A|| of its largest eigenvalue, Y0 = A/||A||, Z0 = I,
    T = (3I - Z Y)/2, Y <- Y T, Z <- T Z; Y -> sqrt(A/||A||).  The trace of Y rises
    monotonically while eigen-directions are still converging; the loop ends when its change
    drops to rounding level (1e-15 relative) or, once below 1e-9 relative, stops shrinking:
    from there on Z keeps growing in the null space of a singular A and would amplify rounding
    noise, while directions with eigenvalue < 1e-18 ||A|| contribute < 1e-9 to the trace.
    Every product runs on this library's float64 matrix-core kernel (a size that is not a multiple of 16 as the
    leading block of a padded matrix, see _pad16).  Returns (sqrt(A), iterations)."""
    d = a.shape[0]
    ap = _pad16(a)
    dp = ap.shape[0]
    for tight in (True, False):
        y, norm = _scaled(ap, "sqrtm_psd", tight)
        if y is None:
            return torch.zeros_like(a), 0
        z = torch.eye(dp, dtype=torch.float64, device=a.device)
        try:
            y, iters = _newton_schulz_f64(y, z, dp)
            break
        except FloatingPointError:
            if not tight:                                 # the rigorous bounds did not help either
                raise
    root = _symmetrize(y, norm ** 0.5, torch.empty_like(ap))
    return (root if dp == d else root[:d, :d].contiguous()), iters


def sqrt_factor(a):
    """A square-root factor of a well-conditioned symmetric PSD matrix a [d,d] (float64, device): F with
    F F^T = a whose trace is close to that of the principal root; about half the time of sqrtm_psd at
    d = 4624.  The FID can work from it (the eigenvalues of S1 S2 are those of F1^T S2 F1, and the last
    step takes a trace); fid_from_stats uses it when called with mixed=True.

    The iteration first runs in fp32-accurate split-bf16 products (a third of the time of a float64 GEMM)
    until fp32 rounding stops it, leaving a symmetric Z~ = a^(-1/2) + E.  Then, in float64, Y0 = a Z~ and
    Z0 = Z~ restart the coupled iteration: W = Z Y = Z a Z^T is symmetric and every later W is a polynomial
    of the first, so it goes to I quadratically whatever E was, and Y = a Z^T ends with Y Y^T = a to
    float64 rounding.  Y = sqrt(a) Q with Q = exp(K) orthogonal, K antisymmetric of the size of
    sqrt(a) E; tr(sqrt(a) K) = 0, so the trace is off in second order only.  E is fp32 rounding amplified
    by the cancellation in Z Y (about sqrt(cond(a)) per step).  Only the part of E that does not commute with a
    matters (an unconverged Z~ is still a polynomial in a), so the fp32 phase hands over as soon as the trace moves
    by less than 1e-3 per step, before the amplified rounding piles up: measured trace error 1e-14 at condition
    20, 2e-8 at condition 1e3 (2e-5 with a hand-over at 2e-6, and three more iterations in total); the fp32
    phase diverges near condition 1e6 (FloatingPointError).  Singular or
    badly conditioned covariances need sqrtm_psd.  Below 512 features this is sqrtm_psd.
    Returns (F, iterations)."""
    d = a.shape[0]
    if d < _FAST_MIN_D:
        return sqrtm_psd(a)
    ap = _pad16(a)
    dp = ap.shape[0]
    a1, norm = _scaled(ap, "sqrt_factor")
    if a1 is None:
        return torch.zeros_like(a), 0
    z, fast_iters = _newton_schulz_f32(a1, dp)
    # every product is a . b^T on this library's kernel, so each iterate is kept with its transpose (the kernel
    # writes both): y = a1 z and yt = z a1 (both factors symmetric), then z and y drift apart from symmetry
    new = lambda: torch.empty_like(a1)
    ys, yts, zs, zts = [new(), new()], [new(), new()], [z, new()], [z.clone(), new()]     # iterate k in [k % 2]
    _dgemm(a1, z, ys[0], False, 0, yts[0])
    w, t = new(), new()
    state = _IterationState(a.device)
    gap = torch.empty(1, dtype=torch.float64, device=a.device)
    iters = 0
    for k in range(_MAX_ITERS):
        i, j = k % 2, (k + 1) % 2
        _dgemm(zs[i], yts[i], w, False, state=state.rec)                   # w = z y
        # t = 1.5 I - 0.25 (w + w^T): symmetric to the bit, so it is its own transpose below
        _lib.call("mdnvae_ns_residual_gap", _lib.ptr(w, "w", torch.float64), dp, _lib.ptr(t, "t", torch.float64),
                  gap.data_ptr(), _lib.stream())
        _dgemm(ys[i], t, ys[j], False, 0, yts[j], state.rec)               # y <- y t
        # the step just taken squares the gap ||I - W|| (times 3/4): from 1e-6 it lands at rounding level, which
        # ends the iteration; a singular a ends it by the rule of sqrtm_psd
        state.step(ys[j], gap)
        _dgemm(t, zts[i], zs[j], False, 0, zts[j], state.rec)              # z <- t z
        if (k + 1) % 2 == 0 or k + 1 == _MAX_ITERS:
            ended, iters = state.look("sqrt_factor")
            if ended:
                break
    y = ys[iters % 2].mul_(norm ** 0.5)
    return (y if dp == d else y[:d, :d].contiguous()), iters + fast_iters


def _sandwich(xt, s):
    """x^T s x for xt = x^T [d,d] and a symmetric s [d,d] on the float64 matrix-core kernel: symmetric by its
    structure, so the second product computes its upper tiles only."""
    d = xt.shape[0]
    xp, sp = _pad16(xt, 0.0), _pad16(_symmetrize(s, 1.0, torch.empty_like(s)), 0.0)
    m = _dgemm(xp, sp, torch.empty_like(sp), False)
    inner = _dgemm(m, xp, torch.empty_like(sp), True)
    return inner if xp.shape[0] == d else inner[:d, :d].contiguous()


def _dgemm(a, b, out, upper, epilogue=0, out_t=None, state=None):
    """out = a b on the float64 matrix instruction (b symmetric; upper: the product is symmetric by its structure and
    only its upper tiles are computed; out_t also receives the transpose); d must be a multiple of 16.  state: the
    record of _IterationState; the product is void once the iteration it belongs to has ended."""
    d = a.shape[0]
    _lib.call("mdnvae_dgemm_sym", _lib.ptr(a, "a", torch.float64), _lib.ptr(b, "b", torch.float64), d, int(upper), int(epilogue),
              _lib.ptr(out, "out", torch.float64), _lib.ptr(out_t, "out_t", torch.float64),
              None if state is None else state[4:].data_ptr(), _lib.stream())
    return out


_QUEUE_AHEAD = 4           # iterations queued between two looks at the device's stopping record


class _IterationState:
    """The stopping rule of the float64 iteration, kept on the device (mdnvae_ns_state_step): the host queues
    _QUEUE_AHEAD iterations at a time and reads the record once per batch; what was queued behind the end of the
    iteration does nothing (the products and the rule check the record when they start)."""

    def __init__(self, device):
        self.rec = torch.zeros(8, dtype=torch.float64, device=device)

    def step(self, y, gap=None):
        _lib.call("mdnvae_ns_state_step", _lib.ptr(y, "y", torch.float64), y.shape[0],
                  None if gap is None else gap.data_ptr(), self.rec.data_ptr(), _lib.stream())

    def look(self, who):
        """(ended, iterations counted so far)"""
        done, iters = self.rec[4:6].tolist()
        if done == 2.0:
            raise FloatingPointError(f"{who}: Newton-Schulz iteration diverged")
        return done != 0.0, int(iters)


def _newton_schulz_f64(y, z, d):
    """Coupled iteration in float64 from the pair (y, z) until the trace of y settles; d a multiple of 16.  The three
    products of a step run on this library's float64 matrix-core kernel, the last two at half the work of a
    general product: with T = 1.5 I - 0.5 Z Y taken from the product as computed (not symmetrised),
    T^T Y = 1.5 Y - 0.5 Y Z Y and T Z = 1.5 Z - 0.5 Z Y Z are symmetric by their structure, whether or not the
    rounded Y and Z still commute -- so only their upper tiles are computed and mirrored, which also makes the
    iterates symmetric to the bit.  (Mirroring Y T with T from a symmetrised Z Y diverges at condition 1e6: that
    product is symmetric only through the commutation.)"""
    state = _IterationState(y.device)
    t, tt = torch.empty_like(y), torch.empty_like(y)
    ys, zs = [y, torch.empty_like(y)], [z, torch.empty_like(y)]      # iterate k lives in ys[k % 2], zs[k % 2]
    iters = 0
    for k in range(_MAX_ITERS):
        y, z, y2, z2 = ys[k % 2], zs[k % 2], ys[(k + 1) % 2], zs[(k + 1) % 2]
        _dgemm(z, y, t, False, 1, tt, state.rec)          # t = 1.5 I - 0.5 z y, and its transpose
        _dgemm(tt, y, y2, True, state=state.rec)
        _dgemm(t, z, z2, True, state=state.rec)
        state.step(y2)
        if (k + 1) % _QUEUE_AHEAD == 0 or k + 1 == _MAX_ITERS:
            ended, iters = state.look("sqrtm_psd")
            if ended:
                break
    return ys[iters % 2], iters


_FAST_MIN_D = 512          # below this a float64 GEMM is too small to be worth avoiding
_FAST_MAX_ITERS = 40
_FAST_CHECK_EVERY = 2
_FAST_STOP = 1e-3          # relative change of the trace at which the fp32 phase hands over


def _newton_schulz_f32(y64, d):
    """The same iteration on padded fp32 copies, every product by mdnvae_split_gemm (three bf16 parts
    per operand, fp32 accuracy on the bf16 matrix cores); ends when the trace of Y has settled to fp32
    level.  Returns Z in float64 and the number of iterations.  The trace is read back every second
    iteration only: an iteration is a few milliseconds of GPU work."""
    dev = y64.device
    ld = (d + 255) // 256 * 256                   # row stride; also the padded M / N of the products
    kp = (d + 15) // 16 * 16
    call, ptr, stream = _lib.call, _lib.ptr, _lib.stream
    nbytes = int(_lib.load().mdnvae_split_operand_bytes(ld, kp))
    ops3 = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(3)]      # split Y, Z, T (pad rows stay zero)
    y = torch.empty(d, ld, dtype=torch.float32, device=dev)
    call("mdnvae_f64_to_f32_padded", ptr(y64, "y", torch.float64), d, ld, ptr(y), stream())
    z = torch.zeros(d, ld, dtype=torch.float32, device=dev)
    z.diagonal().fill_(1.0)
    t, tmp, y2, z2 = (torch.empty(d, ld, dtype=torch.float32, device=dev) for _ in range(4))
    traces = torch.zeros(_FAST_MAX_ITERS + 1, dtype=torch.float64, device=dev)

    def split(src, op):
        call("mdnvae_split_rows", ptr(src), d, kp, ld, 0, op.data_ptr(), stream())

    def product(a_op, b_op, out):                  # out = a . b^T = a . b for the symmetric matrices here
        call("mdnvae_split_gemm", a_op.data_ptr(), b_op.data_ptr(), ld, ld, kp, d, ld, None, None, 0, 0, ptr(out), None, ld, ld,
             256, 6, stream())

    prev, last_delta, iters = None, None, 0
    for iters in range(1, _FAST_MAX_ITERS + 1):
        split(y, ops3[0])
        split(z, ops3[1])
        product(ops3[1], ops3[0], tmp)
        call("mdnvae_ns_residual_f32", ptr(tmp), d, ld, ptr(t), stream())
        split(t, ops3[2])
        product(ops3[0], ops3[2], tmp)
        call("mdnvae_symmetrize_f32", ptr(tmp), d, ld, 1.0, ptr(y2), stream())
        product(ops3[2], ops3[1], tmp)
        call("mdnvae_symmetrize_f32", ptr(tmp), d, ld, 1.0, ptr(z2), stream())
        y, y2, z, z2 = y2, y, z2, z
        call("mdnvae_trace_f32", ptr(y), d, ld, traces[iters:].data_ptr(), stream())
        if iters % _FAST_CHECK_EVERY == 0:
            trs = traces[iters - _FAST_CHECK_EVERY + 1: iters + 1].tolist()
            stop = False
            for tr in trs:
                if not np.isfinite(tr):
                    raise FloatingPointError("sqrtm_psd: Newton-Schulz iteration diverged")
                if prev is not None:
                    delta = abs(tr - prev)
                    if delta <= _FAST_STOP * abs(tr) or (last_delta is not None and delta >= last_delta and delta < 1e-4 * abs(tr)):
                        stop = True
                    last_delta = delta
                prev = tr
            if stop:
                break
    z64 = torch.empty(d, d, dtype=torch.float64, device=dev)
    call("mdnvae_f32_padded_to_f64", ptr(z), d, ld, ptr(z64, "z", torch.float64), stream())
    return z64, iters


def fid_from_stats(mu1, sigma1, mu2, sigma2, device="cuda", mixed=False):
    """||mu1-mu2||^2 + tr(sigma1 + sigma2 - 2 sqrtm(sigma1 sigma2)); inputs numpy or tensors.
    mixed=True takes the two roots by sqrt_factor (well-conditioned covariances only, see there)."""
    mu1, mu2 = _dev_f64(mu1, device).reshape(-1), _dev_f64(mu2, device).reshape(-1)
    s1, s2 = _dev_f64(sigma1, device), _dev_f64(sigma2, device)
    d = mu1.numel()
    if s1.shape != (d, d) or s2.shape != (d, d) or mu2.numel() != d:
        raise ValueError("fid_from_stats: mean/covariance shapes disagree")
    root_of = sqrt_factor if mixed else sqrtm_psd
    f1, _ = root_of(s1)                          # F1 F1^T = S1: eig(S1 S2) = eig(F1^T S2 F1)
    # the eigenvalues of S1 S2 are those of F1^T S2 F1; the principal root is its own transpose
    inner = _sandwich(f1.T.contiguous() if mixed else f1, s2)
    root, _ = root_of(inner)
    out2 = torch.empty(2, dtype=torch.float64, device=s1.device)
    tr1, tr2, trc = (_trace(m, out2) for m in (s1, s2, root))
    diff = (mu1 - mu2).tolist()
    return float(sum(v * v for v in diff) + tr1 + tr2 - 2.0 * trc)


def calculate_fid(*args, device="cuda", group=None, mixed=False):
    """Both reference call forms: calculate_fid(mu1, sigma1, mu2, sigma2) and
    calculate_fid(features1, features2) with feature rows [n,d]."""
    if len(args) == 4:
        return fid_from_stats(*args, device=device, mixed=mixed)
    if len(args) != 2:
        raise TypeError("calculate_fid takes (mu1, sigma1, mu2, sigma2) or (features1, features2)")
    stats = []
    for f in args:
        if not isinstance(f, torch.Tensor):
            f = torch.as_tensor(np.asarray(f, dtype=np.float32))
        stats.append(feature_stats(f.to(device), group))
    (m1, c1), (m2, c2) = stats
    return fid_from_stats(m1, c1, m2, c2, device=device, mixed=mixed)
