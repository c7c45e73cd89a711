"""Autograd wrappers over the C ABI: each op's forward and backward are hand-written HIP
kernels; nothing here computes on the host or falls back to eager PyTorch arithmetic."""
import contextlib

import torch

from . import _lib
from ._lib import call, ptr, stream

RECON_MODES = {"poisson_rate": 0, "poisson_log": 1, "mse": 2}


def _ws(query, device):
    return torch.empty(_lib.workspace_bytes(query), dtype=torch.uint8, device=device)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Reparam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = _c(mu), _c(logvar), _c(eps)
        z = torch.empty_like(mu)
        call("mdnvae_reparam_fwd", ptr(mu, "mu"), ptr(logvar, "logvar"), ptr(eps, "eps"), ptr(z), mu.numel(), stream())
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    def backward(ctx, gz):
        logvar, eps = ctx.saved_tensors
        gz = _c(gz)
        gmu = torch.empty_like(gz)
        glv = torch.empty_like(gz)
        call("mdnvae_reparam_bwd", ptr(logvar), ptr(eps), ptr(gz), ptr(gmu), ptr(glv), gz.numel(), stream())
        return gmu, glv, None


def reparameterize(mu, logvar, eps=None):
    """z = mu + eps*exp(0.5*logvar); eps defaults to a fresh standard-normal draw."""
    if eps is None:
        eps = torch.randn_like(mu)
    return _Reparam.apply(mu, logvar, eps)


class _VaeLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, recon, x, mu, logvar, mode):
        recon, x, mu, logvar = _c(recon), _c(x), _c(mu), _c(logvar)
        if recon.shape != x.shape or recon.dim() != 2:
            raise RuntimeError(f"loss: recon {tuple(recon.shape)} and x {tuple(x.shape)} must be equal [B,D]")
        if mu.shape != logvar.shape or mu.shape[0] != recon.shape[0]:
            raise RuntimeError(f"loss: mu {tuple(mu.shape)} / logvar {tuple(logvar.shape)} do not match batch {recon.shape[0]}")
        B, D = recon.shape
        L = mu.shape[1]
        out = torch.empty(3, dtype=torch.float32, device=recon.device)
        ws = _ws("mdnvae_loss_workspace_bytes", recon.device)
        call("mdnvae_vae_loss_fwd", mode, ptr(recon, "recon"), ptr(x, "x"), B, D, ptr(mu, "mu"), ptr(logvar, "logvar"),
             L, None, None, ws.data_ptr(), ptr(out), stream())
        ctx.save_for_backward(recon, x, mu, logvar)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, gout):
        recon, x, mu, logvar = ctx.saved_tensors
        B, D = recon.shape
        L = mu.shape[1]
        g2 = torch.stack((gout[0] + gout[1], gout[0] + gout[2])).contiguous()
        grecon = torch.empty_like(recon)
        gmu = torch.empty_like(mu)
        glv = torch.empty_like(mu)
        call("mdnvae_vae_loss_bwd", ctx.mode, ptr(recon), ptr(x), B, D, ptr(mu), ptr(logvar), L, None, None,
             ptr(g2), ptr(grecon), ptr(gmu), ptr(glv), stream())
        return grecon, None, gmu, glv, None


def vae_loss_terms(recon, x, mu, logvar, mode):
    """(total, reconstruction term, KL) as 0-dim tensors."""
    out = _VaeLoss.apply(recon, x, mu, logvar, RECON_MODES[mode])
    return out[0], out[1], out[2]


def vae_loss(recon, x, mu, logvar, mode):
    return vae_loss_terms(recon, x, mu, logvar, mode)[0]


def vae_loss_and_grads(recon, x, mu, logvar, mode):
    """One pass over recon/x: returns (loss[3], grad_recon, grad_mu, grad_logvar) for an
    upstream gradient of 1 (used by the fused trainer, which does not go through autograd)."""
    recon, x, mu, logvar = _c(recon), _c(x), _c(mu), _c(logvar)
    B, D = recon.shape
    L = mu.shape[1]
    out = torch.empty(3, dtype=torch.float32, device=recon.device)
    grecon = torch.empty_like(recon)
    gmu = torch.empty_like(mu)
    glv = torch.empty_like(mu)
    ws = _ws("mdnvae_loss_workspace_bytes", recon.device)
    call("mdnvae_vae_loss_fwdbwd", RECON_MODES[mode], ptr(recon, "recon"), ptr(x, "x"), B, D, ptr(mu, "mu"),
         ptr(logvar, "logvar"), L, ptr(grecon), ptr(gmu), ptr(glv), ws.data_ptr(), ptr(out), stream())
    return out, grecon, gmu, glv


# ---------------------------------------------------------------------------------------
# dense layers: library GEMM + HIP bias/activation kernels
# ---------------------------------------------------------------------------------------
ACTS = {None: 0, "none": 0, "sigmoid": 1, "tanh": 2, "relu": 3}


# ---------------------------------------------------------------------------------------
# gradients added straight into .grad
# ---------------------------------------------------------------------------------------
_SINK = None


class accumulate_into_grads:
    """While active, an op whose parameter gradient comes out of one of this library's kernels adds
    it straight into the parameter's existing `.grad` (what autograd's AccumulateGrad would do with
    one more elementwise kernel per parameter - some fifty launches a step for the GCN-decoder
    models) and hands autograd no gradient for that parameter.  `ready(param)`, if given, is called
    after each such addition (post-accumulate hooks do not fire on this path).  Parameters whose
    `.grad` is missing, of another dtype / shape or not contiguous take the ordinary path.
    `folded(param, gt, n_nodes)`, if given, takes over the weight gradient of a symmetric-input
    layer in its folded form gt [T, H] (one value per unordered pair (i,j)): the caller then owns
    writing it to both (i,j) and (j,i) of param.grad, after whatever it does with the half-size
    buffer first (the data-parallel trainer all-reduces it)."""

    def __init__(self, ready=None, folded=None):
        self.ready = ready
        self.folded = folded

    def __enter__(self):
        global _SINK
        self._outer, _SINK = _SINK, self
        return self

    def __exit__(self, *exc):
        global _SINK
        _SINK = self._outer
        return False


def _grad_target(param):
    """The `.grad` tensor the active sink may add into, or None."""
    if _SINK is None or param is None:
        return None
    if not isinstance(param, torch.nn.Parameter) or not param.is_leaf:
        return None          # e.g. weight * mask: autograd carries the gradient to the leaf itself
    g = getattr(param, "grad", None)
    if g is None or g.dtype != torch.float32 or g.shape != param.shape or not g.is_contiguous():
        return None
    return g


def _grad_added(param):
    if _SINK is not None and _SINK.ready is not None:
        _SINK.ready(param)


def _colsum(a, into=None):
    """Column sums of a [B,D]; `into` [D]: add them to it instead of returning a new tensor."""
    B, D = a.shape
    ws = torch.empty(_lib.workspace_bytes_f("mdnvae_colsum_workspace_floats", B, D), dtype=torch.float32, device=a.device)
    out = into if into is not None else torch.empty(D, dtype=torch.float32, device=a.device)
    call("mdnvae_colsum", ptr(a), B, D, ptr(ws), ptr(out), int(into is not None), stream())
    return out


def _bias_grad(gpre, bias_param):
    """Bias gradient of a dense layer: None when it went straight into bias_param.grad."""
    target = _grad_target(bias_param)
    if target is not None:
        _colsum(gpre, into=target)
        _grad_added(bias_param)
        return None
    return _colsum(gpre)


def _library_gemm(a, b):
    """Plain dense product of the nn.Linear layers: a rocBLAS/hipBLASLt GEMM (one call site, so a
    benchmark can time it)."""
    return torch.mm(a, b)


def _library_fused_linear(x, w_t, bias, relu):
    """x @ w_t + bias, with ReLU if asked, as one library GEMM with an epilogue (one call site, so a
    benchmark can time it)."""
    if relu:
        return torch._addmm_activation(bias, x, w_t, use_gelu=False)
    return torch.addmm(bias, x, w_t)


def _library_linear(x, w_t, bias, act):
    """act(x @ w_t + bias) [B,H] for a [K,H] view w_t of the weight.  With a bias and ReLU or no
    activation the library GEMM applies them in its own epilogue (one pass less over the [B,H]
    result); other cases take the plain GEMM and the bias/activation kernel."""
    if bias is not None and x.shape[0] > 0 and (
            act == ACTS[None] or (act == ACTS["relu"] and hasattr(torch, "_addmm_activation"))):
        return _library_fused_linear(x, w_t, bias, act == ACTS["relu"])
    y = _library_gemm(x, w_t)
    call("mdnvae_bias_act", ptr(y), ptr(bias, "bias"), y.shape[0], y.shape[1], act, stream())
    return y


def _small_wgrad_ok(B, n_out, n_in):
    """Layers whose weight is small next to the batch: the split-over-the-batch kernel beats a
    library GEMM that has only a handful of output tiles."""
    return B >= 1024 and n_out % 4 == 0 and n_in % 4 == 0 and n_out * n_in <= (1 << 18)


def _linear_wgrad(gpre, x, into=None):
    """grad_w [N,M] = gpre^T . x for gpre [B,N], x [B,M] (contiguous); `into`: add it there instead."""
    B, N = gpre.shape
    M = x.shape[1]
    ws = torch.empty(_lib.workspace_bytes_f("mdnvae_linear_wgrad_workspace_floats", B, N, M), dtype=torch.float32,
                     device=x.device)
    gw = into if into is not None else torch.empty(N, M, dtype=torch.float32, device=x.device)
    call("mdnvae_linear_wgrad", ptr(gpre, "grad_pre"), ptr(x, "x"), B, N, M, ptr(ws), ptr(gw), int(into is not None),
         stream())
    return gw


def _linear_wgrad_bias(gpre, x, weight_param, bias_param):
    """(grad_w, grad_b) of a small dense layer from one launch pair (the bias gradient rides along in the
    split-over-the-batch weight gradient); None for what went straight into the parameter's .grad."""
    B, N = gpre.shape
    M = x.shape[1]
    ws = torch.empty(_lib.workspace_bytes_f("mdnvae_linear_wgrad_workspace_floats", B, N, M), dtype=torch.float32,
                     device=x.device)
    tw, tb = _grad_target(weight_param), _grad_target(bias_param)
    gw = tw if tw is not None else torch.empty(N, M, dtype=torch.float32, device=x.device)
    gb = tb if tb is not None else torch.empty(N, dtype=torch.float32, device=x.device)
    call("mdnvae_linear_wgrad_bias", ptr(gpre, "grad_pre"), ptr(x, "x"), B, N, M, ptr(ws), ptr(gw), int(tw is not None),
         ptr(gb), int(tb is not None), stream())
    if tw is not None:
        _grad_added(weight_param)
    if tb is not None:
        _grad_added(bias_param)
    return (None if tw is not None else gw), (None if tb is not None else gb)


NARROW_MAX = 8


def _narrow(weight):
    """A layer at most 8 wide on both sides (the mixture gate): one HIP kernel each way instead of
    a library GEMM whose smallest tile dwarfs the weight."""
    return weight.shape[0] <= NARROW_MAX and weight.shape[1] <= NARROW_MAX


def _narrow_wgrad(gpre, x, weight_param, bias_param, want_w, want_b):
    """(gw, gb) of a narrow layer from one pass over the batch; None for what went straight into
    the parameter's .grad (both or neither: the kernel has one accumulate switch)."""
    B, N = gpre.shape
    M = x.shape[1]
    tw = _grad_target(weight_param) if want_w else None
    tb = _grad_target(bias_param) if want_b else None
    direct = (tw is not None) and (tb is not None or not want_b)
    gw = tw if direct else torch.empty(N, M, dtype=torch.float32, device=x.device)
    gb = (tb if direct else torch.empty(N, dtype=torch.float32, device=x.device)) if want_b else None
    ws = torch.empty(_lib.workspace_bytes("mdnvae_narrow_linear_workspace_floats"), dtype=torch.float32, device=x.device)
    call("mdnvae_narrow_linear_wgrad", ptr(gpre, "grad_pre"), ptr(x, "x"), B, N, M, ptr(ws), ptr(gw), ptr(gb),
         int(direct), stream())
    if direct:
        _grad_added(weight_param)
        if want_b:
            _grad_added(bias_param)
        return None, None
    return (gw if want_w else None), gb


SPLIT_LINEAR_MIN_OUTPUTS = 6_000_000     # B * n_out from which the split product has enough 128 x 256 tiles for the chip


def _split_linear_ok(x, weight, bias):
    """Large dense layers (the N*N x N*N intercept of function/VAE_CNN_NN.py:187 at batch 1024, the
    cVAE layers of function_FC_SC/SC-FC.py:95-113 at large batches) run on the hand-written split-bf16
    product: fp32 accuracy, bf16 matrix cores."""
    return (SPLIT_LINEAR and x.dim() == 2 and bias is not None and x.shape[0] >= SPLIT_GEMM_MIN_BATCH
            and weight.shape[1] >= 256 and x.shape[0] * weight.shape[0] >= SPLIT_LINEAR_MIN_OUTPUTS)


SPLIT_LINEAR = True
# Arithmetic of the hand-written split products (results are fp32 tensors in every case):
#   "f16x2"  two fp16 parts per operand, three matrix-core products, each tensor scaled by a power of two taken from
#            its largest magnitude: 22-bit operands, error 3 * 2^-22 of sum |a||b| - the level of an fp32 dot product;
#   "bf16x6" three bf16 parts, six products, error 2^-24 of sum |a||b|, at twice the matrix-core work;
#   "bf16x3" (weight gradients only) two bf16 parts, three products: up to 3 * 2^-16 per product - about 1e-5 of a
#            gradient's norm, 1e-3 of single entries; a little faster than "f16x2" (no magnitude bounds to take).
# ARITH_FWD covers the products whose result feeds further layers (forward, data gradient), ARITH_WGRAD the weight
# and bias gradients.  They are process-wide settings of this module, changed with arithmetic(...) - nothing reads
# the environment - and an autograd node uses in its backward what was set when its forward ran.
ARITH_FWD = "f16x2"
ARITH_WGRAD = "f16x2"
SPLIT_WGRAD_KSPLIT = True            # long-k products with few output tiles: two workgroups per tile


@contextlib.contextmanager
def arithmetic(fwd=None, wgrad=None, factored_dgrad=None):
    """Run the enclosed forward passes (and the backward passes of the graphs they build) with another arithmetic
    for the split products; see ARITH_FWD / ARITH_WGRAD.  factored_dgrad=False: every data gradient is formed and
    handed to autograd (see _FactoredGrad)."""
    global ARITH_FWD, ARITH_WGRAD, FACTORED_DGRAD
    if fwd not in (None, "f16x2", "bf16x6"):
        raise ValueError(f"arithmetic: fwd must be 'f16x2' or 'bf16x6' (got {fwd!r})")
    if wgrad not in (None, "f16x2", "bf16x3", "bf16x6"):
        raise ValueError(f"arithmetic: wgrad must be 'f16x2', 'bf16x3' or 'bf16x6' (got {wgrad!r})")
    saved = ARITH_FWD, ARITH_WGRAD, FACTORED_DGRAD
    ARITH_FWD = fwd or ARITH_FWD
    ARITH_WGRAD = wgrad or ARITH_WGRAD
    FACTORED_DGRAD = FACTORED_DGRAD if factored_dgrad is None else bool(factored_dgrad)
    try:
        yield
    finally:
        ARITH_FWD, ARITH_WGRAD, FACTORED_DGRAD = saved


ABSMAX_SLOTS = 64      # MDNVAE_ABSMAX_SLOTS: partial maxima per tensor


def _absmax_slots(n, device, zero=False):
    """n largest-magnitude records (ABSMAX_SLOTS bit patterns of floats each) on the device, as a list; the
    kernels that fill them clear them first (reset) - or, zero=True, they start cleared by one fill."""
    return list((torch.zeros if zero else torch.empty)(n, ABSMAX_SLOTS, dtype=torch.int32, device=device).unbind(0))


def _check_absmax_slot(slot, device):
    if (not torch.is_tensor(slot) or slot.dtype != torch.int32 or slot.numel() < ABSMAX_SLOTS or not slot.is_contiguous()
            or slot.device != device):
        raise RuntimeError(f"a largest-magnitude record is a contiguous int32[{ABSMAX_SLOTS}] tensor on {device} "
                           f"(got {getattr(slot, 'dtype', type(slot))}, {tuple(getattr(slot, 'shape', ()))})")


def absmax_value(slot):
    """The magnitude a record holds (a host read: for tests and diagnostics)."""
    return float(slot.view(torch.float32).max().item())


def _absmax(t, slot, reset=True, floor=0.0):
    rows, cols = (t.shape[0], t.shape[1]) if t.dim() == 2 else (1, t.numel())
    call("mdnvae_absmax", ptr(t), rows, cols, cols, float(floor), int(reset), slot.data_ptr(), stream())


def _split_linear_fwd(x, weight, bias, act):
    B, K = x.shape
    n_out = weight.shape[0]
    dev = x.device
    Kp, Mp, Np = _round_up(K, 16), _round_up(B, 128), _round_up(n_out, 256)
    wb = _split_operand(Np, Kp, dev)
    wb[(n_out + 31) // 32 * (Kp // 16) * 3072:].zero_()
    xa = _split_operand(Mp, Kp, dev)
    y = torch.empty(B, n_out, dtype=torch.float32, device=dev)
    if ARITH_FWD == "f16x2":
        mx = _absmax_slots(2, dev)
        _absmax(weight, mx[0])
        _absmax(x, mx[1])
        call("mdnvae_split_rows_f16", ptr(weight, "weight"), n_out, K, K, 0, mx[0].data_ptr(), wb.data_ptr(), stream())
        call("mdnvae_split_rows_f16", ptr(x, "x"), B, K, K, 0, mx[1].data_ptr(), xa.data_ptr(), stream())
        call("mdnvae_split_gemm_f16", xa.data_ptr(), wb.data_ptr(), mx[1].data_ptr(), mx[0].data_ptr(), Mp, Np, Kp, B, n_out,
             ptr(bias, "bias"), None, int(act == ACTS["relu"]), 0, ptr(y), None, Np, n_out, 128, -1, stream())
    else:
        call("mdnvae_split_rows", ptr(weight, "weight"), n_out, K, K, 0, wb.data_ptr(), stream())
        call("mdnvae_split_rows", ptr(x, "x"), B, K, K, 0, xa.data_ptr(), stream())
        call("mdnvae_split_gemm", xa.data_ptr(), wb.data_ptr(), Mp, Np, Kp, B, n_out, ptr(bias, "bias"), None,
             int(act == ACTS["relu"]), 0, ptr(y), None, Np, n_out, 128, 6, stream())
    if act not in (ACTS[None], ACTS["relu"]):
        call("mdnvae_bias_act", ptr(y), None, B, n_out, act, stream())
    return y


def _split_linear_dgrad(gpre, weight):
    """gx [B, n_in] = gpre [B, n_out] . weight [n_out, n_in]"""
    B, n_out = gpre.shape
    n_in = weight.shape[1]
    dev = gpre.device
    Kp, Mp, Np = _round_up(n_out, 16), _round_up(B, 128), _round_up(n_in, 256)
    ga = _split_operand(Mp, Kp, dev)
    wb = _split_operand(Np, Kp, dev)
    wb[(n_in + 31) // 32 * (Kp // 16) * 3072:].zero_()
    gx = torch.empty(B, n_in, dtype=torch.float32, device=dev)
    if ARITH_FWD == "f16x2":
        mx = _absmax_slots(2, dev)
        _absmax(gpre, mx[0])
        _absmax(weight, mx[1])
        call("mdnvae_split_rows_f16", ptr(gpre, "grad_pre"), B, n_out, n_out, 0, mx[0].data_ptr(), ga.data_ptr(), stream())
        call("mdnvae_split_cols_f16", ptr(weight, "weight"), None, n_out, n_in, n_in, 0, 0, mx[1].data_ptr(), wb.data_ptr(),
             stream())
        call("mdnvae_split_gemm_f16", ga.data_ptr(), wb.data_ptr(), mx[0].data_ptr(), mx[1].data_ptr(), Mp, Np, Kp, B, n_in,
             None, None, 0, 0, ptr(gx), None, Np, n_in, 128, -1, stream())
        return gx
    call("mdnvae_split_rows", ptr(gpre, "grad_pre"), B, n_out, n_out, 0, ga.data_ptr(), stream())
    call("mdnvae_split_cols", ptr(weight, "weight"), n_out, n_in, n_in, 0, wb.data_ptr(), stream())
    call("mdnvae_split_gemm", ga.data_ptr(), wb.data_ptr(), Mp, Np, Kp, B, n_in, None, None, 0, 0, ptr(gx), None, Np, n_in, 128, 6, stream())
    return gx


def _split_linear_wgrad(gpre, x, param, arith):
    """grad_w [n_out, n_in] = gpre^T . x; None when it was added straight into param.grad"""
    B, n_out = gpre.shape
    n_in = x.shape[1]
    dev = x.device
    Kp, Mp, Np = _round_up(B, 16), _round_up(n_out, 128), _round_up(n_in, 256)
    pa = _split_operand(Mp, Kp, dev)
    pa[(n_out + 31) // 32 * (Kp // 16) * 3072:].zero_()
    xb = _split_operand(Np, Kp, dev)
    xb[(n_in + 31) // 32 * (Kp // 16) * 3072:].zero_()
    target = _grad_target(param)
    out = target if target is not None else torch.empty(n_out, n_in, dtype=torch.float32, device=dev)
    if arith == "f16x2":
        mx = _absmax_slots(2, dev)
        _absmax(gpre, mx[0])
        _absmax(x, mx[1])
        call("mdnvae_split_cols_f16", ptr(gpre, "grad_pre"), None, B, n_out, n_out, 0, 0, mx[0].data_ptr(), pa.data_ptr(), stream())
        call("mdnvae_split_cols_f16", ptr(x, "x"), None, B, n_in, n_in, 0, 0, mx[1].data_ptr(), xb.data_ptr(), stream())
        call("mdnvae_split_gemm_f16", pa.data_ptr(), xb.data_ptr(), mx[0].data_ptr(), mx[1].data_ptr(), Mp, Np, Kp, n_out, n_in,
             None, None, 0, int(target is not None), ptr(out), None, Np, n_in, 128, -1, stream())
    else:
        parts, terms = (3, 6) if arith == "bf16x6" else (2, 3)
        call("mdnvae_split_cols_gated", ptr(gpre, "grad_pre"), None, B, n_out, n_out, 0, parts, 0, pa.data_ptr(), stream())
        call("mdnvae_split_cols_gated", ptr(x, "x"), None, B, n_in, n_in, 0, parts, 0, xb.data_ptr(), stream())
        call("mdnvae_split_gemm", pa.data_ptr(), xb.data_ptr(), Mp, Np, Kp, n_out, n_in, None, None, 0, int(target is not None),
             ptr(out), None, Np, n_in, 128, terms, stream())
    if target is not None:
        _grad_added(param)
        return None
    return out


class _FactoredGrad:
    """Where the gradient of a tensor may be left as a product instead of being formed: the producer of y
    (the first-layer pair) hangs one on each output tensor OBJECT, the dense layer that consumes that very
    object (not a view, slice, reshape or detached alias of it: those are other objects and never carry the
    record) finds it at forward time and, in its backward, hands over (grad_pre, weight) - its data gradient
    is grad_pre . weight - and returns zeros that occupy no memory.  The producer's backward forms the
    product inside the pass that needs it anyway (ReLU gate + operand split).  Anything else that flows
    back into y arrives the usual way and is added.

    Consequence for callers: the gradient of y itself is then not observable - torch.autograd.grad(loss, y),
    y.register_hook and y.retain_grad() would see the zeros.  A y that already has hooks or retains its
    gradient when the consumer runs is therefore left alone (dense data gradient); to observe y's gradient
    in every case run under factored_dgrad(False)."""
    __slots__ = ("shape", "taken", "gpre", "weight")

    def __init__(self, shape):
        self.shape, self.taken, self.gpre, self.weight = tuple(shape), False, None, None


_FACTORED_ATTR = "_mdnvae_factored_grad"
FACTORED_DGRAD = True


def _factored_slot_of(x, weight):
    """The record x carries if this consumer may take it: x is the producer's own output (same object, same
    [B,H] shape, dense rows), part of the graph, unobserved, and the weight is one the producer's kernel handles."""
    slot = getattr(x, _FACTORED_ATTR, None)
    if slot is None or slot.taken or not x.requires_grad or tuple(x.shape) != slot.shape or not x.is_contiguous():
        return None
    if x.retains_grad or getattr(x, "_backward_hooks", None):
        return None
    if weight.dim() != 2 or weight.shape[1] != slot.shape[1] or weight.shape[0] not in (64, 128) or weight.shape[1] % 4:
        return None
    return slot


_ZERO_SCALARS = {}


def _zero_scalar(device):
    """One zero per device, made once (a fill launch each time otherwise); only ever read, through expanded views."""
    key = str(device)
    if key not in _ZERO_SCALARS:
        _ZERO_SCALARS[key] = torch.zeros((), dtype=torch.float32, device=device)
    return _ZERO_SCALARS[key]


def _is_free_zeros(t):
    return t is not None and t.dim() == 2 and t.stride() == (0, 0)


class _LinearAct(torch.autograd.Function):
    """act(x @ W^T + b): the product is a rocBLAS GEMM - or, for large layers, the split-bf16 product -
    bias+activation and the backward's elementwise / column-sum parts are HIP kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x, weight = _c(x), _c(weight)
        ptr(x, "x"), ptr(weight, "weight")
        if _narrow(weight) and x.dim() == 2:
            y = torch.empty(x.shape[0], weight.shape[0], dtype=torch.float32, device=x.device)
            call("mdnvae_narrow_linear_fwd", ptr(x), ptr(weight), ptr(bias, "bias"), x.shape[0], weight.shape[0],
                 weight.shape[1], act, ptr(y), stream())
        elif _split_linear_ok(x, weight, bias):
            y = _split_linear_fwd(x, weight, bias, act)
        else:
            y = _library_linear(x, weight.t(), bias, act)
        ctx.save_for_backward(x, weight, y)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)              # the objects whose .grad a sink may add into
        ctx.arith_wgrad = ARITH_WGRAD
        ctx.factored = _factored_slot_of(x, weight) if FACTORED_DGRAD else None
        if ctx.factored is not None:
            ctx.factored.taken = True
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = _c(gy)
        if ctx.act != 0:
            gpre = torch.empty_like(gy)
            call("mdnvae_act_bwd", ptr(y), ptr(gy), ptr(gpre), gy.numel(), ctx.act, stream())
        else:
            gpre = gy
        split = ctx.has_bias and _split_linear_ok(x, weight, ctx.params[1])
        if ctx.needs_input_grad[0] and ctx.factored is not None:
            ctx.factored.gpre, ctx.factored.weight = _c(gpre), weight
            gx = _zero_scalar(gy.device).expand(x.shape)
        elif ctx.needs_input_grad[0]:
            gx = _split_linear_dgrad(_c(gpre), weight) if split else _library_gemm(gpre, weight)
        else:
            gx = None
        if _narrow(weight) and x.shape[0] >= 1 and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            gw, gb = _narrow_wgrad(_c(gpre), x, ctx.params[0], ctx.params[1], ctx.needs_input_grad[1],
                                   ctx.has_bias and ctx.needs_input_grad[2])
            return gx, gw, gb, None
        gw = None
        if ctx.needs_input_grad[1]:
            if split:
                gw = _split_linear_wgrad(_c(gpre), x, ctx.params[0], ctx.arith_wgrad)
            elif (_small_wgrad_ok(x.shape[0], weight.shape[0], weight.shape[1]) and ctx.has_bias and ctx.needs_input_grad[2]
                  and x.dim() == 2):
                gw, gb = _linear_wgrad_bias(_c(gpre), x, ctx.params[0], ctx.params[1])
                return gx, gw, gb, None
            elif _small_wgrad_ok(x.shape[0], weight.shape[0], weight.shape[1]):
                target = _grad_target(ctx.params[0])
                if target is not None:
                    _linear_wgrad(_c(gpre), x, into=target)
                    _grad_added(ctx.params[0])
                else:
                    gw = _linear_wgrad(_c(gpre), x)
            else:
                # [K,B].[B,H] then a transposed view: the library runs this layout ~20% faster than gpre^T . x
                gw = _library_gemm(x.t(), gpre).t()
        gb = _bias_grad(gpre, ctx.params[1]) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None


def linear_act(x, weight, bias=None, act=None):
    return _LinearAct.apply(x, weight, bias, ACTS[act])


def tri_pack(x, n_nodes, asym, absmax=None):
    """Upper triangles [B, tri_row_stride(N)] of a batch of symmetric matrices x [B, N*N]; `asym` (int32[1]
    on the device) is OR-ed with 1 if a matrix is not symmetric.  The result carries no gradient:
    the symmetric first layer is for network inputs.  `absmax` (int32[ABSMAX_SLOTS] on the device, the record
    _absmax_slots makes): filled with partial maxima (bit patterns) of the magnitudes packed, which
    sym_linear_pair(x_absmax=...) takes as the operand's scale."""
    x = _c(x.detach()).reshape(-1, n_nodes * n_nodes)
    if absmax is not None:
        _check_absmax_slot(absmax, x.device)
    ld = tri_row_stride(n_nodes)
    out = torch.empty(x.shape[0], ld, dtype=torch.float32, device=x.device)
    if absmax is not None:
        call("mdnvae_tri_pack_absmax", ptr(x, "x"), x.shape[0], n_nodes, ld, ptr(out), asym.data_ptr(), 1,
             absmax.data_ptr(), stream())
    else:
        call("mdnvae_tri_pack", ptr(x, "x"), x.shape[0], n_nodes, ld, ptr(out), asym.data_ptr(), stream())
    return out


def tri_row_stride(n_nodes):
    """Row length of the packed triangles: N(N+1)/2 rounded up to a multiple of 64 floats (zero
    filled), the inner-dimension granularity the library GEMM kernels run fastest at."""
    return (n_nodes * (n_nodes + 1) // 2 + 63) // 64 * 64


class _SymLinearAct(torch.autograd.Function):
    """act(x @ W^T + b) for symmetric x given as packed upper triangles: the weight is folded onto
    the triangle (w_ij + w_ji), the two GEMMs run at half the inner / outer dimension, and the
    folded weight gradient is written back to both (i,j) and (j,i)."""

    @staticmethod
    def forward(ctx, xt, weight, bias, act, n_nodes):
        weight = _c(weight)
        H, T = weight.shape[0], xt.shape[1]
        wt = torch.empty(H, T, dtype=torch.float32, device=xt.device)
        call("mdnvae_tri_fold", ptr(weight, "weight"), H, n_nodes, T, ptr(wt), stream())
        y = _library_linear(xt, wt.t(), bias, act)
        ctx.save_for_backward(xt, y)
        ctx.act, ctx.n_nodes, ctx.has_bias = act, n_nodes, bias is not None
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        xt, y = ctx.saved_tensors
        gy = _c(gy)
        if ctx.act != 0:
            gpre = torch.empty_like(gy)
            call("mdnvae_act_bwd", ptr(y), ptr(gy), ptr(gpre), gy.numel(), ctx.act, stream())
        else:
            gpre = gy
        gw = None
        if ctx.needs_input_grad[1]:
            H, n = gpre.shape[1], ctx.n_nodes
            gt = _library_gemm(xt.t(), gpre)                       # [T, H]
            target = _grad_target(ctx.params[0])
            if target is not None and _SINK.folded is not None:
                _SINK.folded(ctx.params[0], gt, n)
            elif target is not None:
                call("mdnvae_tri_unfold", ptr(gt), H, n, H, 1, ptr(target), 1, stream())
                _grad_added(ctx.params[0])
            else:
                gw = torch.empty(H, n * n, dtype=torch.float32, device=gy.device)
                call("mdnvae_tri_unfold", ptr(gt), H, n, H, 1, ptr(gw), 0, stream())
        gb = _bias_grad(gpre, ctx.params[1]) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return None, gw, gb, None, None


def sym_linear_act(x_tri, weight, bias=None, act=None, n_nodes=None):
    return _SymLinearAct.apply(x_tri, weight, bias, ACTS[act], n_nodes)


# ---------------------------------------------------------------------------------------
# both symmetric-input first layers as one split-bf16 product (fp32 accuracy on the bf16 matrix cores)
# ---------------------------------------------------------------------------------------
SPLIT_GEMM_MIN_BATCH = 1024      # below this the 256-row output tiles leave most of the chip idle


def _split_operand(rows, k, device):
    return torch.empty(int(_lib.load().mdnvae_split_operand_bytes(rows, k)), dtype=torch.uint8, device=device)


def _round_up(v, m):
    return (v + m - 1) // m * m


def _pair_wgrad_shape(B, T, H, n_nodes, arith):
    """(parts, Kp, t_real, rows, halves, tile, Tp) of the weight-gradient product of the symmetric first-layer pair."""
    parts = 3 if arith == "bf16x6" else 2
    Kp = _round_up(B, 16)
    # the row of ones goes where the packed triangle's zero padding starts (one past the end if there is none)
    t_real = min(n_nodes * (n_nodes + 1) // 2, T)
    rows = max(T, t_real + 1)
    # few output tiles and a long k (the batch): 256-row tiles with the k range split in two keep twice
    # the fragments in registers per LDS byte read
    halves = arith != "bf16x6" and SPLIT_WGRAD_KSPLIT and B >= 4096 and (rows + 255) // 256 * (2 * H // 256) <= 128
    tile = 256 if halves else 128
    return parts, Kp, t_real, rows, halves, tile, _round_up(rows, tile)


class _SymLinearPair(torch.autograd.Function):
    """relu(x W1^T + b1), relu(x W2^T + b2) for symmetric x given as packed upper triangles xt [B,T]
    (self.fc11 / self.fc12 of function/MM_Gate.py:157-158, which read the same input): the two folded
    weights are stacked, every operand is split into three bf16 parts, and ONE hand-written MFMA
    kernel forms the product with bias + ReLU in its epilogue; the backward forms both folded weight
    gradients [T,H] with the same kernel from x^T and grad_pre^T."""

    @staticmethod
    def forward(ctx, xt, w1, b1, w2, b2, n_nodes, x_absmax=None):
        w1, w2 = _c(w1), _c(w2)
        B, T = xt.shape
        H = w1.shape[0]
        if x_absmax is not None:
            _check_absmax_slot(x_absmax, xt.device)
        if w2.shape != w1.shape or H % 256 != 0 or T % 16 != 0 or b1 is None or b2 is None:
            raise RuntimeError(f"sym_linear_pair: two [H,N*N] weights with biases, H a multiple of 256 (got {tuple(w1.shape)}, "
                               f"{tuple(w2.shape)}), T a multiple of 16 (got {T})")
        dev = xt.device
        xb = None
        wt = torch.empty(2 * H, T, dtype=torch.float32, device=dev)
        wb = _split_operand(2 * H, T, dev)
        Mp = _round_up(B, 256)
        xa = _split_operand(Mp, T, dev)
        y1 = torch.empty(B, H, dtype=torch.float32, device=dev)
        y2 = torch.empty(B, H, dtype=torch.float32, device=dev)
        if ARITH_FWD == "f16x2":
            # the folding pass (and, given x_absmax, the packing pass) leaves the largest magnitude it wrote
            mx = _absmax_slots(2, dev)
            wmax, xmax = mx[0], (x_absmax if x_absmax is not None else mx[1])
            call("mdnvae_tri_fold_absmax", ptr(w1, "weight"), H, n_nodes, T, ptr(wt[:H]), 1, wmax.data_ptr(), stream())
            call("mdnvae_tri_fold_absmax", ptr(w2, "weight"), H, n_nodes, T, ptr(wt[H:]), 0, wmax.data_ptr(), stream())
            if x_absmax is None:
                _absmax(xt, xmax)
            call("mdnvae_split_rows_f16", ptr(wt), 2 * H, T, T, 0, wmax.data_ptr(), wb.data_ptr(), stream())
            ctx.xmax = xmax
            if any(ctx.needs_input_grad[1:5]) and T % 16 == 0:
                # the backward's transposed operand of the batch from the same read of it
                parts, Kp, t_real, rows, halves, tile, Tp = _pair_wgrad_shape(B, T, H, n_nodes, ARITH_WGRAD)
                xb = _split_operand(Tp, Kp, dev)
                call("mdnvae_split_rows_f16_cols", ptr(xt, "x"), B, t_real, T, 1, 0 if ARITH_WGRAD == "f16x2" else parts,
                     xmax.data_ptr(), xa.data_ptr(), xb.data_ptr(), stream())
            else:
                call("mdnvae_split_rows_f16", ptr(xt, "x"), B, T, T, 0, xmax.data_ptr(), xa.data_ptr(), stream())
            call("mdnvae_split_gemm_f16", xa.data_ptr(), wb.data_ptr(), xmax.data_ptr(), wmax.data_ptr(), Mp, 2 * H, T, B, H,
                 ptr(b1, "bias"), ptr(b2, "bias"), 1, 0, ptr(y1), ptr(y2), H, H, 256, -1, stream())
        else:
            ctx.xmax = None
            call("mdnvae_tri_fold", ptr(w1, "weight"), H, n_nodes, T, ptr(wt[:H]), stream())
            call("mdnvae_tri_fold", ptr(w2, "weight"), H, n_nodes, T, ptr(wt[H:]), stream())
            call("mdnvae_split_rows", ptr(wt), 2 * H, T, T, 0, wb.data_ptr(), stream())
            call("mdnvae_split_rows", ptr(xt, "x"), B, T, T, 0, xa.data_ptr(), stream())
            call("mdnvae_split_gemm", xa.data_ptr(), wb.data_ptr(), Mp, 2 * H, T, B, H, ptr(b1, "bias"), ptr(b2, "bias"), 1, 0,
                 ptr(y1), ptr(y2), H, H, 256, 6, stream())
        ctx.save_for_backward(xt, y1, y2)
        ctx.xb = xb
        ctx.n_nodes = n_nodes
        ctx.params = (w1, b1, w2, b2)
        ctx.arith_wgrad = ARITH_WGRAD
        ctx.slots = None
        if FACTORED_DGRAD and any(ctx.needs_input_grad[1:5]):
            ctx.slots = (_FactoredGrad(y1.shape), _FactoredGrad(y2.shape))
            for y, slot in zip((y1, y2), ctx.slots):
                setattr(y, _FACTORED_ATTR, slot)
        return y1, y2

    @staticmethod
    def backward(ctx, g1, g2):
        xt, y1, y2 = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        B, T = xt.shape
        H, n, dev = y1.shape[1], ctx.n_nodes, xt.device
        gws, gbs = [None, None], [None, None]
        if not any(ctx.needs_input_grad[1:5]):
            return None, None, None, None, None, None, None
        slots = ctx.slots or (None, None)
        # One product gives both folded weight gradients and both bias gradients: its operands are x^T with a
        # row of ones appended (so that row T of the result is the column sum of grad_pre) and grad_pre^T,
        # which is split straight from the output gradients with the ReLU gate applied on the way
        # (grad_pre itself is never written)
        arith = ctx.arith_wgrad
        f16 = arith == "f16x2"
        parts, Kp, t_real, rows, halves, tile, Tp = _pair_wgrad_shape(B, T, H, n, arith)
        pb = _split_operand(2 * H, Kp, dev)
        towers = []
        for y, g, slot in ((y1, g1, slots[0]), (y2, g2, slots[1])):
            gpre_next = w_next = None
            if slot is not None and slot.gpre is not None:
                gpre_next, w_next, slot.gpre, slot.weight = slot.gpre, slot.weight, None, None
                if not (gpre_next.dim() == 2 and gpre_next.shape[0] == B and gpre_next.is_contiguous()
                        and tuple(w_next.shape) == (gpre_next.shape[1], H) and w_next.is_contiguous()):
                    # a hand-over the fused pass cannot take (_LinearAct checks at forward time and never leaves
                    # one): the product is formed the ordinary way
                    prod = _library_gemm(gpre_next, w_next)
                    g = prod if _is_free_zeros(g) else g + prod
                    gpre_next = w_next = None
            dense = None if (gpre_next is not None and _is_free_zeros(g)) else _c(g)
            towers.append((y, dense, gpre_next, w_next))
        gmax = None
        if f16:
            # one scale for both towers' gradients (they are the columns of one operand): their largest magnitude,
            # or - where a gradient is only known as a product - a bound of it from the factors
            mx = _absmax_slots(5, dev, zero=True)
            gmax = mx[4]
            for i, (y, dense, gpre_next, w_next) in enumerate(towers):
                if gpre_next is None:
                    _absmax(dense, gmax, reset=False)
                    continue
                _absmax(gpre_next, mx[2 * i], reset=False)
                if dense is not None:
                    _absmax(dense, mx[2 * i + 1], reset=False)
                call("mdnvae_dgrad_bound", mx[2 * i].data_ptr(), ptr(w_next), w_next.shape[0], H,
                     mx[2 * i + 1].data_ptr() if dense is not None else None, 0, gmax.data_ptr(), stream())
        for i, (y, dense, gpre_next, w_next) in enumerate(towers):
            if gpre_next is not None and f16:
                # the consuming layer left its data gradient as a product: formed here, under the gate
                call("mdnvae_split_cols_dgrad_f16", ptr(gpre_next), ptr(w_next), ptr(dense), ptr(y), B, H, gpre_next.shape[1],
                     i * (H // 32), gmax.data_ptr(), mx[2 * i].data_ptr(), pb.data_ptr(), stream())
            elif gpre_next is not None:
                call("mdnvae_split_cols_dgrad", ptr(gpre_next), ptr(w_next), ptr(dense), ptr(y), B, H, gpre_next.shape[1],
                     parts, i * (H // 32), pb.data_ptr(), stream())
            elif f16:
                call("mdnvae_split_cols_f16", ptr(dense), ptr(y), B, H, H, 0, i * (H // 32), gmax.data_ptr(), pb.data_ptr(),
                     stream())
            else:
                call("mdnvae_split_cols_gated", ptr(dense), ptr(y), B, H, H, 0, parts, i * (H // 32), pb.data_ptr(), stream())
        xb, ctx.xb = ctx.xb, None
        xmax = ctx.xmax
        if xb is None:
            xb = _split_operand(Tp, Kp, dev)
            if f16:
                if xmax is None:
                    xmax = _absmax_slots(1, dev)[0]
                    _absmax(xt, xmax)
                call("mdnvae_split_cols_f16", ptr(xt), None, B, t_real, T, 1, 0, xmax.data_ptr(), xb.data_ptr(), stream())
            else:
                call("mdnvae_split_cols_gated", ptr(xt), None, B, t_real, T, 1, parts, 0, xb.data_ptr(), stream())
        gts = list((torch.zeros if halves else torch.empty)(2, rows, H, dtype=torch.float32, device=dev).unbind(0))    # one fill
        if f16:
            call("mdnvae_split_gemm_f16", xb.data_ptr(), pb.data_ptr(), xmax.data_ptr(), gmax.data_ptr(), Tp, 2 * H, Kp,
                 t_real + 1, H, None, None, 0, 2 if halves else 0, ptr(gts[0]), ptr(gts[1]), H, H, tile, t_real, stream())
        else:
            call("mdnvae_split_gemm", xb.data_ptr(), pb.data_ptr(), Tp, 2 * H, Kp, t_real + 1, H, None, None, 0,
                 2 if halves else 0, ptr(gts[0]), ptr(gts[1]), H, H, tile, 6 if arith == "bf16x6" else 3, stream())
        for i, (param, bias) in enumerate(((w1, b1), (w2, b2))):
            gt, gb = gts[i][:t_real], gts[i][t_real]
            if ctx.needs_input_grad[2 + 2 * i]:
                target = _grad_target(bias)
                if target is not None:
                    target.add_(gb)
                    _grad_added(bias)
                else:
                    gbs[i] = gb
            if not ctx.needs_input_grad[1 + 2 * i]:
                continue
            target = _grad_target(param)
            if target is not None and _SINK.folded is not None:
                _SINK.folded(param, gt, n)
            elif target is not None:
                call("mdnvae_tri_unfold", ptr(gt), H, n, H, 1, ptr(target), 1, stream())
                _grad_added(param)
            else:
                gws[i] = torch.empty(H, n * n, dtype=torch.float32, device=dev)
                call("mdnvae_tri_unfold", ptr(gt), H, n, H, 1, ptr(gws[i]), 0, stream())
        return None, gws[0], gbs[0], gws[1], gbs[1], None, None


def sym_linear_pair(x_tri, w1, b1, w2, b2, n_nodes, x_absmax=None):
    """x_absmax: the slot tri_pack(..., absmax=slot) filled for x_tri (saves a pass over x_tri)."""
    return _SymLinearPair.apply(x_tri, w1, b1, w2, b2, n_nodes, x_absmax)


class _LinearPair(torch.autograd.Function):
    """relu(x W1^T + b1), relu(x W2^T + b2) on all columns of x [B,D] (self.fc11 / self.fc12 of
    function/MM_Gate.py:157-158 as the reference runs them: no symmetry promise): one split-bf16 product
    forward, one per weight for the gradients, which land in the parameters' own [H,D] layout (added
    straight into .grad when a sink is active).  D need not be a multiple of anything: the splitting
    kernels take the rows as they lie (N*N is odd for the connectomes) and pad k with zeros."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x, w1, w2 = _c(x), _c(w1), _c(w2)
        B, D = x.shape
        H = w1.shape[0]
        if w2.shape != w1.shape or w1.shape[1] != D or H % 256 != 0 or b1 is None or b2 is None:
            raise RuntimeError(f"linear_pair: two [H,{D}] weights with biases, H a multiple of 256 (got {tuple(w1.shape)}, "
                               f"{tuple(w2.shape)})")
        dev = x.device
        Kp = _round_up(D, 16)
        wb = _split_operand(2 * H, Kp, dev)
        Mp = _round_up(B, 256)
        xa = _split_operand(Mp, Kp, dev)
        y1 = torch.empty(B, H, dtype=torch.float32, device=dev)
        y2 = torch.empty(B, H, dtype=torch.float32, device=dev)
        if ARITH_FWD == "f16x2":
            mx = _absmax_slots(2, dev)
            wmax, xmax = mx[0], mx[1]
            _absmax(w1, wmax)
            _absmax(w2, wmax, reset=False)
            _absmax(x, xmax)
            call("mdnvae_split_rows_f16", ptr(w1, "weight"), H, D, D, 0, wmax.data_ptr(), wb.data_ptr(), stream())
            call("mdnvae_split_rows_f16", ptr(w2, "weight"), H, D, D, H // 32, wmax.data_ptr(), wb.data_ptr(), stream())
            call("mdnvae_split_rows_f16", ptr(x, "x"), B, D, D, 0, xmax.data_ptr(), xa.data_ptr(), stream())
            call("mdnvae_split_gemm_f16", xa.data_ptr(), wb.data_ptr(), xmax.data_ptr(), wmax.data_ptr(), Mp, 2 * H, Kp, B, H,
                 ptr(b1, "bias"), ptr(b2, "bias"), 1, 0, ptr(y1), ptr(y2), H, H, 256, -1, stream())
            ctx.xmax = xmax
        else:
            ctx.xmax = None
            call("mdnvae_split_rows", ptr(w1, "weight"), H, D, D, 0, wb.data_ptr(), stream())
            call("mdnvae_split_rows", ptr(w2, "weight"), H, D, D, H // 32, wb.data_ptr(), stream())
            call("mdnvae_split_rows", ptr(x, "x"), B, D, D, 0, xa.data_ptr(), stream())
            call("mdnvae_split_gemm", xa.data_ptr(), wb.data_ptr(), Mp, 2 * H, Kp, B, H, ptr(b1, "bias"), ptr(b2, "bias"), 1, 0,
                 ptr(y1), ptr(y2), H, H, 256, 6, stream())
        ctx.save_for_backward(x, y1, y2)
        ctx.params = (w1, b1, w2, b2)
        ctx.arith_wgrad = ARITH_WGRAD
        return y1, y2

    @staticmethod
    def backward(ctx, g1, g2):
        x, y1, y2 = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        B, D = x.shape
        H, dev = y1.shape[1], x.device
        gpre = []
        for y, g in ((y1, g1), (y2, g2)):
            gp = torch.empty_like(y)
            call("mdnvae_act_bwd", ptr(y), ptr(_c(g)), ptr(gp), gp.numel(), ACTS["relu"], stream())
            gpre.append(gp)
        gws = [None, None]
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[3]:
            Kp, Np = _round_up(B, 16), _round_up(D, 256)
            xb = _split_operand(Np, Kp, dev)
            tail = (D + 31) // 32 * (Kp // 16) * 3072          # operand row tiles past D: zero (their columns are not stored)
            xb[tail:].zero_()
            arith = ctx.arith_wgrad
            f16 = arith == "f16x2"
            parts, terms = (3, 6) if arith == "bf16x6" else (2, 3)
            if f16:
                xmax = ctx.xmax
                if xmax is None:
                    xmax = _absmax_slots(1, dev)[0]
                    _absmax(x, xmax)
                gmax = _absmax_slots(1, dev)[0]
                call("mdnvae_split_cols_f16", ptr(x), None, B, D, D, 0, 0, xmax.data_ptr(), xb.data_ptr(), stream())
            else:
                call("mdnvae_split_cols_gated", ptr(x), None, B, D, D, 0, parts, 0, xb.data_ptr(), stream())
            pb = _split_operand(H, Kp, dev)
            for i, (param, gp) in enumerate(((w1, gpre[0]), (w2, gpre[1]))):
                if not ctx.needs_input_grad[1 + 2 * i]:
                    continue
                target = _grad_target(param)
                out = target if target is not None else torch.empty(H, D, dtype=torch.float32, device=dev)
                if f16:
                    _absmax(gp, gmax)
                    call("mdnvae_split_cols_f16", ptr(gp), None, B, H, H, 0, 0, gmax.data_ptr(), pb.data_ptr(), stream())
                    call("mdnvae_split_gemm_f16", pb.data_ptr(), xb.data_ptr(), gmax.data_ptr(), xmax.data_ptr(), H, Np, Kp, H, D,
                         None, None, 0, int(target is not None), ptr(out), None, Np, D, 128, -1, stream())
                else:
                    call("mdnvae_split_cols_gated", ptr(gp), None, B, H, H, 0, parts, 0, pb.data_ptr(), stream())
                    call("mdnvae_split_gemm", pb.data_ptr(), xb.data_ptr(), H, Np, Kp, H, D, None, None, 0, int(target is not None),
                         ptr(out), None, Np, D, 128, terms, stream())
                if target is not None:
                    _grad_added(param)
                else:
                    gws[i] = out
        gb1 = _bias_grad(gpre[0], b1) if ctx.needs_input_grad[2] else None
        gb2 = _bias_grad(gpre[1], b2) if ctx.needs_input_grad[4] else None
        return None, gws[0], gb1, gws[1], gb2


def linear_pair(x, w1, b1, w2, b2):
    return _LinearPair.apply(x, w1, b1, w2, b2)


class _AddAct(torch.autograd.Function):
    """act(a + b) with one kernel each way (residual joins)."""

    @staticmethod
    def forward(ctx, a, b, act):
        a, b = _c(a), _c(b)
        if a.shape != b.shape:
            raise RuntimeError(f"add_act: shapes {tuple(a.shape)} and {tuple(b.shape)} differ")
        y = torch.empty_like(a)
        call("mdnvae_add_act", ptr(a, "a"), ptr(b, "b"), ptr(y), a.numel(), act, stream())
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = _c(gy)
        if ctx.act == 0:
            return gy, gy, None
        g = torch.empty_like(gy)
        call("mdnvae_act_bwd", ptr(y), ptr(gy), ptr(g), gy.numel(), ctx.act, stream())
        return g, g, None


def add_act(a, b, act=None):
    return _AddAct.apply(a, b, ACTS[act])


class _OuterSum(torch.autograd.Function):
    """out [B,N*N] = act(sum_r g_r g_r^T) of node vectors g [B,R,N] (the decoder's outer-product
    stage on its own)."""

    @staticmethod
    def forward(ctx, g, epi_act):
        if g.dim() != 3:
            raise RuntimeError(f"outer_sum: node vectors must be [B,R,N], got {tuple(g.shape)}")
        B, R, N = g.shape
        nodes = torch.zeros(B, R, DEC_NP, dtype=torch.float32, device=g.device)
        nodes[:, :, :N].copy_(g)
        out = torch.empty(B, N * N, dtype=torch.float32, device=g.device)
        call("mdnvae_outer_sum_fwd", ptr(nodes, "nodes"), B, R, N, epi_act, ptr(out), stream())
        ctx.save_for_backward(nodes)
        ctx.meta = (B, R, N, epi_act)
        return out

    @staticmethod
    def backward(ctx, gout):
        (nodes,) = ctx.saved_tensors
        B, R, N, epi_act = ctx.meta
        gout = _c(gout)
        gn = torch.empty_like(nodes)
        call("mdnvae_outer_sum_bwd", ptr(gout), ptr(nodes), B, R, N, epi_act, ptr(gn), stream())
        return gn[:, :, :N].contiguous(), None


def outer_sum(g, epi_act=None):
    return _OuterSum.apply(g, ACTS[epi_act] if not isinstance(epi_act, int) else epi_act)


class _Exp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y = torch.empty_like(x)
        call("mdnvae_exp", ptr(x, "x"), ptr(y), x.numel(), stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = _c(gy)
        gx = torch.empty_like(gy)
        call("mdnvae_mul", ptr(gy), ptr(y), ptr(gx), gy.numel(), stream())
        return gx


def exp(x):
    return _Exp.apply(x)


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        if a.shape != b.shape:
            raise RuntimeError(f"mul: shapes {tuple(a.shape)} and {tuple(b.shape)} differ")
        y = torch.empty_like(a)
        call("mdnvae_mul", ptr(a, "a"), ptr(b, "b"), ptr(y), a.numel(), stream())
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _c(g)
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = torch.empty_like(g)
            call("mdnvae_mul", ptr(g), ptr(b), ptr(ga), g.numel(), stream())
        if ctx.needs_input_grad[1]:
            gb = torch.empty_like(g)
            call("mdnvae_mul", ptr(g), ptr(a), ptr(gb), g.numel(), stream())
        return ga, gb


def mul(a, b):
    return _Mul.apply(a, b)


def diag_affine(x, w, b):
    """x[B,D]*w[D] + b[D] (stand-alone use of the diagonal intercept layer)."""
    y = mul(x, w.unsqueeze(0).expand_as(x).contiguous())
    return _LinearBias.apply(y, b)


class _LinearBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, bias):
        y = y.clone()
        call("mdnvae_bias_act", ptr(y), ptr(_c(bias), "bias"), y.shape[0], y.shape[1], 0, stream())
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        return g, _colsum(g)


# ---------------------------------------------------------------------------------------
# chain decoder
# ---------------------------------------------------------------------------------------
DEC_NP = 96
DEC_KP = 96


class ChainSpec:
    """Static description of a chain decoder: R channels x NL layers.
    layers[l] = dict(kind='linear'|'gcn', act=..., mask=Tensor|None, adj=IntTensor|None)."""

    def __init__(self, n_nodes, latent_dim, channels, layers, epi_act=None, xlat_channels=0, do_exp=False,
                 components=False, nodes_only=False):
        self.N, self.L, self.R = int(n_nodes), int(latent_dim), int(channels)
        self.layers = layers
        self.NL = len(layers)
        self.epi_act = ACTS[epi_act]
        self.xlat_channels = int(xlat_channels)
        self.do_exp = bool(do_exp)
        self.components = bool(components)
        self.nodes_only = bool(nodes_only)      # the decoder returns its node vectors [B,R,NP], no matrices
        self.acts = 0
        for l, ly in enumerate(layers):
            a = ACTS[ly.get("act")]
            if a == 3:
                raise ValueError("chain decoder layers support none/sigmoid/tanh")
            self.acts |= a << (2 * l)


def _layer_tables(spec, params):
    """Host-side tables of the batched pack / unpack calls, layer (r,l) at r*NL+l."""
    import ctypes
    n = spec.R * spec.NL
    kind, n_out, k = (ctypes.c_int * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)()
    w, bias, aux = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
    keep = []                                   # contiguous copies must outlive the launch
    for r in range(spec.R):
        for l, ly in enumerate(spec.layers):
            i = r * spec.NL + l
            wi, bi = _c(params[2 * i]), params[2 * i + 1]
            keep.append(wi)
            kind[i] = 0 if ly["kind"] == "linear" else 1
            n_out[i], k[i] = wi.shape
            w[i], bias[i] = ptr(wi, "weight"), ptr(bi, "bias")
            aux[i] = ptr(ly.get("mask"), "mask") if kind[i] == 0 else ptr(ly["adj"], "adjacency", torch.int32)
    return n, kind, n_out, k, w, bias, aux, keep


def _pack_panels(spec, params, device):
    """params: flat list [w(r=0,l=0), b(0,0), w(0,1), b(0,1), ...] -> (wt, wn) panels, one launch."""
    n, kind, n_out, k, w, bias, aux, keep = _layer_tables(spec, params)
    wt = torch.empty(n * DEC_KP * DEC_NP, dtype=torch.float32, device=device)
    wn = torch.empty(n * DEC_KP * DEC_NP, dtype=torch.float32, device=device)
    call("mdnvae_decoder_pack_layers", n, _addr(kind), _addr(w), _addr(bias), _addr(aux), _addr(n_out), _addr(k),
         ptr(wt), ptr(wn), stream())
    return wt, wn


def _addr(ctypes_array):
    import ctypes
    return ctypes.addressof(ctypes_array)


class _ChainDecoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, spec, icpt_w, icpt_b, drop, *params):
        z = _c(z)
        B = z.shape[0]
        if z.shape[1] != spec.L:
            raise RuntimeError(f"decoder: z has {z.shape[1]} columns, expected latent_dim {spec.L}")
        dev = z.device
        wt, wn = _pack_panels(spec, params, dev)
        D = spec.N * spec.N
        if spec.nodes_only:
            if icpt_w is not None or drop is not None or spec.do_exp:
                raise RuntimeError("decoder: node vectors alone take no intercept, exp or dropout")
            out = None
        else:
            out = torch.empty((B, spec.R, D) if spec.components else (B, D), dtype=torch.float32, device=dev)
        xlat = torch.empty(B, spec.N, dtype=torch.float32, device=dev) if spec.xlat_channels else None
        nodes = torch.empty(B, spec.R, DEC_NP, dtype=torch.float32, device=dev)
        # when a backward pass will follow (and no dropout multiplier hides the last layer's output),
        # the forward keeps every layer's input rows so that backward does not recompute the chain
        hin = None
        if drop is None and any(ctx.needs_input_grad):
            hin = torch.empty(spec.R * spec.NL, B, DEC_KP, dtype=torch.float32, device=dev)
        call("mdnvae_decoder_fwd", ptr(z, "z"), ptr(wt), B, spec.N, spec.L, spec.R, spec.NL, spec.acts, spec.epi_act,
             ptr(icpt_w, "icpt_w"), ptr(icpt_b, "icpt_b"), int(spec.do_exp), ptr(drop, "drop"), spec.xlat_channels,
             int(spec.components), ptr(out), ptr(xlat), ptr(nodes), ptr(hin), stream())
        ctx.spec = spec
        ctx.save_for_backward(z, wt, wn, icpt_w, icpt_b, drop, nodes, hin)
        ctx.param_shapes = [tuple(p.shape) if p is not None else None for p in params]
        ctx.param_objs = params                  # the objects whose .grad a sink may add into
        if xlat is None:
            xlat = torch.zeros(B, 0, dtype=torch.float32, device=dev)
            ctx.mark_non_differentiable(xlat)
        return (nodes if spec.nodes_only else out), xlat

    @staticmethod
    def backward(ctx, gout, gxlat):
        spec = ctx.spec
        z, wt, wn, icpt_w, icpt_b, drop, nodes, saved_hin = ctx.saved_tensors
        B = z.shape[0]
        dev = z.device
        gout = _c(gout)
        if spec.xlat_channels and gxlat is not None:
            gxlat = _c(gxlat)
        else:
            gxlat = None
        RL = spec.R * spec.NL
        hin = saved_hin if saved_hin is not None else torch.empty(RL, B, DEC_KP, dtype=torch.float32, device=dev)
        dp = torch.empty(RL, B, DEC_NP, dtype=torch.float32, device=dev)
        gz = torch.empty_like(z)
        # workspace for one partial latent gradient per channel (the forms that give a workgroup one channel)
        gz_parts = torch.empty(spec.R, B, spec.L, dtype=torch.float32, device=dev) if saved_hin is not None else None
        ga = torch.empty_like(gout) if icpt_w is not None else None
        if spec.nodes_only:                      # gout is the gradient of the node vectors themselves
            node_grad, gout = gout, None
        else:
            node_grad = torch.empty_like(nodes)
        call("mdnvae_decoder_bwd", ptr(z), ptr(wt), ptr(wn), B, spec.N, spec.L, spec.R, spec.NL, spec.acts,
             spec.epi_act, ptr(icpt_w), ptr(icpt_b), int(spec.do_exp), ptr(drop), spec.xlat_channels,
             int(spec.components), ptr(gout),
             ptr(gxlat), ptr(gz), ptr(hin), ptr(dp), ptr(ga), ptr(nodes), ptr(node_grad), int(saved_hin is not None),
             ptr(gz_parts), stream())
        # weight (+bias, last row) gradient of every layer: [KP x B] . [B x NP], split over the batch
        dwt = torch.empty(RL, DEC_KP, DEC_NP, dtype=torch.float32, device=dev)
        if B > 0:
            ws = torch.empty(_lib.workspace_bytes_f("mdnvae_decoder_wgrad_workspace_floats", B, RL), dtype=torch.float32,
                             device=dev)
            call("mdnvae_decoder_wgrad", ptr(hin), ptr(dp), B, RL, ptr(ws), ptr(dwt), stream())
        else:
            dwt.zero_()
        import ctypes
        n = spec.R * spec.NL
        kind, n_out, k = (ctypes.c_int * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)()
        aux, dw, db = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        # with a gradient sink active and every layer parameter holding a .grad, the unpack adds into
        # those directly and autograd gets no gradients for them
        targets = [_grad_target(p) if p is not None else None for p in ctx.param_objs]
        into = all(t is not None or p is None for t, p in zip(targets, ctx.param_objs))
        grads = []
        for r in range(spec.R):
            for l, ly in enumerate(spec.layers):
                idx = r * spec.NL + l
                shape_w = ctx.param_shapes[2 * idx]
                has_b = ctx.param_shapes[2 * idx + 1] is not None
                if into:
                    gw, gb = targets[2 * idx], targets[2 * idx + 1]
                else:
                    gw = torch.empty(shape_w, dtype=torch.float32, device=dev)
                    gb = torch.empty(shape_w[0], dtype=torch.float32, device=dev) if has_b else None
                kind[idx] = 0 if ly["kind"] == "linear" else 1
                n_out[idx], k[idx] = shape_w
                aux[idx] = ptr(ly.get("mask")) if kind[idx] == 0 else ptr(ly["adj"], "adjacency", torch.int32)
                dw[idx], db[idx] = ptr(gw), ptr(gb)
                grads += [None, None] if into else [gw, gb]
        call("mdnvae_decoder_unpack_layer_grads", n, _addr(kind), ptr(dwt), _addr(aux), _addr(n_out), _addr(k), _addr(dw),
             _addr(db), int(into), stream())
        if into:
            for p in ctx.param_objs:
                if p is not None:
                    _grad_added(p)
        g_iw = g_ib = None
        if icpt_w is not None:
            D = spec.N * spec.N
            ws = torch.empty(_lib.workspace_bytes_f("mdnvae_colsum2_workspace_floats", B, D), dtype=torch.float32, device=dev)
            g_ib = torch.empty(D, dtype=torch.float32, device=dev)
            g_iw = torch.empty(D, dtype=torch.float32, device=dev)
            call("mdnvae_colsum2", ptr(gout), ptr(ga), B, D, ptr(ws), ptr(g_ib), ptr(g_iw), stream())
        return (gz, None, g_iw, g_ib, None, *grads)


def chain_decoder(z, spec, params, icpt_w=None, icpt_b=None, drop=None):
    """params: flat list of (weight, bias) per (channel, layer).
    Returns (out [B,N*N], x_latent [B,N]; empty when the spec asks for none)."""
    return _ChainDecoder.apply(z, spec, icpt_w, icpt_b, drop, *params)


# ---------------------------------------------------------------------------------------
# conv / ResNet encoder
# ---------------------------------------------------------------------------------------
def _conv_out(n, ks, stride, pad):
    return (n + 2 * pad - ks) // stride + 1


class _ConvLayer:
    """Geometry of one encoder convolution and the calls that use it."""

    def __init__(self, cin, cout, ks, stride, pad, reflect, h):
        self.cin, self.cout, self.ks, self.stride, self.pad, self.reflect = cin, cout, ks, stride, pad, reflect
        self.h = h
        self.oh = _conv_out(h, ks, stride, pad)

    def pack(self, w, dev, need_bwd, parts=0):
        """parts = 0: weights for the fp32 matrix instruction; 1..3: bf16 fragments in that many split
        parts (1 = plain bf16, 3 = fp32-accurate) for the bf16 matrix cores, as a uint8 buffer."""
        lib = _lib.load()
        if parts and self.cin > 1:
            wf = torch.empty(lib.mdnvae_conv_packed_bytes_bf16(self.cout, self.cin, self.ks, 0, parts), dtype=torch.uint8,
                             device=dev)
            wb = None
            if need_bwd:
                wb = torch.empty(lib.mdnvae_conv_packed_bytes_bf16(self.cout, self.cin, self.ks, 1, parts), dtype=torch.uint8,
                                 device=dev)
            call("mdnvae_conv_pack_weights_bf16", ptr(_c(w), "conv weight"), self.cout, self.cin, self.ks, parts,
                 wf.data_ptr(), None if wb is None else wb.data_ptr(), stream())
            return wf, wb
        wf = torch.empty(lib.mdnvae_conv_packed_floats(self.cout, self.cin, self.ks, 0), dtype=torch.float32, device=dev)
        wb = None
        if need_bwd and self.cin > 1:
            wb = torch.empty(lib.mdnvae_conv_packed_floats(self.cout, self.cin, self.ks, 1), dtype=torch.float32, device=dev)
        call("mdnvae_conv_pack_weights", ptr(_c(w), "conv weight"), self.cout, self.cin, self.ks, ptr(wf), ptr(wb), stream())
        return wf, wb

    def fwd(self, x, stats, relu_in, wf, bias, B, parts=0):
        """wf from pack() with the same parts."""
        out = torch.empty(B, self.cout, self.oh, self.oh, dtype=torch.float32, device=x.device)
        geom = (B, self.cin, self.h, self.h, self.cout, self.oh, self.oh, self.ks, self.stride, self.pad, int(self.reflect), 0)
        if parts and self.cin > 1:
            call("mdnvae_conv2d_bf16", ptr(x), ptr(stats), int(relu_in), ptr(wf, "packed weights", torch.uint8),
                 ptr(bias, "conv bias"), ptr(out), *geom, parts, stream())
        else:
            call("mdnvae_conv2d", ptr(x), ptr(stats), int(relu_in), ptr(wf), ptr(bias, "conv bias"), ptr(out), *geom, stream())
        return out

    def fwd_stats(self, x, wf, bias, B):
        """The single-channel stem and the instance-norm statistics of its output planes from one kernel."""
        if self.cin != 1 or not self.reflect or self.stride != 1:
            raise RuntimeError("fwd_stats is the stem's product (one input channel, reflection padding, stride 1)")
        out = torch.empty(B, self.cout, self.oh, self.oh, dtype=torch.float32, device=x.device)
        st = torch.empty(B * self.cout, 2, dtype=torch.float32, device=x.device)
        ws = torch.empty(_lib.load().mdnvae_conv_stem_stats_workspace_floats(B, self.cout, self.oh, self.oh), dtype=torch.float32,
                         device=x.device)
        call("mdnvae_conv_stem_stats", ptr(x), ptr(wf), ptr(bias, "conv bias"), ptr(out), B, self.h, self.h, self.cout, self.ks,
             self.pad, ptr(ws), ptr(st), stream())
        return out, st

    def dgrad(self, gout, wb, B, parts=0):
        """Gradient w.r.t. the conv input; for reflection-padded layers on the padded plane."""
        if self.reflect:
            oh, pad = self.h + 2 * self.pad, 0
        else:
            oh, pad = self.h, self.pad
        gin = torch.empty(B, self.cin, oh, oh, dtype=torch.float32, device=gout.device)
        geom = (B, self.cout, self.oh, self.oh, self.cin, oh, oh, self.ks, self.stride, pad, 0, 1)
        if parts:
            call("mdnvae_conv2d_bf16", ptr(gout), None, 0, ptr(wb, "packed weights", torch.uint8), None, ptr(gin), *geom,
                 parts, stream())
        else:
            call("mdnvae_conv2d", ptr(gout), None, 0, ptr(wb), None, ptr(gin), *geom, stream())
        return gin

    def wgrad(self, x, stats, relu_in, gout, B, parts=0):
        lib = _lib.load()
        ws = torch.empty(lib.mdnvae_conv_wgrad_workspace_floats(B, self.cout, self.cin, self.ks), dtype=torch.float32,
                         device=x.device)
        gw = torch.empty(self.cout, self.cin, self.ks, self.ks, dtype=torch.float32, device=x.device)
        split = (parts,) if parts and self.cin > 1 else ()
        call("mdnvae_conv2d_wgrad_bf16" if split else "mdnvae_conv2d_wgrad", ptr(x), ptr(stats),
             int(relu_in), ptr(gout), ptr(ws), ptr(gw), B, self.cin, self.h,
             self.h, self.cout, self.oh, self.oh, self.ks, self.stride, self.pad, int(self.reflect), *split, stream())
        return gw


def _stem_wgrad_bf16(layer, x, gout, B):
    """The stem's weight gradient with both operands rounded to bf16 (the bf16 mode); gout fp32 [B,C,H,W] or bf16
    [B,C,pitch] as the instance-norm backward pass leaves it with omode 3."""
    ws = torch.empty(_lib.load().mdnvae_conv_wgrad_workspace_floats(B, layer.cout, 1, layer.ks), dtype=torch.float32, device=x.device)
    gw = torch.empty(layer.cout, 1, layer.ks, layer.ks, dtype=torch.float32, device=x.device)
    g16 = gout.dtype == torch.bfloat16
    if g16 and gout.shape[-1] != (layer.oh * layer.oh + 7) // 8 * 8:
        raise RuntimeError("bf16 stem gradient planes must be (output plane rounded up to a multiple of 8) entries apart")
    call("mdnvae_conv_stem_wgrad_bf16", ptr(x), ptr(gout, "stem output gradient", gout.dtype), ptr(ws), ptr(gw), B, layer.h, layer.h,
         layer.cout, layer.ks, layer.pad, int(g16), stream())
    return gw


def _plane_stats(x):
    B, C, H, W = x.shape
    st = torch.empty(B * C, 2, dtype=torch.float32, device=x.device)
    call("mdnvae_plane_stats", ptr(x), B * C, H * W, ptr(st), stream())
    return st


def _norm_act_add(raw, stats, skip, relu):
    B, C, H, W = raw.shape
    out = torch.empty_like(raw)
    call("mdnvae_norm_act_add", ptr(raw), ptr(stats), ptr(skip), int(relu), B * C, H * W, ptr(out), stream())
    return out


def _inorm_fwd(raw, skip, relu):
    """(activation, stats): instance norm (+ReLU) (+skip) of raw in one pass."""
    B, C, H, W = raw.shape
    out = torch.empty_like(raw)
    st = torch.empty(B * C, 2, dtype=torch.float32, device=raw.device)
    call("mdnvae_inorm_fwd", ptr(raw), ptr(skip), int(relu), B * C, H * W, ptr(out), ptr(st), stream())
    return out, st


def _inorm_bwd(raw, stats, gact, relu, folded):
    B, C, H, W = raw.shape
    graw = torch.empty_like(raw)
    gsum = torch.empty(B, C, dtype=torch.float32, device=raw.device)
    call("mdnvae_inorm_bwd", ptr(raw), ptr(stats), ptr(gact), int(relu), int(folded), B * C, H, W, ptr(graw),
         ptr(gsum), stream())
    return graw, _colsum(gsum)


def encoder_layers(n_nodes, n_res=5):
    l1 = _ConvLayer(1, 48, 7, 1, 3, True, n_nodes)
    l2 = _ConvLayer(48, 96, 3, 2, 1, False, l1.oh)
    l3 = _ConvLayer(96, 192, 3, 2, 1, False, l2.oh)
    res = [_ConvLayer(192, 192, 3, 1, 1, True, l3.oh) for _ in range(2 * n_res)]
    return [l1, l2, l3] + res


def _mul_parts(parts, layer):
    """How a layer's forward and data-gradient products run under encoder mode `parts`: in the
    fp32-accurate mode (3) the stride-2 layers stay on the fp32 matrix instruction, which is the faster
    of two kernels held to the same bar there (their few-tap data-gradient launches and wide slabs
    leave the bf16 kernel staging-bound); weight gradients use `parts` as given."""
    if layer.cin == 1 or (parts == 3 and layer.stride != 1):
        return 0
    return parts


def _wgrad_parts(parts):
    """Parts of the weight-gradient products under encoder mode `parts`.  A weight gradient goes to the
    optimiser and nowhere else: nothing downstream amplifies its rounding, unlike activations and data
    gradients, which cross up to thirteen instance norms.  So the fp32-accurate mode takes two parts
    there (5e-6 of the largest entry against a bar of 1e-4) for two thirds of the time of three."""
    return 2 if parts == 3 else parts


# A list here receives, from every conv-encoder forward, which entries each ReLU passed (one bool [B,C,H,W] per ReLU
# in network order): what a comparison against a higher-precision run needs to tell arithmetic apart from the side of
# zero a rounding-level normalised value fell on.
RELU_GATE_SINK = None


def _relu_gate(y, stats, H, W, pitch):
    B, C = y.shape[0], y.shape[1]
    v = y.reshape(B, C, -1)[:, :, :H * pitch].reshape(B, C, H, pitch)[..., :W].float()
    return v > stats.view(B, C, 2)[:, :, 0, None, None]


class _ConvEncoder(torch.autograd.Function):
    """x [B, n*n] -> pooled features [B, 192*(h/8)^2]; every arithmetic step is a HIP kernel."""

    @staticmethod
    def forward(ctx, x, n_nodes, parts, *wb):
        x = _c(x)
        B = x.shape[0]
        dev = x.device
        ptr(x, "x")
        layers = encoder_layers(n_nodes, (len(wb) // 2 - 3) // 2)
        need_bwd = any(ctx.needs_input_grad[3:])
        ws = [w for w in wb[0::2]]
        bs = [b for b in wb[1::2]]
        mp = [_mul_parts(parts, ly) for ly in layers]
        packed = [ly.pack(w, dev, need_bwd, m) for ly, w, m in zip(layers, ws, mp)]
        # a normalised activation that only feeds the next convolution is never written: the convolution
        # normalises the raw output of its producer while loading it (statistics from one read of the
        # raw planes); only the residual stream x + block(x) is materialised
        raw1, st1 = layers[0].fwd_stats(x, packed[0][0], bs[0], B)
        raw2 = layers[1].fwd(raw1, st1, 1, packed[1][0], bs[1], B, mp[1])
        st2 = _plane_stats(raw2)
        raw3 = layers[2].fwd(raw2, st2, 1, packed[2][0], bs[2], B, mp[2])
        xk, st3 = _inorm_fwd(raw3, None, 1)
        saved = [x, raw1, st1, raw2, st2, raw3, st3, xk]
        nres = (len(layers) - 3) // 2
        sink = RELU_GATE_SINK
        if sink is not None:
            sink += [_relu_gate(r, s_, ly.oh, ly.oh, ly.oh) for r, s_, ly in ((raw1, st1, layers[0]), (raw2, st2, layers[1]), (raw3, st3, layers[2]))]
        for k in range(nres):
            la, lb = layers[3 + 2 * k], layers[4 + 2 * k]
            ra = la.fwd(xk, None, 0, packed[3 + 2 * k][0], bs[3 + 2 * k], B, mp[3 + 2 * k])
            sa = _plane_stats(ra)
            if sink is not None:
                sink.append(_relu_gate(ra, sa, la.oh, la.oh, la.oh))
            rb = lb.fwd(ra, sa, 1, packed[4 + 2 * k][0], bs[4 + 2 * k], B, mp[4 + 2 * k])
            xk, sb = _inorm_fwd(rb, xk, 0)
            saved += [ra, sa, rb, sb, xk]
        H = layers[-1].oh
        K = 8
        PH = H // K
        feat = torch.empty(B, 192 * PH * PH, dtype=torch.float32, device=dev)
        argmax = torch.empty(B, 192 * PH * PH, dtype=torch.int32, device=dev)
        call("mdnvae_maxpool_fwd", ptr(xk), B * 192, H, H, K, ptr(feat), ptr(argmax, "argmax", torch.int32), stream())
        if need_bwd:
            ctx.save_for_backward(*saved, argmax, *[p[1] for p in packed[1:]])
            ctx.n_nodes, ctx.nres, ctx.parts = n_nodes, nres, parts
        return feat

    @staticmethod
    def backward(ctx, gfeat):
        sv = ctx.saved_tensors
        nres, parts = ctx.nres, ctx.parts
        wparts = _wgrad_parts(parts)
        layers = encoder_layers(ctx.n_nodes, nres)
        x, raw1, st1, raw2, st2, raw3, st3, x0 = sv[:8]
        blocks = [sv[8 + 5 * k: 13 + 5 * k] for k in range(nres)]
        argmax = sv[8 + 5 * nres]
        wbwd = sv[9 + 5 * nres:]          # packed data-gradient weights of layers 1..
        B = x.shape[0]
        H = layers[-1].oh
        gfeat = _c(gfeat)
        G = torch.empty(B, 192, H, H, dtype=torch.float32, device=x.device)
        call("mdnvae_maxpool_bwd", ptr(gfeat), ptr(argmax, "argmax", torch.int32), B * 192, H, H, 8, ptr(G), stream())
        gw = [None] * len(layers)
        gb = [None] * len(layers)
        for k in range(nres - 1, -1, -1):
            ra, sa, rb, sb, _ = blocks[k]
            xin = x0 if k == 0 else blocks[k - 1][4]
            ia, ib = 3 + 2 * k, 4 + 2 * k
            g_rb, gb[ib] = _inorm_bwd(rb, sb, G, 0, 0)
            gw[ib] = layers[ib].wgrad(ra, sa, 1, g_rb, B, wparts)
            gp_a = layers[ib].dgrad(g_rb, wbwd[ib - 1], B, _mul_parts(parts, layers[ib]))
            g_ra, gb[ia] = _inorm_bwd(ra, sa, gp_a, 1, 1)
            gw[ia] = layers[ia].wgrad(xin, None, 0, g_ra, B, wparts)
            gp_x = layers[ia].dgrad(g_ra, wbwd[ia - 1], B, _mul_parts(parts, layers[ia]))
            Gn = torch.empty_like(G)
            call("mdnvae_fold_add", ptr(G), ptr(gp_x), B * 192, H, H, ptr(Gn), stream())
            G = Gn
        g_r3, gb[2] = _inorm_bwd(raw3, st3, G, 1, 0)
        gw[2] = layers[2].wgrad(raw2, st2, 1, g_r3, B, wparts)
        g_a2 = layers[2].dgrad(g_r3, wbwd[1], B, _mul_parts(parts, layers[2]))
        g_r2, gb[1] = _inorm_bwd(raw2, st2, g_a2, 1, 0)
        gw[1] = layers[1].wgrad(raw1, st1, 1, g_r2, B, wparts)
        g_a1 = layers[1].dgrad(g_r2, wbwd[0], B, _mul_parts(parts, layers[1]))
        g_r1, gb[0] = _inorm_bwd(raw1, st1, g_a1, 1, 0)
        gw[0] = layers[0].wgrad(x.view(B, 1, ctx.n_nodes, ctx.n_nodes), None, 0, g_r1, B)
        grads = []
        for w_, b_ in zip(gw, gb):
            grads += [w_, b_]
        return (None, None, None, *grads)


# ---------------------------------------------------------------------------------------
# the same encoder on channel-blocked 16-bit operands (csrc/cbconv.hip)
# ---------------------------------------------------------------------------------------
def _ru(v, m):
    return (v + m - 1) // m * m


class _CBGeom:
    """Grids of the channel-blocked encoder for n_nodes x n_nodes inputs.  A convolution's output rows are
    positions of its INPUT grid (row pitch = that grid's width), Qr of them per sample; PP is the number of
    positions per (group-plane, sample) of a channel-blocked tensor: its own rows x pitch, and far enough
    past the last output row for the largest tap offset."""

    def __init__(self, n):
        self.n = n
        self.h2 = (n + 1) // 2                 # output of the first stride-2 layer
        self.h3 = (self.h2 + 1) // 2           # output of the second one = the residual blocks' planes
        self.p2, self.p3, self.pr = self.h2 + 1, self.h3 + 1, self.h3 + 2
        self.q2, self.q3 = _ru(self.h2 * self.p2, 8), _ru(self.h3 * self.p3, 8)
        self.qr = _ru(self.h3 * self.pr, 8)                  # residual forward rows
        self.qd = _ru((self.h3 + 2) * self.pr, 8)            # residual data-gradient rows (reflection-padded grid)
        s2d = lambda rows, pitch, q: _ru(max((rows + 1) * pitch, q + pitch + 1) + 1, 4)
        self.pp_a1 = s2d(self.h2, self.p2, self.q2)          # parity planes of the first stride-2 layer's input
        self.pp_a2 = s2d(self.h3, self.p3, self.q3)
        self.pp_d2 = self.pp_a1                              # output gradients of the stride-2 layers (zero row / column behind)
        self.pp_d3 = self.pp_a2
        self.pp_r = _ru(max((self.h3 + 2) * self.pr, self.qr + 2 * self.pr + 2) + 1, 4)      # reflection-padded activations
        self.pp_dr = _ru(max((self.h3 + 4) * self.pr, self.qd + 2 * self.pr + 2) + 1, 4)     # zero-bordered gradients
        # weight-gradient position ranges (whole 32-position steps around the output pixels)
        self.k_r = ((2 * self.pr) // 32 * 32, _ru((2 + self.h3) * self.pr, 32))
        self.k_2 = (0, _ru(self.h2 * self.p2, 32))
        self.k_3 = (0, _ru(self.h3 * self.p3, 32))
        if self.qr < 256 or self.q3 < 256:
            raise ValueError(f"n_nodes = {n}: planes of {self.h3} x {self.h3} are smaller than the 256-position tile")


def _cb_alloc(groups, B, PP, dev, parts=1):
    """Channel-blocked tensor [parts][groups][B][PP][8] as bytes (+ slack for clamped reads)."""
    return torch.empty(parts * groups * B * PP * 16 + 4096, dtype=torch.uint8, device=dev), groups * B * PP * 16


def _cb_pack(w, kind, dev, parts=1, amax=None):
    cout, cin = w.shape[0], w.shape[1]
    n = _lib.load().mdnvae_cb_packed_bytes(cout, cin, kind, parts)
    if n == 0:
        raise RuntimeError(f"channel-blocked convolution: unsupported layer {cin} -> {cout} (kind {kind}, parts {parts})")
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    call("mdnvae_cb_pack_weights", ptr(_c(w), "conv weight"), cout, cin, kind, parts, None if amax is None else amax.data_ptr(),
         out.data_ptr(), stream())
    return out


def _cb_pack_many(ws, kinds, dev, parts=1, amaxes=None):
    """_cb_pack of several layers, 16 to a launch."""
    import ctypes
    if len(ws) > 16:
        return _cb_pack_many(ws[:16], kinds[:16], dev, parts, None if amaxes is None else amaxes[:16]) + \
            _cb_pack_many(ws[16:], kinds[16:], dev, parts, None if amaxes is None else amaxes[16:])
    lib = _lib.load()
    n = len(ws)
    ws = [_c(w) for w in ws]
    outs = []
    for w, k in zip(ws, kinds):
        ptr(w, "conv weight")
        nb = lib.mdnvae_cb_packed_bytes(w.shape[0], w.shape[1], k, parts)
        if nb == 0:
            raise RuntimeError(f"channel-blocked convolution: unsupported layer {w.shape[1]} -> {w.shape[0]} (kind {k}, parts {parts})")
        outs.append(torch.empty(nb, dtype=torch.uint8, device=dev))
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    wp, op = P(*[w.data_ptr() for w in ws]), P(*[o.data_ptr() for o in outs])
    ap = P(*[a.data_ptr() for a in amaxes]) if amaxes is not None and amaxes[0] is not None else None
    co, ci, kd = I(*[w.shape[0] for w in ws]), I(*[w.shape[1] for w in ws]), I(*kinds)
    call("mdnvae_cb_pack_weights_batch", n, _addr(wp), _addr(co), _addr(ci), _addr(kd), parts, None if ap is None else _addr(ap),
         _addr(op), stream())
    return outs


def _slot(t):
    return None if t is None else t.data_ptr()


def _cb_conv(act, part_stride, wp, bias, B, cout, cin, kind, rows, pitch, PP, out16=False, parts=1, out_scale=1.0,
             amax_a=None, amax_w=None):
    """out16: the planes leave as 16-bit values: fp16 from the forward kinds (what reads them normalises and rounds to
    bf16 anyway), bf16 from the data-gradient kinds (the pass that reads them rounds its result to bf16); bf16 mode only.
    parts = 2: two fp16 parts per operand; out_scale and the records amax_a / amax_w undo their scales."""
    nout = {0: cout, 1: cout, 2: cin, 3: 4 * cin}[kind]
    code = 0 if not out16 else (1 if kind <= 1 else 2)
    dt = (torch.float32, torch.float16, torch.bfloat16)[code]
    out = torch.empty(B, nout, _ru(rows * pitch, 8), dtype=dt, device=act.device)
    call("mdnvae_cb_conv", act.data_ptr(), part_stride, wp.data_ptr(), ptr(bias, "conv bias"), out.data_ptr(), code, B, cout,
         cin, kind, rows, pitch, PP, parts, float(out_scale), _slot(amax_a), _slot(amax_w), stream())
    return out


def _cb_stats(y, H, W, pitch):
    B, C, Q = y.shape
    st = torch.empty(B * C, 2, dtype=torch.float32, device=y.device)
    call("mdnvae_cb_plane_stats", ptr(y), B * C, H, W, pitch, Q, ptr(st), stream())
    return st


def _cb_norm_fwd(y, stats, H, W, pitch_y, relu, mode, PP, skip=None, want_stream=False, want_act=True, parts=1, scale=1.0):
    """(channel-blocked activation, its part stride, fp32 residual stream [B,C,H,W] if asked, statistics) of a raw conv
    output; stats = None: the pass computes them itself."""
    B, C = y.shape[0], y.shape[1]
    Qy = y.numel() // (B * C)
    dev = y.device
    act, ps = (None, 0)
    if want_act:
        act, ps = _cb_alloc((C // 8) * (4 if mode == 1 else 1), B, PP, dev, parts)
    so = torch.empty(B, C, H, W, dtype=torch.float32, device=dev) if want_stream else None
    compute = stats is None
    if compute:
        stats = torch.empty(B * C, 2, dtype=torch.float32, device=dev)
    y16 = y.dtype == torch.float16
    call("mdnvae_cb_norm_fwd", ptr(y, "raw conv output", y.dtype), int(y16), ptr(stats), int(compute), ptr(skip), ptr(so),
         None if act is None else act.data_ptr(), ps, B, C, H, W, pitch_y, Qy, int(relu), mode, PP, parts, float(scale), stream())
    return act, ps, so, stats


def _cb_norm_bwd(g, gmode, pitch_g, y, stats, pitch_y, relu, C, H, W, omode, PP, base=None, want_sum=False, parts=1,
                 amax_dc=None):
    """(dc, its part stride, plane sums [B,C], base + gradient or None).  parts = 2: amax_dc is a cleared record; a
    probe run of the pass fills it with the largest |dc| first, and the pass proper scales its two fp16 parts by it."""
    B = y.shape[0]
    dev = y.device
    Qy = y.numel() // (B * C)
    Qg = 0 if g is None else g.numel() // (B * (4 * C if gmode == 2 else C))
    if omode == 2:
        dc = torch.empty(B, C, H, W, dtype=torch.float32, device=dev)
        ps = 0
    elif omode == 3:                     # bf16 planes, PP = H * W rounded up to a multiple of 8 apart
        PP = (H * W + 7) // 8 * 8
        dc = torch.empty(B, C, PP, dtype=torch.bfloat16, device=dev)
        ps = 0
    else:
        dc, ps = _cb_alloc(C // 8, B, PP, dev, parts)
    gsum = torch.empty(B, C, dtype=torch.float32, device=dev)
    so = torch.empty(B, C, H, W, dtype=torch.float32, device=dev) if want_sum else None
    y16 = y.dtype == torch.float16
    g16 = 2 if g is not None and g.dtype == torch.bfloat16 else 0
    p_ = 1 if omode >= 2 else parts
    args = (None if g is None else ptr(g, "gradient planes", g.dtype), gmode, pitch_g, Qg, ptr(base), ptr(so),
            ptr(y, "raw conv output", y.dtype), int(y16), ptr(stats), pitch_y, Qy,
            int(relu), B, C, H, W, omode, dc.data_ptr(), ps, PP, p_, 1.0, ptr(gsum))
    if p_ == 2:
        call("mdnvae_cb_norm_bwd", *args, None, amax_dc.data_ptr(), g16, stream())
    call("mdnvae_cb_norm_bwd", *args, _slot(amax_dc) if p_ == 2 else None, None, g16, stream())
    return dc, ps, gsum, so


def _cb_wgrad(g, act, B, cout, cin, planes, PPg, PPa, krange, pitch, parts=1, out_scale=1.0, g_ps=0, a_ps=0, amax_g=None):
    lib = _lib.load()
    ws = torch.empty(lib.mdnvae_cb_wgrad_workspace_floats(B, cout, cin, planes, parts), dtype=torch.float32, device=g.device)
    gw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=g.device)
    call("mdnvae_cb_wgrad", g.data_ptr(), act.data_ptr(), ptr(ws), ptr(gw), B, cout, cin, planes, PPg, PPa, krange[0], krange[1],
         pitch, parts, float(out_scale), g_ps, a_ps, _slot(amax_g), stream())
    return gw


def _pow2_below(v):
    """The largest power of two that keeps v * scale under 2^15 (half of fp16's range)."""
    import math
    return 2.0 ** math.floor(math.log2(32768.0 / max(float(v), 1e-30)))


class _ConvEncoderCB(torch.autograd.Function):
    """The encoder of _ConvEncoder with 16-bit operands in the channel-blocked layout: x [B, n*n] -> pooled features.
    parts = 1: bf16 operands (the reduced-precision mode).  parts = 2: every operand as two fp16 parts, three
    matrix-core products per multiply-add (fp32-accurate; there the 7x7 stem and its weight gradient are the fp32
    kernels of _ConvEncoder, in bf16 mode the stem's weight gradient rounds its operands to bf16): activations carry a fixed power-of-two scale (a normalised
    plane is bounded by sqrt(H W), the residual stream by that times the number of blocks + 1), weights the scale of
    their largest magnitude, gradient operands the scale of theirs, which a probe run of the instance-norm backward
    pass records on the device before the pass proper writes them.  The 7x7 single-channel stem and the max pool are
    the fp32 kernels of _ConvEncoder."""

    @staticmethod
    def forward(ctx, x, n_nodes, parts, *wb):
        x = _c(x)
        B, dev = x.shape[0], x.device
        ptr(x, "x")
        G = _CBGeom(n_nodes)
        nres = (len(wb) // 2 - 3) // 2
        need_bwd = any(ctx.needs_input_grad[3:])
        ws, bs = list(wb[0::2]), list(wb[1::2])
        two = parts == 2
        stem = _ConvLayer(1, 48, 7, 1, 3, True, n_nodes)
        wstem, _ = stem.pack(ws[0], dev, False)
        kinds = [1, 1] + [0] * (2 * nres)
        # records of the largest weight magnitudes (two parts)
        nlayers = 2 + 2 * nres
        rec = torch.zeros(nlayers, ABSMAX_SLOTS, dtype=torch.int32, device=dev) if two else None
        wmax = [None] * nlayers
        if two:
            for i in range(nlayers):
                wmax[i] = rec[i]
                _absmax(_c(ws[1 + i]).reshape(ws[1 + i].shape[0], -1), wmax[i], reset=False)
        wf = _cb_pack_many(ws[1:], kinds, dev, parts, wmax)
        # fixed scales of the activation tensors
        s1 = _pow2_below(n_nodes) if two else 1.0
        s2 = _pow2_below(G.h2) if two else 1.0
        sr = _pow2_below((nres + 1) * G.h3) if two else 1.0
        o16 = not two
        raw1, st1 = stem.fwd_stats(x, wstem, bs[0], B)                                 # [B,48,n,n] and its plane statistics
        a1, p1, _, _ = _cb_norm_fwd(raw1, st1, n_nodes, n_nodes, n_nodes, True, 1, G.pp_a1, parts=parts, scale=s1)
        y2 = _cb_conv(a1, p1, wf[0], bs[1], B, 96, 48, 1, G.h2, G.p2, G.pp_a1, out16=o16, parts=parts, out_scale=1.0 / s1, amax_w=wmax[0])
        a2, p2, _, st2 = _cb_norm_fwd(y2, None, G.h2, G.h2, G.p2, True, 1, G.pp_a2, parts=parts, scale=s2)
        y3 = _cb_conv(a2, p2, wf[1], bs[2], B, 192, 96, 1, G.h3, G.p3, G.pp_a2, out16=o16, parts=parts, out_scale=1.0 / s2, amax_w=wmax[1])
        H = G.h3
        xp, pr, xs, st3 = _cb_norm_fwd(y3, None, H, H, G.p3, True, 0, G.pp_r, want_stream=True, parts=parts, scale=sr)
        saved = [x, raw1, st1, a1, y2, st2, a2, y3, st3]
        sink = RELU_GATE_SINK
        if sink is not None:
            sink += [_relu_gate(raw1, st1, n_nodes, n_nodes, n_nodes), _relu_gate(y2, st2, G.h2, G.h2, G.p2), _relu_gate(y3, st3, H, H, G.p3)]
        for k in range(nres):
            ia, ib = 2 + 2 * k, 3 + 2 * k
            ya = _cb_conv(xp, pr, wf[ia], bs[ia + 1], B, 192, 192, 0, H, G.pr, G.pp_r, out16=o16, parts=parts, out_scale=1.0 / sr, amax_w=wmax[ia])
            aa, _, _, sa = _cb_norm_fwd(ya, None, H, H, G.pr, True, 0, G.pp_r, parts=parts, scale=sr)
            if sink is not None:
                sink.append(_relu_gate(ya, sa, H, H, G.pr))
            yb = _cb_conv(aa, pr, wf[ib], bs[ib + 1], B, 192, 192, 0, H, G.pr, G.pp_r, out16=o16, parts=parts, out_scale=1.0 / sr, amax_w=wmax[ib])
            xp_next, _, xs, sb = _cb_norm_fwd(yb, None, H, H, G.pr, False, 0, G.pp_r, skip=xs, want_stream=True, want_act=k + 1 < nres,
                                              parts=parts, scale=sr)
            saved += [xp, ya, sa, aa, yb, sb]
            xp = xp_next
        K = 8
        PH = H // K
        feat = torch.empty(B, 192 * PH * PH, dtype=torch.float32, device=dev)
        argmax = torch.empty(B, 192 * PH * PH, dtype=torch.int32, device=dev)
        call("mdnvae_maxpool_fwd", ptr(xs), B * 192, H, H, K, ptr(feat), ptr(argmax, "argmax", torch.int32), stream())
        if need_bwd:
            ctx.save_for_backward(*saved, argmax, *ws[1:])
            ctx.geom = (n_nodes, nres, parts, (s1, s2, sr), (p1, p2, pr))
            ctx.wmax = wmax
        return feat

    @staticmethod
    def backward(ctx, gfeat):
        n, nres, parts, (s1, s2, sr), (p1, p2, pr) = ctx.geom
        wmax = ctx.wmax                             # wmax[layer - 1] (layer 0 = stem)
        two = parts == 2
        g16 = not two                               # bf16 mode: data gradients leave their convolutions as bf16 planes
        sv = ctx.saved_tensors
        G = _CBGeom(n)
        x, raw1, st1, a1, y2, st2, a2, y3, st3 = sv[:9]
        blocks = [sv[9 + 6 * k: 15 + 6 * k] for k in range(nres)]
        argmax = sv[9 + 6 * nres]
        ws = sv[10 + 6 * nres:]                     # weights of layers 1.. (the two stride-2 layers, then the blocks)
        B, dev = x.shape[0], x.device
        H = G.h3
        wt = _cb_pack_many(list(ws), [3 if i < 2 else 2 for i in range(len(ws))], dev, parts, wmax)
        gfeat = _c(gfeat)
        Gs = torch.empty(B, 192, H, H, dtype=torch.float32, device=dev)
        call("mdnvae_maxpool_bwd", ptr(gfeat), ptr(argmax, "argmax", torch.int32), B * 192, H, H, 8, ptr(Gs), stream())
        nl = 3 + 2 * nres
        gw, gb = [None] * nl, [None] * nl
        # records of this pass (two parts): the largest magnitude of every gradient operand, left by a probe run of
        # the pass that writes it
        rec = torch.zeros(nl + 1, ABSMAX_SLOTS, dtype=torch.int32, device=dev) if two else None
        nrec = [0]

        def slot():
            if not two:
                return None
            nrec[0] += 1
            return rec[nrec[0] - 1]

        dcb = dmax = pd = None
        for k in range(nres - 1, -1, -1):
            xp, ya, sa, aa, yb, sb = blocks[k]
            La, Lb = 3 + 2 * k, 4 + 2 * k           # layer numbers (0 = stem)
            if dcb is None:
                dmax = slot()
                dcb, pd, gs, _ = _cb_norm_bwd(Gs, 0, H, yb, sb, G.pr, False, 192, H, H, 0, G.pp_dr, parts=parts, amax_dc=dmax)
                gb[Lb] = _colsum(gs)
            gw[Lb] = _cb_wgrad(dcb, aa, B, 192, 192, 1, G.pp_dr, G.pp_r, G.k_r, G.pr, parts, 1.0 / sr, pd, pr, dmax)
            dap = _cb_conv(dcb, pd, wt[Lb - 1], None, B, 192, 192, 2, H + 2, G.pr, G.pp_dr, out16=g16, parts=parts, amax_a=dmax,
                           amax_w=wmax[Lb - 1])
            dmax = slot()
            dca, pd, gs, _ = _cb_norm_bwd(dap, 1, G.pr, ya, sa, G.pr, True, 192, H, H, 0, G.pp_dr, parts=parts, amax_dc=dmax)
            gb[La] = _colsum(gs)
            gw[La] = _cb_wgrad(dca, xp, B, 192, 192, 1, G.pp_dr, G.pp_r, G.k_r, G.pr, parts, 1.0 / sr, pd, pr, dmax)
            dxp = _cb_conv(dca, pd, wt[La - 1], None, B, 192, 192, 2, H + 2, G.pr, G.pp_dr, out16=g16, parts=parts, amax_a=dmax,
                           amax_w=wmax[La - 1])
            dmax = slot()
            if k > 0:
                # the stream's gradient in front of this block, and straight on into the block before it
                ybp, sbp = blocks[k - 1][4], blocks[k - 1][5]
                dcb, pd, gs, Gs = _cb_norm_bwd(dxp, 1, G.pr, ybp, sbp, G.pr, False, 192, H, H, 0, G.pp_dr, base=Gs, want_sum=True,
                                               parts=parts, amax_dc=dmax)
                gb[Lb - 2] = _colsum(gs)
            else:
                dc3, pd, gs, _ = _cb_norm_bwd(dxp, 1, G.pr, y3, st3, G.p3, True, 192, H, H, 1, G.pp_d3, base=Gs, parts=parts,
                                              amax_dc=dmax)
                gb[2] = _colsum(gs)
        gw[2] = _cb_wgrad(dc3, a2, B, 192, 96, 4, G.pp_d3, G.pp_a2, G.k_3, G.p3, parts, 1.0 / s2, pd, p2, dmax)
        da2 = _cb_conv(dc3, pd, wt[1], None, B, 192, 96, 3, G.h3, G.p3, G.pp_d3, out16=g16, parts=parts, amax_a=dmax, amax_w=wmax[1])
        dmax = slot()
        dc2, pd, gs, _ = _cb_norm_bwd(da2, 2, G.p3, y2, st2, G.p2, True, 96, G.h2, G.h2, 1, G.pp_d2, parts=parts, amax_dc=dmax)
        gb[1] = _colsum(gs)
        gw[1] = _cb_wgrad(dc2, a1, B, 96, 48, 4, G.pp_d2, G.pp_a1, G.k_2, G.p2, parts, 1.0 / s1, pd, p1, dmax)
        da1 = _cb_conv(dc2, pd, wt[0], None, B, 96, 48, 3, G.h2, G.p2, G.pp_d2, out16=g16, parts=parts, amax_a=dmax, amax_w=wmax[0])
        g_r1, _, gs, _ = _cb_norm_bwd(da1, 2, G.p2, raw1, st1, n, True, 48, n, n, 2 if two else 3, 0)
        gb[0] = _colsum(gs)
        stem = _ConvLayer(1, 48, 7, 1, 3, True, n)
        gw[0] = stem.wgrad(x.view(B, 1, n, n), None, 0, g_r1, B) if two else _stem_wgrad_bf16(stem, _c(x.view(B, 1, n, n)), g_r1, B)
        grads = []
        for w_, b_ in zip(gw, gb):
            grads += [w_, b_]
        return (None, None, None, *grads)


def conv_encoder(x, n_nodes, weights_and_biases, parts=0, blocked=False):
    """weights_and_biases: [(weight, bias)] for the stem, the two stride-2 convs and the residual
    block convs, in network order.  parts: 0 = the fp32 matrix instruction; 1..3 = the 3x3 convolutions on
    the bf16 matrix cores with operands split into that many bf16 parts (1: operands rounded to bf16,
    3: fp32 accuracy); the stem's forward product, the norms and the pool are fp32 in every mode (the channel-blocked
    bf16 mode rounds the operands of the stem's weight gradient to bf16 like those of every other product).  blocked: activations between
    the 3x3 convolutions in the channel-blocked 16-bit layout (_ConvEncoderCB) instead of fp32 planes; parts is then
    1 (bf16) or 2 (two fp16 parts per operand, fp32-accurate)."""
    flat = []
    for w, b in weights_and_biases:
        flat += [w, b]
    if blocked:
        if int(parts) not in (1, 2):
            raise ValueError("the channel-blocked encoder takes bf16 operands (parts = 1) or two fp16 parts (parts = 2)")
        return _ConvEncoderCB.apply(x, n_nodes, int(parts), *flat)
    return _ConvEncoder.apply(x, n_nodes, int(parts), *flat)


# ---------------------------------------------------------------------------------------
# trait-gated mixture likelihood (build-defined; the reference has no counterpart)
# ---------------------------------------------------------------------------------------
class _MdnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, comp, x, logits, log_sigma, mu, logvar):
        comp, x, logits, log_sigma, mu, logvar = (_c(t) for t in (comp, x, logits, log_sigma, mu, logvar))
        if comp.dim() != 3 or x.dim() != 2 or comp.shape[0] != x.shape[0] or comp.shape[2] != x.shape[1]:
            raise RuntimeError(f"mdn loss: comp {tuple(comp.shape)} must be [B,K,D] for x {tuple(x.shape)}")
        B, K, D = comp.shape
        if logits.shape != (B, K) or log_sigma.shape != (K,):
            raise RuntimeError("mdn loss: gate_logits must be [B,K] and log_sigma [K]")
        L = mu.shape[1]
        dev = comp.device
        out = torch.empty(3, dtype=torch.float32, device=dev)
        resp = torch.empty(B, K, dtype=torch.float32, device=dev)
        ss = torch.empty(B, K, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.load().mdnvae_mdn_workspace_bytes(B), dtype=torch.uint8, device=dev)
        call("mdnvae_mdn_loss_fwd", ptr(comp, "comp_mean"), ptr(x, "x"), ptr(logits, "gate_logits"),
             ptr(log_sigma, "log_sigma"), B, K, D, ptr(mu, "mu"), ptr(logvar, "logvar"), L, None, None, ptr(resp),
             ptr(ss), ws.data_ptr(), ptr(out), stream())
        ctx.save_for_backward(comp, x, logits, log_sigma, mu, logvar, resp, ss)
        return out

    @staticmethod
    def backward(ctx, gout):
        comp, x, logits, log_sigma, mu, logvar, resp, ss = ctx.saved_tensors
        B, K, D = comp.shape
        L = mu.shape[1]
        g2 = torch.stack((gout[0] + gout[1], gout[0] + gout[2])).contiguous()
        gcomp = torch.empty_like(comp)
        glogits = torch.empty_like(logits)
        gls_rows = torch.empty_like(logits)
        gmu = torch.empty_like(mu)
        glv = torch.empty_like(mu)
        ws = torch.empty(_lib.load().mdnvae_mdn_workspace_bytes(B), dtype=torch.uint8, device=comp.device)
        call("mdnvae_mdn_loss_bwd", ptr(comp), ptr(x), ptr(logits), ptr(log_sigma), ptr(resp), ptr(ss), B, K, D,
             ptr(mu), ptr(logvar), L, ptr(g2), ptr(gcomp), ptr(glogits), ptr(gls_rows), ptr(gmu), ptr(glv),
             ws.data_ptr(), stream())
        gls = _colsum(gls_rows) if B > 0 else torch.zeros_like(log_sigma)
        return gcomp, None, glogits, gls, gmu, glv


def mdn_loss_terms(comp_mean, x, gate_logits, log_sigma, mu, logvar):
    """(nll + KL, nll, KL) of the trait-gated Gaussian mixture."""
    out = _MdnLoss.apply(comp_mean, x, gate_logits, log_sigma, mu, logvar)
    return out[0], out[1], out[2]


def mdn_loss(comp_mean, x, gate_logits, log_sigma, mu, logvar):
    return mdn_loss_terms(comp_mean, x, gate_logits, log_sigma, mu, logvar)[0]


def outer_sum_loss_and_grads(nodes, x, mu, logvar, mode, n_nodes, symmetric_x=False):
    """Reconstruction loss + KL for recon = sum_r g_r g_r^T given by node vectors nodes [B,R,NP]
    (chain decoder with nodes_only; no activation or intercept on the sum), and the gradients of
    the total (upstream 1), from one pass over x: (terms[3], g_nodes, g_mu, g_logvar).
    `symmetric_x`: the caller vouches that every x[b] is a symmetric matrix (tri_pack has checked
    this very buffer, say); with input that is not, the node-vector gradient is wrong."""
    nodes, x, mu, logvar = (_c(t.detach()) for t in (nodes, x, mu, logvar))
    if nodes.dim() != 3 or nodes.shape[2] != DEC_NP:
        raise RuntimeError(f"outer-sum loss: nodes {tuple(nodes.shape)} must be [B,R,{DEC_NP}]")
    B, R, _ = nodes.shape
    if x.shape != (B, n_nodes * n_nodes):
        raise RuntimeError(f"outer-sum loss: x {tuple(x.shape)} must be [{B},{n_nodes * n_nodes}]")
    dev = nodes.device
    out = torch.empty(3, dtype=torch.float32, device=dev)
    gnodes = torch.empty_like(nodes)
    gmu, glv = torch.empty_like(mu), torch.empty_like(mu)
    ws = torch.empty(_lib.load().mdnvae_outer_loss_workspace_bytes(B), dtype=torch.uint8, device=dev)
    call("mdnvae_outer_loss_fwdbwd", RECON_MODES[mode], ptr(nodes, "nodes"), ptr(x, "x"), B, R, n_nodes, ptr(mu, "mu"),
         ptr(logvar, "logvar"), mu.shape[1], ptr(gnodes), ptr(gmu), ptr(glv), int(bool(symmetric_x)), ws.data_ptr(),
         ptr(out), stream())
    return out, gnodes, gmu, glv


def _mdn_outer_call(nodes, x, logits, log_sigma, mu, logvar, n_nodes, symmetric_x=False):
    nodes, x, logits, log_sigma, mu, logvar = (_c(t) for t in (nodes, x, logits, log_sigma, mu, logvar))
    if nodes.dim() != 3 or nodes.shape[2] != DEC_NP:
        raise RuntimeError(f"mdn outer loss: nodes {tuple(nodes.shape)} must be [B,K,{DEC_NP}]")
    B, K, _ = nodes.shape
    if x.shape != (B, n_nodes * n_nodes) or logits.shape != (B, K) or log_sigma.shape != (K,):
        raise RuntimeError("mdn outer loss: x must be [B,N*N], gate_logits [B,K], log_sigma [K]")
    dev = nodes.device
    out = torch.empty(3, dtype=torch.float32, device=dev)
    resp = torch.empty(B, K, dtype=torch.float32, device=dev)
    gnodes = torch.empty_like(nodes)
    glogits = torch.empty_like(logits)
    gls_rows = torch.empty_like(logits)
    gmu, glv = torch.empty_like(mu), torch.empty_like(mu)
    ws = torch.empty(_lib.load().mdnvae_mdn_workspace_bytes(B), dtype=torch.uint8, device=dev)
    call("mdnvae_mdn_outer_fwdbwd", ptr(nodes, "nodes"), ptr(x, "x"), ptr(logits, "gate_logits"),
         ptr(log_sigma, "log_sigma"), B, K, n_nodes, ptr(mu, "mu"), ptr(logvar, "logvar"), mu.shape[1], ptr(gnodes),
         ptr(glogits), ptr(gls_rows), ptr(gmu), ptr(glv), ptr(resp), int(bool(symmetric_x)), ws.data_ptr(), None,
         stream())
    # the sweep leaves per-sample terms; one launch adds them up and, next to it, the per-sample log-sigma gradients
    if B > 0:
        gls = torch.empty_like(log_sigma)
        call("mdnvae_mdn_finalize", ws.data_ptr(), B, ptr(out), ptr(gls_rows), K, ptr(gls), stream())
    else:
        gls = torch.zeros_like(log_sigma)
        call("mdnvae_mdn_finalize", ws.data_ptr(), 0, ptr(out), None, K, None, stream())
    return out, gnodes, glogits, gls, gmu, glv, resp


def mdn_outer_loss_and_grads(nodes, x, gate_logits, log_sigma, mu, logvar, n_nodes, symmetric_x=False):
    """The mixture likelihood + KL for component means g_k g_k^T given by their node vectors
    nodes [B,K,NP] (chain decoder with nodes_only), and the gradients of the total (upstream 1),
    from one pass over x: (terms[3], g_nodes, g_logits, g_log_sigma, g_mu, g_logvar).
    `symmetric_x`: the caller vouches that every x[b] is a symmetric matrix (tri_pack has checked
    this very buffer, say); the pass is then a plain stream over x.  With input that is not, the
    node-vector gradient is wrong."""
    with torch.no_grad():
        return _mdn_outer_call(nodes, x, gate_logits, log_sigma, mu, logvar, n_nodes, symmetric_x)[:6]


class _MdnOuterLoss(torch.autograd.Function):
    """Differentiable form of the above: the pass runs in forward, backward scales what it saved."""

    @staticmethod
    def forward(ctx, nodes, x, logits, log_sigma, mu, logvar, n_nodes, symmetric_x):
        out, gnodes, glogits, gls, gmu, glv, _ = _mdn_outer_call(nodes, x, logits, log_sigma, mu, logvar, n_nodes,
                                                                 symmetric_x)
        ctx.save_for_backward(gnodes, glogits, gls, gmu, glv)
        return out

    @staticmethod
    def backward(ctx, gout):
        gnodes, glogits, gls, gmu, glv = ctx.saved_tensors
        g_nll, g_kl = gout[0] + gout[1], gout[0] + gout[2]
        return gnodes * g_nll, None, glogits * g_nll, gls * g_nll, gmu * g_kl, glv * g_kl, None, None


def mdn_outer_loss_terms(nodes, x, gate_logits, log_sigma, mu, logvar, n_nodes, symmetric_x=False):
    out = _MdnOuterLoss.apply(nodes, x, gate_logits, log_sigma, mu, logvar, n_nodes, symmetric_x)
    return out[0], out[1], out[2]


def softmax_rows(logits):
    """softmax over the last dimension of a small [B,K] tensor (evaluation helper; not differentiable)."""
    with torch.no_grad():
        m = logits.max(dim=1, keepdim=True).values
        e = exp(logits - m)
        return e / e.sum(dim=1, keepdim=True)
