"""Autograd wrappers over the C ABI: each op's forward and backward are hand-written HIP
kernels; nothing here computes on the host or falls back to eager PyTorch arithmetic."""
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
