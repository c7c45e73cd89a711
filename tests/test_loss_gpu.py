"""Fused reparameterisation / KL / reconstruction-loss kernels against the oracle and the
golden loss fixtures.  Tolerances: loss 1e-5 relative, gradients 1e-4 relative (the
parity bar of the build); the asserts below are tighter where fp32 allows."""
import numpy as np
import pytest
import torch

from conftest import load_fixture
from oracle import ref_eager as R
import mdnvae_amd as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOSS_RTOL = 1e-5
GRAD_RTOL = 1e-4


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


MODES = [("poisson_rate", "rate", "x_cnt"), ("poisson_log", "lograte", "x_cnt"), ("mse", "r01", "x01")]


@pytest.mark.parametrize("mode,rkey,xkey", MODES)
def test_loss_against_reference_fixture(mode, rkey, xkey):
    fix = load_fixture("losses")
    r = torch.from_numpy(fix[rkey]).to(DEV).requires_grad_(True)
    x = torch.from_numpy(fix[xkey]).to(DEV)
    mu = torch.from_numpy(fix["mu"]).to(DEV).requires_grad_(True)
    lv = torch.from_numpy(fix["logvar"]).to(DEV).requires_grad_(True)
    loss = M.vae_loss(r, x, mu, lv, mode)
    loss.backward()
    want = float(fix[f"{mode}/loss64"])
    assert abs(loss.item() - want) <= LOSS_RTOL * abs(want)
    assert _rel(r.grad, torch.from_numpy(fix[f"{mode}/grad_recon"])) <= GRAD_RTOL
    assert _rel(mu.grad, torch.from_numpy(fix[f"{mode}/grad_mu"])) <= GRAD_RTOL
    assert _rel(lv.grad, torch.from_numpy(fix[f"{mode}/grad_logvar"])) <= GRAD_RTOL


@pytest.mark.parametrize("mode", ["poisson_rate", "poisson_log", "mse"])
@pytest.mark.parametrize("B,D,L", [(1, 1, 1), (3, 7569, 68), (5, 4624, 68), (2, 5, 3), (0, 4624, 68), (257, 4624, 64)])
def test_loss_ragged_shapes_against_oracle(mode, B, D, L):
    g = torch.Generator().manual_seed(B * 1000 + D)
    x = torch.rand(B, D, generator=g) * 2
    r = torch.rand(B, D, generator=g) * 3 + 0.05 if mode != "poisson_log" else torch.randn(B, D, generator=g)
    mu = torch.randn(B, L, generator=g)
    lv = torch.randn(B, L, generator=g) * 0.7
    rc, mc, lc = (t.clone().double().requires_grad_(True) for t in (r, mu, lv))
    want = R.vae_loss(rc, x.double(), mc, lc, mode)
    want.backward()
    rd, md, ld = (t.to(DEV).requires_grad_(True) for t in (r, mu, lv))
    total, rec, kl = M.vae_loss_terms(rd, x.to(DEV), md, ld, mode)
    (total * 1.5 + rec * 0.25).backward()        # exercises the two gradient scales
    assert abs(total.item() - want.item()) <= LOSS_RTOL * max(abs(want.item()), 1e-6)
    assert abs(kl.item() - R.kl_term(mu.double(), lv.double()).item()) <= LOSS_RTOL * max(abs(kl.item()), 1e-6)
    if B:
        assert _rel(rd.grad, rc.grad * 1.75) <= 1e-5
        assert _rel(md.grad, mc.grad * 1.5) <= 1e-5
        assert _rel(ld.grad, lc.grad * 1.5) <= 1e-5


def test_fused_forward_backward_matches_two_pass():
    g = torch.Generator().manual_seed(3)
    B, D, L = 64, 7569, 68
    r = (torch.randn(B, D, generator=g) * 0.5).to(DEV)
    x = (torch.rand(B, D, generator=g) * 2).to(DEV)
    mu = torch.randn(B, L, generator=g).to(DEV)
    lv = (torch.randn(B, L, generator=g) * 0.5).to(DEV)
    out, gr, gm, gl = M.vae_loss_and_grads(r, x, mu, lv, "poisson_log")
    r2, m2, l2 = (t.clone().requires_grad_(True) for t in (r, mu, lv))
    loss = M.vae_loss(r2, x, m2, l2, "poisson_log")
    loss.backward()
    assert out[0].item() == loss.item()          # same kernel, same reduction order
    assert torch.equal(gr, r2.grad) and torch.equal(gm, m2.grad) and torch.equal(gl, l2.grad)


def test_reparameterize_forward_backward():
    g = torch.Generator().manual_seed(11)
    mu, lv, eps = (torch.randn(37, 68, generator=g) for _ in range(3))
    mc, lc = mu.clone().double().requires_grad_(True), lv.clone().double().requires_grad_(True)
    zc = R.reparameterize(mc, lc, eps.double())
    cot = torch.randn(37, 68, generator=g)
    (zc * cot.double()).sum().backward()
    md, ld = mu.to(DEV).requires_grad_(True), lv.to(DEV).requires_grad_(True)
    z = M.reparameterize(md, ld, eps.to(DEV))
    (z * cot.to(DEV)).sum().backward()
    assert _rel(z, zc) <= 1e-6 and _rel(md.grad, mc.grad) <= 1e-6 and _rel(ld.grad, lc.grad) <= 1e-5


def test_loss_is_bit_reproducible_and_linear_at_full_size():
    # size-independent properties at the bench's full size: run-to-run bit equality, and
    # the loss of a concatenation equals the sum of the parts to fp32 rounding
    g = torch.Generator().manual_seed(5)
    B, D, L = 4096, 7569, 68
    r = torch.randn(B, D, generator=g).mul_(0.3).to(DEV)
    x = torch.rand(B, D, generator=g).to(DEV)
    mu = torch.randn(B, L, generator=g).to(DEV)
    lv = torch.randn(B, L, generator=g).mul_(0.3).to(DEV)
    a = M.vae_loss(r, x, mu, lv, "poisson_log").item()
    b = M.vae_loss(r, x, mu, lv, "poisson_log").item()
    assert a == b
    h = B // 2
    parts = M.vae_loss(r[:h], x[:h], mu[:h], lv[:h], "poisson_log").item() + \
        M.vae_loss(r[h:], x[h:], mu[h:], lv[h:], "poisson_log").item()
    assert abs(a - parts) <= 2e-6 * abs(a)
