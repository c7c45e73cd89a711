"""Adam kernel and whole training steps on the GPU against the CPU oracle."""
import pytest
import torch

from conftest import load_fixture, fixture_params
from oracle import ref_eager as R
import mdnvae_amd as M
from mdnvae_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_adam_kernel_matches_torch_adam():
    g = torch.Generator().manual_seed(2)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** (step - 3))
        ref.grad = grad.clone()
        opt.step()
        ops.call("mdnvae_adam_step", ops.ptr(p), ops.ptr(grad.to(DEV)), ops.ptr(m), ops.ptr(v), n, step, 1e-3, 0.9, 0.999,
                 1e-8, 1.0, ops.stream())
        assert float((p.cpu() - ref.detach()).abs().max()) <= 2e-7 * float(ref.detach().abs().max())


def test_two_training_steps_match_the_oracle():
    """function/MM_Gate.py:195-208 for two steps: same data, same epsilon, Adam lr 1e-3."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    model = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68)
    model.load_state_dict(params)
    model = model.to(DEV)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    x = torch.from_numpy(fix["x"])
    eps = [torch.from_numpy(fix["eps"]), torch.from_numpy(fix["eps"]).flip(0)]
    trainer = M.Trainer(model, "poisson_log")
    got = [trainer.step((x.to(DEV), adjacency), x.to(DEV), eps=e.to(DEV)).item() for e in eps]

    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()}
    want = []
    for t, e in enumerate(eps):
        for v in p.values():
            v.grad = None
        recon, mu, lv, _, _ = R.vae_forward("mm_gate", p, x, e, adjacency=fix["adjacency"])
        loss = R.vae_loss(recon, x, mu, lv, "poisson_log")
        loss.backward()
        want.append(loss.item())
        with torch.no_grad():
            R.adam_step(p, {k: v.grad for k, v in p.items()}, state, t + 1)
    assert abs(got[0] - want[0]) <= 1e-5 * abs(want[0])
    assert abs(got[1] - want[1]) <= 1e-4 * abs(want[1])
    for k, v in model.state_dict().items():
        assert float((v.cpu() - p[k].detach()).abs().max()) <= 5e-4, k     # two lr=1e-3 steps move a weight by <= 2e-3


def test_conv_vae_trainer_reduces_the_loss():
    torch.manual_seed(11111)
    model = M.ConvVAE(decoder="gcn", intercept="diag", final="none", n_nodes=68).to(DEV)
    x = M.data.synthetic_sc(16, 68, seed=1, device=DEV)
    model.set_mask([torch.ones(68, 68)] * 5)
    trainer = M.Trainer(model, "poisson_log")
    gen = torch.Generator(device=DEV).manual_seed(0)
    losses = [trainer.step((x,), x, eps=torch.randn(16, 68, device=DEV, generator=gen)).item() for _ in range(12)]
    assert all(l == l for l in losses)
    assert losses[-1] < losses[0]
