"""Trait-gated Gaussian-mixture head.  BUILD-DEFINED model: the reference contains no
mixture-density code, so parity against the reference is UNPINNED here; the oracle is this
build's own float64 statement of the specification (oracle/ref_eager.py: mdn_nll)."""
import numpy as np
import pytest
import torch

from conftest import load_fixture, fixture_params
from oracle import ref_eager as R
import mdnvae_amd as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("B,K,D,L", [(7, 5, 4624, 68), (3, 5, 7569, 68), (1, 1, 10, 2), (4, 8, 36, 5), (0, 5, 16, 4)])
def test_mdn_loss_against_oracle(B, K, D, L):
    g = torch.Generator().manual_seed(B * 31 + K)
    comp = torch.rand(B, K, D, generator=g)
    x = torch.rand(B, D, generator=g)
    logits = torch.randn(B, K, generator=g)
    log_sigma = torch.randn(K, generator=g) * 0.3
    mu = torch.randn(B, L, generator=g)
    lv = torch.randn(B, L, generator=g) * 0.5
    cc, lc, sc, mc, vc = (t.clone().double().requires_grad_(True) for t in (comp, logits, log_sigma, mu, lv))
    want = R.mdn_nll(cc, sc, lc, x.double()) + R.kl_term(mc, vc)
    if B:
        want.backward()
    cd, ld, sd, md, vd = (t.to(DEV).requires_grad_(True) for t in (comp, logits, log_sigma, mu, lv))
    total, nll, kl = M.mdn_loss_terms(cd, x.to(DEV), ld, sd, md, vd)
    assert abs(total.item() - want.item()) <= 1e-5 * max(abs(want.item()), 1e-6)
    if B:
        (total * 1.25 + nll * 0.5).backward()
        assert _rel(cd.grad, cc.grad * 1.75) <= 1e-4
        assert _rel(ld.grad, lc.grad * 1.75) <= 1e-4
        assert _rel(sd.grad, sc.grad * 1.75) <= 1e-4
        assert _rel(md.grad, mc.grad * 1.25) <= 1e-5 and _rel(vd.grad, vc.grad * 1.25) <= 1e-5


def test_mdn_responsibilities_are_a_distribution_and_loss_is_reproducible():
    g = torch.Generator().manual_seed(4)
    B, K, D, L = 2048, 5, 7569, 68
    comp = torch.rand(B, K, D, generator=g).to(DEV)
    x = torch.rand(B, D, generator=g).to(DEV)
    logits = torch.randn(B, K, generator=g).to(DEV)
    ls = torch.zeros(K, device=DEV)
    mu = torch.randn(B, L, generator=g).to(DEV)
    lv = torch.zeros(B, L, device=DEV)
    a = M.mdn_loss(comp, x, logits, ls, mu, lv).item()
    b = M.mdn_loss(comp, x, logits, ls, mu, lv).item()
    assert a == b
    h = B // 2
    parts = M.mdn_loss(comp[:h], x[:h], logits[:h], ls, mu[:h], lv[:h]).item() + \
        M.mdn_loss(comp[h:], x[h:], logits[h:], ls, mu[h:], lv[h:]).item()
    assert abs(a - parts) <= 2e-6 * abs(a)


def test_mdn_gate_vae_end_to_end_against_oracle():
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    torch.manual_seed(5)
    model = M.MDNGateVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, n_traits=4)
    model.load_state_dict(params, strict=False)
    with torch.no_grad():
        model.log_sigma.copy_(torch.tensor([0.1, -0.2, 0.0, 0.3, -0.1]))
    p = {k: v.detach().clone().double().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to(DEV)
    x = torch.from_numpy(fix["x"])
    eps = torch.from_numpy(fix["eps"])
    traits = torch.randn(x.shape[0], 4)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    comp, logits, mu, lv = model(x.to(DEV), adjacency, traits.to(DEV), eps=eps.to(DEV))
    loss = model.loss(comp, logits, x.to(DEV), mu, lv)
    loss.backward()
    # oracle
    mu_c, lv_c = R.fnn_encode(p, x.double())
    z_c = R.reparameterize(mu_c, lv_c, eps.double())
    comp_c = R.gcn_components(p, z_c, fix["adjacency"])
    logit_c = R.mdn_gate_logits(traits.double(), p["gate.weight"], p["gate.bias"])
    want = R.mdn_nll(comp_c, p["log_sigma"], logit_c, x.double()) + R.kl_term(mu_c, lv_c)
    want.backward()
    assert _rel(comp, comp_c) <= 1e-5
    assert abs(loss.item() - want.item()) <= 1e-5 * abs(want.item())
    for name, prm in model.named_parameters():
        assert _rel(prm.grad, p[name].grad) <= 1e-4, name
    mean = model.expected(comp, logits)
    want_mean = (torch.softmax(logit_c, dim=1).unsqueeze(2) * comp_c).sum(dim=1)
    assert _rel(mean, want_mean) <= 1e-5


def _outer_oracle(nodes, x, logits, log_sigma, mu, lv, n):
    """float64 restatement of the rank-one mixture: component k is g_k g_k^T."""
    g = nodes[:, :, :n]
    comp = (g.unsqueeze(3) * g.unsqueeze(2)).reshape(g.shape[0], g.shape[1], n * n)
    return R.mdn_nll(comp, log_sigma, logits, x) + R.kl_term(mu, lv)


@pytest.mark.parametrize("B,K,n,L", [(6, 5, 87, 68), (5, 5, 68, 68), (3, 1, 5, 2), (4, 8, 95, 7), (2, 3, 64, 4), (0, 5, 9, 3)])
def test_rank_one_mixture_from_node_vectors_against_oracle(B, K, n, L):
    """One-pass likelihood + gradients from node vectors against the float64 oracle, for a
    general (not symmetric) target; the padding columns of the node gradient are zero."""
    torch.manual_seed(100 * B + n)
    NP = 96
    nodes = torch.zeros(B, K, NP)
    nodes[:, :, :n] = torch.rand(B, K, n) * 0.8 + 0.1
    x = torch.rand(B, n * n)
    logits = torch.randn(B, K)
    ls = torch.randn(K) * 0.2 + 1.5            # broad components: responsibilities stay mixed
    mu, lv = torch.randn(B, L) * 0.3, torch.randn(B, L) * 0.2
    terms, g_nodes, g_logits, g_ls, g_mu, g_lv = M.ops.mdn_outer_loss_and_grads(
        nodes.to(DEV), x.to(DEV), logits.to(DEV), ls.to(DEV), mu.to(DEV), lv.to(DEV), n)
    if B == 0:
        assert terms.tolist() == [0.0, 0.0, 0.0]
        return
    leaves = [t.double().requires_grad_(True) for t in (nodes, logits, ls, mu, lv)]
    want = _outer_oracle(leaves[0], x.double(), leaves[1], leaves[2], leaves[3], leaves[4], n)
    want.backward()
    assert abs(terms[0].item() - want.item()) <= 1e-5 * abs(want.item())
    assert abs((terms[1] + terms[2]).item() - terms[0].item()) <= 1e-5 * abs(terms[0].item())
    for got, leaf, name in zip((g_nodes, g_logits, g_ls, g_mu, g_lv), leaves, ("nodes", "logits", "log_sigma", "mu", "logvar")):
        assert _rel(got, leaf.grad) <= 1e-4, name
    assert float(g_nodes[:, :, n:].abs().max()) == 0.0


@pytest.mark.parametrize("B,K,n,L", [(6, 5, 87, 68), (5, 5, 68, 68), (3, 1, 5, 2), (4, 8, 95, 7), (2, 3, 64, 4), (0, 5, 9, 3)])
def test_rank_one_mixture_on_symmetric_targets_against_oracle(B, K, n, L):
    """symmetric_x=True (the streaming form, no row sums) against the float64 oracle and against
    the general form on the same symmetric target."""
    torch.manual_seed(7 * B + n)
    nodes = torch.zeros(B, K, 96)
    nodes[:, :, :n] = torch.rand(B, K, n) * 0.8 + 0.1
    x = torch.rand(B, n, n)
    x = (x + x.transpose(1, 2)).reshape(B, n * n).contiguous()
    logits = torch.randn(B, K)
    ls = torch.randn(K) * 0.2 + 1.5
    mu, lv = torch.randn(B, L) * 0.3, torch.randn(B, L) * 0.2
    dev = [t.to(DEV) for t in (nodes, x, logits, ls, mu, lv)]
    sym = M.ops.mdn_outer_loss_and_grads(*dev, n, symmetric_x=True)
    gen = M.ops.mdn_outer_loss_and_grads(*dev, n)
    if B == 0:
        assert sym[0].tolist() == [0.0, 0.0, 0.0]
        return
    leaves = [t.double().requires_grad_(True) for t in (nodes, logits, ls, mu, lv)]
    want = _outer_oracle(leaves[0], x.double(), leaves[1], leaves[2], leaves[3], leaves[4], n)
    want.backward()
    assert abs(sym[0][0].item() - want.item()) <= 1e-5 * abs(want.item())
    # 87 and 68 take the LDS-streamed sweep (sums of squares from |g|^4 - 2 g.xg + |x|^2), the other
    # sizes the row-wise one: the same numbers as the general form up to fp32 rounding either way
    assert _rel(sym[0], gen[0]) <= 2e-6
    for got, other, leaf, name in zip(sym[1:], gen[1:], leaves, ("nodes", "logits", "log_sigma", "mu", "logvar")):
        assert _rel(got, leaf.grad) <= 1e-4, name
        assert _rel(got, other) <= 5e-5, name
    assert float(sym[1][:, :, n:].abs().max()) == 0.0


@pytest.mark.parametrize("B,K,n,L,kind", [(300, 5, 87, 68, "sc"), (257, 8, 68, 128, "rand"), (64, 1, 87, 4, "rand"),
                                           (33, 3, 68, 70, "sc"), (2049, 5, 87, 68, "rand"), (1, 5, 87, 68, "sc"),
                                           (40, 5, 87, 130, "rand")])
def test_streamed_symmetric_sweep_against_oracle(B, K, n, L, kind):
    """The LDS-streamed sweep (n = 87 / 68, latent rows up to 128 wide; wider rows fall back to the
    row-wise sweep) on batches that give waves several samples each and ragged tails, with x starting
    at an odd float of a larger buffer (every sample's run then starts at a different phase), on
    uniform and on heavy-tailed structural-connectome targets."""
    torch.manual_seed(B + n + L)
    nodes = torch.zeros(B, K, 96)
    nodes[:, :, :n] = torch.rand(B, K, n) * 0.8 + 0.1
    if kind == "sc":
        x = M.data.synthetic_sc(B, n, seed=B, device=DEV).cpu()
    else:
        x = torch.rand(B, n, n)
        x = (x + x.transpose(1, 2)).reshape(B, n * n)
    logits = torch.randn(B, K)
    ls = torch.randn(K) * 0.2 + (2.5 if kind == "sc" else 1.5)
    mu, lv = torch.randn(B, L) * 0.3, torch.randn(B, L) * 0.2
    big = torch.empty(B * n * n + 7, device=DEV)
    xd = big[3:3 + B * n * n].view(B, n * n)
    xd.copy_(x)
    dev = [t.to(DEV) for t in (nodes, logits, ls, mu, lv)]
    terms, g_nodes, g_logits, g_ls, g_mu, g_lv = M.ops.mdn_outer_loss_and_grads(dev[0], xd, *dev[1:], n, symmetric_x=True)
    leaves = [t.double().requires_grad_(True) for t in (nodes, logits, ls, mu, lv)]
    want = _outer_oracle(leaves[0], x.double(), leaves[1], leaves[2], leaves[3], leaves[4], n)
    want.backward()
    assert abs(terms[0].item() - want.item()) <= 1e-5 * abs(want.item())
    for got, leaf, name in zip((g_nodes, g_logits, g_ls, g_mu, g_lv), leaves, ("nodes", "logits", "log_sigma", "mu", "logvar")):
        assert _rel(got, leaf.grad) <= 1e-4, name
    assert float(g_nodes[:, :, n:].abs().max()) == 0.0
    # per-sample: every sample's node gradient, not only the largest entries of the batch
    per = (g_nodes.double().cpu() - leaves[0].grad).abs().amax(dim=(1, 2)) / leaves[0].grad.abs().amax(dim=(1, 2)).clamp_min(1e-30)
    assert float(per.max()) <= 2e-4


def test_rank_one_mixture_matches_the_materialised_path_and_scales_with_upstream():
    torch.manual_seed(3)
    B, K, n, L = 7, 5, 68, 68
    nodes = torch.zeros(B, K, 96)
    nodes[:, :, :n] = torch.rand(B, K, n)
    x = torch.rand(B, n * n)
    logits, ls = torch.randn(B, K), torch.full((K,), 1.2)
    mu, lv = torch.randn(B, L) * 0.1, torch.randn(B, L) * 0.1
    nd = nodes.to(DEV).requires_grad_(True)
    lg = logits.to(DEV).requires_grad_(True)
    t = M.ops.mdn_outer_loss_terms(nd, x.to(DEV), lg, ls.to(DEV), mu.to(DEV), lv.to(DEV), n)
    (2.0 * t[1] + 0.5 * t[2]).backward()
    g = nd.detach()[:, :, :n]
    comp = (g.unsqueeze(3) * g.unsqueeze(2)).reshape(B, K, n * n).contiguous().requires_grad_(True)
    lg2 = logits.to(DEV).requires_grad_(True)
    t2 = M.ops.mdn_loss_terms(comp, x.to(DEV), lg2, ls.to(DEV), mu.to(DEV), lv.to(DEV))
    (2.0 * t2[1] + 0.5 * t2[2]).backward()
    for a, b in zip(t, t2):
        assert abs(a.item() - b.item()) <= 2e-6 * abs(b.item())
    assert _rel(lg.grad, lg2.grad) <= 1e-5
    # chain rule through the outer product by hand: dg = (G + G^T) g
    G = comp.grad.reshape(B, K, n, n)
    want = torch.einsum("bkij,bkj->bki", G + G.transpose(2, 3), g)
    assert _rel(nd.grad[:, :, :n], want) <= 1e-5


def test_mdn_trainer_fused_and_materialised_steps_agree():
    """Trainer(loss_mode='mdn'): the fused step (node vectors -> one-pass likelihood) and the step
    through the materialised component matrices give the same loss and the same gradients."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    x = torch.from_numpy(fix["x"]).to(DEV)
    eps = torch.from_numpy(fix["eps"]).to(DEV)
    traits = torch.randn(x.shape[0], 4, generator=torch.Generator().manual_seed(1)).to(DEV)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    got = {}
    for fused in (False, True):
        torch.manual_seed(5)
        model = M.MDNGateVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, n_traits=4)
        model.load_state_dict(params, strict=False)
        trainer = M.Trainer(model.to(DEV), "mdn", fused_loss=fused)
        loss = trainer.forward_backward((x, adjacency, traits), x, eps=eps)
        got[fused] = (loss.item(), trainer.buf.flat_g.clone())
    assert abs(got[True][0] - got[False][0]) <= 1e-5 * abs(got[False][0])
    assert _rel(got[True][1], got[False][1]) <= 1e-4


def test_mdn_trainer_streams_a_target_that_is_the_verified_symmetric_input(monkeypatch):
    """symmetric_input=True and target = the input buffer: the step takes the streaming likelihood
    (the encoder's packing pass has checked that buffer) and gives the gradients of the general
    step; a separate target tensor keeps the general form."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    x = torch.from_numpy(fix["x"]).reshape(-1, 68, 68)
    x = (0.5 * (x + x.transpose(1, 2))).reshape(-1, 68 * 68).contiguous().to(DEV)
    eps = torch.from_numpy(fix["eps"]).to(DEV)
    traits = torch.randn(x.shape[0], 4, generator=torch.Generator().manual_seed(1)).to(DEV)
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    seen = []
    inner = M.ops._mdn_outer_call
    monkeypatch.setattr(M.ops, "_mdn_outer_call", lambda *a: (seen.append(a[-1]), inner(*a))[1])
    got = []
    for target in (x, x.clone()):
        torch.manual_seed(5)
        model = M.MDNGateVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=True)
        model.load_state_dict(params, strict=False)
        trainer = M.Trainer(model.to(DEV), "mdn")
        loss = trainer.forward_backward((x, adjacency, traits), target, eps=eps)
        model.check_inputs()
        got.append((loss.item(), trainer.buf.flat_g.clone()))
    assert seen == [True, False]
    assert abs(got[0][0] - got[1][0]) <= 1e-6 * abs(got[1][0])
    assert _rel(got[0][1], got[1][1]) <= 1e-5


def test_mixture_model_on_the_reference_traits_and_subjects():
    """The trait-gated mixture model on the reference's own data: the 100 structural connectomes of
    data/HCP_samples.mat conditioned on data/trait.pt (tests/golden/traits.npz).  The mixture head is
    build-defined (the reference never loads trait.pt), so the expected values are the build's own
    float64 statement: loss 1e-5, gradients 1e-4, for the plain and the symmetric-input model; then
    one epoch of training lowers the loss."""
    fix = load_fixture("traits")
    params = fixture_params(load_fixture("mm_gate"))
    x = torch.from_numpy(fix["x"])
    traits = torch.from_numpy(fix["traits"])
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    eps = torch.randn(x.shape[0], 68, generator=torch.Generator().manual_seed(11))
    want = {}
    for sym in (False, True):
        torch.manual_seed(5)
        model = M.MDNGateVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=sym)
        model.load_state_dict(params, strict=False)
        with torch.no_grad():
            model.log_sigma.fill_(3.0)           # broad components: responsibilities stay mixed on these heavy-tailed targets
        if not want:                                  # float64 statement of the same model
            p = {k: v.detach().double().clone().requires_grad_(True) for k, v in model.state_dict().items()
                 if not k.startswith("_")}
            mu, lv = R.fnn_encode(p, x.double())
            z = R.reparameterize(mu, lv, eps.double())
            comp = R.gcn_components(p, z, adjacency)
            logits = R.mdn_gate_logits(traits.double(), p["gate.weight"], p["gate.bias"])
            total = R.mdn_nll(comp, p["log_sigma"], logits, x.double()) + R.kl_term(mu, lv)
            total.backward()
            want = {"loss": total.item(), "grads": {k: v.grad for k, v in p.items()}}
        trainer = M.Trainer(model.to(DEV), "mdn", folded_adam=False)
        xd = x.to(DEV)
        loss = trainer.forward_backward((xd, adjacency, traits.to(DEV)), xd, eps=eps.to(DEV))
        model.check_inputs()
        assert abs(loss.item() - want["loss"]) <= 1e-5 * abs(want["loss"])
        for name, prm in model.named_parameters():
            assert _rel(prm.grad, want["grads"][name]) <= 1e-4, (sym, name)
    # the trait gate matters on this data and training moves: an epoch lowers the loss
    torch.manual_seed(5)
    model = M.MDNGateVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, n_traits=4, symmetric_input=True)
    trainer = M.Trainer(model.to(DEV), "mdn")
    gen = torch.Generator(device=DEV).manual_seed(1)
    tr = traits.to(DEV)
    first = trainer.train_epoch(x.to(DEV), batch_size=32, extra=lambda idx: (adjacency, tr.index_select(0, idx)),
                                generator=gen, eps_generator=torch.Generator(device=DEV).manual_seed(2))
    for _ in range(3):
        last = trainer.train_epoch(x.to(DEV), batch_size=32, extra=lambda idx: (adjacency, tr.index_select(0, idx)),
                                   generator=gen, eps_generator=torch.Generator(device=DEV).manual_seed(2))
    assert last < first
