"""Chain decoder (rank-1 / masked / k-NN GCN), dense layers, GCN-VAE and cVAE end to end on
the GPU, against the reference-generated fixtures and the CPU oracle.
Bars: outputs 1e-5, loss 1e-5, gradients 1e-4 (relative)."""
import numpy as np
import pytest
import torch

import recipe
from conftest import load_fixture, fixture_params
from oracle import ref_eager as R
import mdnvae_amd as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OUT_RTOL, LOSS_RTOL, GRAD_RTOL = 1e-5, 1e-5, 1e-4


def _rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _load(model, params):
    missing = model.load_state_dict(params, strict=False)
    assert not missing.missing_keys, missing.missing_keys
    return model.to(DEV)


def _check_named_grads(fix, model, names=None, prefix_skip=()):
    for name, p in model.named_parameters():
        if names is not None and name not in names:
            continue
        if any(name.startswith(s) for s in prefix_skip):
            continue
        assert p.grad is not None, name
        g = p.grad
        if name == "fcintercept.weight" and g.dim() == 1 and f"gsamp/{name}" in fix.files:
            full = torch.zeros(g.numel(), g.numel())
            full.diagonal().copy_(g.cpu())
            g = full
        recipe.check_grad(fix, name, g, GRAD_RTOL, "hip")


def test_gcn_vae_against_reference_fixture():
    fix = load_fixture("mm_gate")
    model = _load(M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68), fixture_params(fix))
    x = torch.from_numpy(fix["x"]).to(DEV)
    eps = torch.from_numpy(fix["eps"]).to(DEV)
    adjacency = [list(map(int, row)) for row in fix["adjacency"]]
    recon, mu, logvar = model(x, adjacency, eps=eps)
    loss = M.loss_function_poisson_log(recon, x, mu, logvar)
    loss.backward()
    assert _rel(mu, fix["mu"]) <= OUT_RTOL and _rel(logvar, fix["logvar"]) <= OUT_RTOL
    assert _rel(recon, fix["recon"]) <= OUT_RTOL
    assert abs(loss.item() - float(fix["loss"])) <= LOSS_RTOL * abs(float(fix["loss"]))
    _check_named_grads(fix, model)


def test_cvae_hinge_step_against_reference_fixture():
    fix = load_fixture("cvae")
    model = _load(M.cVAE(in_dim=68 * 68, latent_dim=64), fixture_params(fix))
    fc = torch.from_numpy(fix["fc"]).to(DEV)
    sc = torch.from_numpy(fix["sc"]).to(DEV)
    fc_no = fc.clone()
    fc_no[torch.from_numpy(fix["mask_idx"]).to(DEV)] = 0.0
    recon, mu, logvar = model(fc, sc, eps=torch.from_numpy(fix["eps_a"]).to(DEV))
    with_fc, rl, kl = M.loss_function_mse_terms(recon, sc, mu, logvar)
    recon_no, mu_no, lv_no = model(fc_no, sc, eps=torch.from_numpy(fix["eps_b"]).to(DEV))
    without, _, _ = M.loss_function_mse_terms(recon_no, sc, mu_no, lv_no)
    loss = with_fc + torch.relu(with_fc - without + 0.01)       # function_FC_SC/SC-FC.py:156-157
    loss.backward()
    assert _rel(recon, fix["recon"]) <= OUT_RTOL and _rel(recon_no, fix["recon_no"]) <= OUT_RTOL
    for got, key in ((loss, "loss"), (with_fc, "loss_with"), (without, "loss_no"), (rl, "recon_loss"), (kl, "kld")):
        assert abs(got.item() - float(fix[key])) <= LOSS_RTOL * abs(float(fix[key])), key
    _check_named_grads(fix, model)


VARIANT_CTOR = {
    "cnn_nn_sc": dict(decoder="nn", intercept="dense", final="exp"),
    "cnn_gcn_sc": dict(decoder="gcn", intercept="diag", final="none"),
    "cnn_gcn_fc": dict(decoder="gcn", intercept="diag", final="none"),
    "cnn_nn_fc": dict(decoder="nn", intercept="none", final="none", dropout=0.5),
}
DEC_NAMES = [f"{n}.{s}" for n in ("fc3", "fc32", "fc33", "fc34", "fc35", "fc4", "fc5", "fc6", "fc7", "fc8", "fcintercept")
             for s in ("weight", "bias")]


@pytest.mark.parametrize("variant", list(VARIANT_CTOR))
def test_rank1_decoder_against_reference_fixture(variant):
    """decode(z) with the fixture's z: recon, x_latent and every decoder gradient are fixed by
    z alone, so they must match what the reference produced for the whole model."""
    fix = load_fixture(variant)
    model = M.ConvVAE(**VARIANT_CTOR[variant]).eval()
    model = _load(model, fixture_params(fix))
    if "mask" in fix.files:
        model.set_mask([torch.from_numpy(fix["mask"])] * 5 + [torch.eye(68 * 68)])
    z = torch.from_numpy(fix["z"]).to(DEV)
    x = torch.from_numpy(fix["x"]).to(DEV)
    mu = torch.from_numpy(fix["mu"]).to(DEV)
    lv = torch.from_numpy(fix["logvar"]).to(DEV)
    recon, x_latent = model.decode(z)
    loss = M.vae_loss(recon, x, mu, lv, R.loss_mode(variant))
    loss.backward()
    assert _rel(recon, fix["recon"]) <= OUT_RTOL
    assert _rel(x_latent, fix["x_latent"]) <= OUT_RTOL
    assert abs(loss.item() - float(fix["loss"])) <= LOSS_RTOL * abs(float(fix["loss"]))
    names = [n for n in DEC_NAMES if f"gnorm/{n}" in fix.files]
    _check_named_grads(fix, model, names=names)


def _rank1_model(N, L, act, g):
    """The rank-1 decoder side of ConvVAE for any node count (the encoder is not used here)."""
    model = M.ConvVAE.__new__(M.ConvVAE)
    torch.nn.Module.__init__(model)
    model.n_nodes, model.latent_dim, model.act = N, L, act
    model.decoder_kind, model.intercept_kind, model.final, model.dropout_p, model._spec = "gcn", "diag", "none", 0.0, None
    for name in ("fc3", "fc32", "fc33", "fc34", "fc35"):
        setattr(model, name, M.Linear(L, N))
    for name in ("fc4", "fc5", "fc6", "fc7", "fc8"):
        setattr(model, name, M.GraphCNN(N, N))
    model.fcintercept = M.DiagonalIntercept(N * N)
    with torch.no_grad():
        model.fcintercept.weight.uniform_(-1, 1, generator=g)
    mask = (torch.rand(N, N, generator=g) < 0.4).float()
    mask.fill_diagonal_(1.0)
    model.set_mask([mask] * 5)
    return model.to(DEV), mask


@pytest.mark.parametrize("N,L,B,act,use_drop", [(87, 68, 33, "sigmoid", False), (87, 87, 1, "tanh", True),
                                                (68, 68, 17, "sigmoid", True), (20, 7, 40, "tanh", False)])
def test_rank1_decoder_shapes_against_oracle(N, L, B, act, use_drop):
    g = torch.Generator().manual_seed(N * 100 + B)
    model, mask = _rank1_model(N, L, act, g)
    z = torch.randn(B, L, generator=g)
    cot = torch.randn(B, N * N, generator=g)
    cotx = torch.randn(B, N, generator=g)
    drop = (torch.bernoulli(torch.full((5, B, N), 0.5), generator=g) * 2.0) if use_drop else None
    # oracle in float64
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    zc = z.double().requires_grad_(True)
    aref = torch.sigmoid if act == "sigmoid" else torch.tanh
    recon_c, xl_c = R.rank1_decode(p, zc, mask=mask.double(), intercept="diagvec", final="none", act=aref,
                                   drop_masks=None if drop is None else list(drop.double()), n_nodes=N)
    ((recon_c * cot.double()).sum() + (xl_c * cotx.double()).sum()).backward()
    # HIP path
    zd = z.to(DEV).requires_grad_(True)
    spec = model._chain_spec()
    params = []
    for a, b in model._names:
        params += [getattr(model, a).weight, getattr(model, a).bias, getattr(model, b).weight, getattr(model, b).bias]
    recon, xl = M.ops.chain_decoder(zd, spec, params, model.fcintercept.weight, model.fcintercept.bias,
                                    None if drop is None else drop.to(DEV).contiguous())
    ((recon * cot.to(DEV)).sum() + (xl * cotx.to(DEV)).sum()).backward()
    assert _rel(recon, recon_c) <= OUT_RTOL and _rel(xl, xl_c) <= OUT_RTOL
    assert _rel(zd.grad, zc.grad) <= GRAD_RTOL
    for name, prm in model.named_parameters():
        assert _rel(prm.grad, p[name].grad) <= GRAD_RTOL, name


@pytest.mark.parametrize("N,k,B", [(87, 32, 19), (68, 32, 1), (30, 5, 64)])
def test_gcn_decoder_shapes_against_oracle(N, k, B):
    g = torch.Generator().manual_seed(N + B)
    L = 68
    dec = M.MultiChannelGCNDecoder(R=5, n_nodes=N, k=k, n_layers=3, latent_dim=L)
    with torch.no_grad():
        for prm in dec.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) * 0.3)
    A = torch.rand(N, N, generator=g).numpy()
    adj = R.knn_adjacency(A + A.T, k)
    dec = dec.to(DEV)
    z = torch.randn(B, L, generator=g)
    cot = torch.randn(B, N * N, generator=g)
    p = {"decoder." + kk: v.detach().cpu().double().requires_grad_(True) for kk, v in dec.state_dict().items()}
    zc = z.double().requires_grad_(True)
    out_c = R.gcn_decode(p, zc, adj)
    (out_c * cot.double()).sum().backward()
    zd = z.to(DEV).requires_grad_(True)
    out = dec(zd, [list(map(int, r)) for r in adj])
    (out * cot.to(DEV)).sum().backward()
    assert _rel(out, out_c) <= OUT_RTOL
    assert _rel(zd.grad, zc.grad) <= GRAD_RTOL
    for name, prm in dec.named_parameters():
        assert _rel(prm.grad, p["decoder." + name].grad) <= GRAD_RTOL, name


def test_decoder_rejects_oversize_and_bad_adjacency():
    dec = M.MultiChannelGCNDecoder(R=5, n_nodes=68, k=32).to(DEV)
    z = torch.zeros(2, 68, device=DEV)
    with pytest.raises(ValueError):
        dec(z, [[0] * 31] * 68)
    with pytest.raises(ValueError):
        dec(z, [[99] * 32] * 68)
    big = M.MultiChannelGCNDecoder(R=5, n_nodes=120, k=8, latent_dim=68).to(DEV)
    with pytest.raises(RuntimeError, match="panel|n_nodes"):
        big(z, [[0] * 8] * 120)


@pytest.mark.parametrize("mode", ["poisson_log", "mse", "poisson_rate"])
@pytest.mark.parametrize("B,Rch,n,L", [(6, 5, 87, 68), (5, 5, 68, 68), (3, 1, 5, 2), (4, 8, 95, 7), (0, 5, 9, 3)])
def test_outer_sum_loss_from_node_vectors_against_oracle(mode, B, Rch, n, L):
    """loss_function on recon = sum_r g_r g_r^T taken straight from the node vectors: value and
    gradients against the float64 oracle (general, not symmetric, targets)."""
    torch.manual_seed(10 * B + n)
    nodes = torch.zeros(B, Rch, 96)
    nodes[:, :, :n] = torch.rand(B, Rch, n) * 0.5 + 0.1
    x = torch.rand(B, n * n) * 2.0
    mu, lv = torch.randn(B, L) * 0.3, torch.randn(B, L) * 0.2
    terms, g_nodes, g_mu, g_lv = M.ops.outer_sum_loss_and_grads(nodes.to(DEV), x.to(DEV), mu.to(DEV), lv.to(DEV), mode, n)
    if B == 0:
        assert terms.tolist() == [0.0, 0.0, 0.0]
        return
    leaves = [t.double().requires_grad_(True) for t in (nodes, mu, lv)]
    g = leaves[0][:, :, :n]
    recon = (g.unsqueeze(3) * g.unsqueeze(2)).sum(dim=1).reshape(B, n * n)
    want = R.vae_loss(recon, x.double(), leaves[1], leaves[2], mode)
    want.backward()
    assert abs(terms[0].item() - want.item()) <= 1e-5 * abs(want.item())
    for got, leaf, name in zip((g_nodes, g_mu, g_lv), leaves, ("nodes", "mu", "logvar")):
        assert _rel(got, leaf.grad) <= 1e-4, name
    assert float(g_nodes[:, :, n:].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["poisson_log", "mse", "poisson_rate"])
@pytest.mark.parametrize("B,Rch,n,L", [(6, 5, 87, 68), (5, 5, 68, 68), (3, 1, 5, 2), (4, 8, 95, 7), (2, 3, 64, 4), (0, 5, 9, 3)])
def test_outer_sum_loss_on_symmetric_targets_against_oracle(mode, B, Rch, n, L):
    """symmetric_x=True (no row sums) against the float64 oracle and against the general form on
    the same symmetric target."""
    torch.manual_seed(3 * B + n)
    nodes = torch.zeros(B, Rch, 96)
    nodes[:, :, :n] = torch.rand(B, Rch, n) * 0.5 + 0.1
    x = torch.rand(B, n, n)
    x = (x + x.transpose(1, 2)).reshape(B, n * n).contiguous()
    mu, lv = torch.randn(B, L) * 0.3, torch.randn(B, L) * 0.2
    dev = [t.to(DEV) for t in (nodes, x, mu, lv)]
    sym = M.ops.outer_sum_loss_and_grads(*dev, mode, n, symmetric_x=True)
    gen = M.ops.outer_sum_loss_and_grads(*dev, mode, n)
    if B == 0:
        assert sym[0].tolist() == [0.0, 0.0, 0.0]
        return
    leaves = [t.double().requires_grad_(True) for t in (nodes, mu, lv)]
    g = leaves[0][:, :, :n]
    recon = (g.unsqueeze(3) * g.unsqueeze(2)).sum(dim=1).reshape(B, n * n)
    want = R.vae_loss(recon, x.double(), leaves[1], leaves[2], mode)
    want.backward()
    assert abs(sym[0][0].item() - want.item()) <= 1e-5 * abs(want.item())
    assert abs(sym[0][0].item() - gen[0][0].item()) <= 1e-6 * abs(want.item())
    for got, other, leaf, name in zip(sym[1:], gen[1:], leaves, ("nodes", "mu", "logvar")):
        assert _rel(got, leaf.grad) <= 1e-4, name
        assert _rel(got, other) <= 1e-5, name
    assert float(sym[1][:, :, n:].abs().max()) == 0.0


def test_mm_gate_trainer_streams_a_target_that_is_the_verified_symmetric_input(monkeypatch):
    """symmetric_input=True and target = the input buffer: the fused step takes the form without
    row sums; a separate target tensor keeps the general form; both give the same gradients."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    x = torch.from_numpy(fix["x"]).reshape(-1, 68, 68)
    x = (0.5 * (x + x.transpose(1, 2))).reshape(-1, 68 * 68).contiguous().to(DEV)
    eps = torch.from_numpy(fix["eps"]).to(DEV)
    adjacency = [list(map(int, row)) for row in fix["adjacency"]]
    seen = []
    inner = M.ops.outer_sum_loss_and_grads
    monkeypatch.setattr(M.ops, "outer_sum_loss_and_grads",
                        lambda *a, **kw: (seen.append(kw.get("symmetric_x")), inner(*a, **kw))[1])
    got = []
    for target in (x, x.clone()):
        model = _load(M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68, symmetric_input=True), params)
        trainer = M.Trainer(model, "poisson_log")
        loss = trainer.forward_backward((x, adjacency), target, eps=eps)
        model.check_inputs()
        got.append((loss.item(), trainer.buf.flat_g.clone()))
    assert seen == [True, False]
    assert abs(got[0][0] - got[1][0]) <= 1e-6 * abs(got[1][0])
    assert _rel(got[0][1], got[1][1]) <= 1e-5


def test_mm_gate_trainer_fused_and_plain_steps_agree():
    """Trainer on the GCN-decoder VAE of function/MM_Gate.py: the step that takes loss and gradients
    from the node vectors and the step through the materialised reconstruction agree, and both match
    the reference fixture's loss."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    x = torch.from_numpy(fix["x"]).to(DEV)
    eps = torch.from_numpy(fix["eps"]).to(DEV)
    adjacency = [list(map(int, row)) for row in fix["adjacency"]]
    got = {}
    for fused in (False, True):
        model = _load(M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68), params)
        trainer = M.Trainer(model, "poisson_log", fused_loss=fused)
        loss = trainer.forward_backward((x, adjacency), x, eps=eps)
        got[fused] = (loss.item(), trainer.buf.flat_g.clone())
        assert abs(loss.item() - float(fix["loss"])) <= LOSS_RTOL * abs(float(fix["loss"]))
        _check_named_grads(fix, model)
    assert _rel(got[True][1], got[False][1]) <= 1e-4


@pytest.mark.parametrize("n_nodes", [68, 87])
def test_large_batch_chain_forward_matches_the_small_batch_kernel(n_nodes):
    """From 8192 samples up the chain decoder's forward runs as one wave per 16 samples (no barrier between
    layers); below, as 32-sample tiles with the columns split over three waves.  The same decoder on the same
    latent rows must give the same node vectors through both - and the same gradients, which the backward
    pass forms from the layer inputs the forward saved."""
    torch.manual_seed(n_nodes)
    B = 8192 + 37
    dec = M.MultiChannelGCNDecoder(R=5, n_nodes=n_nodes, k=32, n_layers=3, latent_dim=68).to(DEV)
    adjacency = [[(u + 3 * j) % n_nodes for j in range(32)] for u in range(n_nodes)]
    z = torch.randn(B, 68, device=DEV)
    cot = torch.randn(B, 5, 96, device=DEV)

    def run(lo, hi):
        zz = z[lo:hi].clone().requires_grad_(True)
        for p in dec.parameters():
            p.grad = None
        nodes = dec.node_vectors(zz, adjacency)
        (nodes * cot[lo:hi]).sum().backward()
        return nodes.detach(), zz.grad.detach(), [p.grad.detach().clone() for p in dec.parameters()]

    half = B // 2
    big, lo, hi = run(0, B), run(0, half), run(half, B)
    assert _rel(big[0], torch.cat([lo[0], hi[0]])) <= OUT_RTOL
    assert _rel(big[1], torch.cat([lo[1], hi[1]])) <= GRAD_RTOL
    for gb, gl, gh in zip(big[2], lo[2], hi[2]):          # parameter gradients: the halves add up to the whole
        assert _rel(gb, gl + gh) <= GRAD_RTOL


@pytest.mark.parametrize("N,L,act", [(87, 68, "tanh"), (30, 7, "sigmoid")])
def test_large_batch_rank1_decoder_matches_the_small_batch_kernels(N, L, act):
    """The two-layer chains with x_latent (the latent gradient then has a second source) through the
    large-batch kernels, forward and backward, against the same rows in two halves through the others."""
    g = torch.Generator().manual_seed(N)
    model, _ = _rank1_model(N, L, act, g)
    B = 8192 + 21
    z = torch.randn(B, L, generator=g).to(DEV)
    cot = (torch.randn(B, N * N, generator=g) * 0.1).to(DEV)
    cotx = torch.randn(B, N, generator=g).to(DEV)
    spec = model._chain_spec()
    params = []
    for a, b in model._names:
        params += [getattr(model, a).weight, getattr(model, a).bias, getattr(model, b).weight, getattr(model, b).bias]

    def run(lo, hi):
        zz = z[lo:hi].clone().requires_grad_(True)
        for p in model.parameters():
            p.grad = None
        recon, xl = M.ops.chain_decoder(zz, spec, params, model.fcintercept.weight, model.fcintercept.bias, None)
        ((recon * cot[lo:hi]).sum() + (xl * cotx[lo:hi]).sum()).backward()
        return recon.detach(), xl.detach(), zz.grad.detach(), [p.grad.detach().clone() for p in model.parameters()]

    half = B // 2
    big, lo, hi = run(0, B), run(0, half), run(half, B)
    for i, bar in ((0, OUT_RTOL), (1, OUT_RTOL), (2, GRAD_RTOL)):
        assert _rel(big[i], torch.cat([lo[i], hi[i]])) <= bar, i
    for gb, gl, gh in zip(big[3], lo[3], hi[3]):
        assert _rel(gb, gl + gh) <= GRAD_RTOL
