"""The symmetric-input first layer (packed triangle, folded weights) against the plain layer and the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_fixture, fixture_params
from oracle import ref_eager as R
import mdnvae_amd as M
from mdnvae_amd import ops, _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _tri_pairs(n):
    iu = np.triu_indices(n)           # row-major walk of the upper triangle
    return iu[0], iu[1]


@pytest.mark.parametrize("n,rows", [(87, 5), (68, 3), (1, 2), (2, 4), (130, 2), (16, 0)])
def test_pack_fold_unfold_against_numpy(n, rows):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((rows, n, n)).astype(np.float32)
    sym = a + a.transpose(0, 2, 1)
    i, j = _tri_pairs(n)
    T = n * (n + 1) // 2
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    packed = ops.tri_pack(torch.from_numpy(sym).to(DEV), n, flag)
    ld = ops.tri_row_stride(n)
    assert ld % 64 == 0 and 0 <= ld - T < 64 and packed.shape == (rows, ld)
    assert np.array_equal(packed.cpu().numpy()[:, :T], sym[:, i, j])
    assert not packed.cpu().numpy()[:, T:].any()
    assert int(flag.item()) == 0
    # fold: w_ij + w_ji off the diagonal, w_ii on it
    w = torch.from_numpy(a.reshape(rows, n * n)).to(DEV)
    folded = torch.full((rows, ld), float("nan"), device=DEV)
    _lib.call("mdnvae_tri_fold", _lib.ptr(w), rows, n, ld, _lib.ptr(folded), _lib.stream())
    want = np.where(i == j, a[:, i, j], a[:, i, j] + a[:, j, i])
    assert np.array_equal(folded.cpu().numpy()[:, :T], want.astype(np.float32))
    assert not folded.cpu().numpy()[:, T:].any()
    # unfold from both storage orders
    g = rng.standard_normal((rows, T)).astype(np.float32)
    full = np.zeros((rows, n, n), np.float32)
    full[:, i, j] = g
    full[:, j, i] = g
    for src, st, sh in ((torch.from_numpy(g).to(DEV), 1, T), (torch.from_numpy(g.T.copy()).to(DEV), rows, 1)):
        out = torch.full((rows, n * n), float("nan"), device=DEV)
        if rows:
            _lib.call("mdnvae_tri_unfold", _lib.ptr(src), rows, n, st, sh, _lib.ptr(out), 0, _lib.stream())
        assert np.array_equal(out.cpu().numpy(), full.reshape(rows, n * n))


def test_pack_flags_asymmetric_and_nan_input():
    n = 20
    x = torch.randn(6, n, n, device=DEV)
    x = x + x.transpose(1, 2)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.tri_pack(x, n, flag)
    assert int(flag.item()) == 0
    y = x.clone()
    y[4, 3, 17] += 1e-6 * max(1.0, float(y[4, 3, 17].abs()))
    ops.tri_pack(y, n, flag)
    assert int(flag.item()) == 1
    flag.zero_()
    y = x.clone()
    y[2, 5, 5] = float("nan")
    ops.tri_pack(y, n, flag)
    assert int(flag.item()) == 1


def test_bad_arguments_are_refused():
    x = torch.zeros(4, device=DEV)
    with pytest.raises(Exception):
        _lib.call("mdnvae_tri_pack", _lib.ptr(x), 1, 193, 193 * 97, _lib.ptr(x), _lib.ptr(x), _lib.stream())
    with pytest.raises(Exception):
        _lib.call("mdnvae_tri_pack", _lib.ptr(x), 1, 4, 9, _lib.ptr(x), _lib.ptr(x), _lib.stream())   # stride < 10
    with pytest.raises(Exception):
        _lib.call("mdnvae_tri_unfold", _lib.ptr(x), 1, 2, 2, 2, _lib.ptr(x), 0, _lib.stream())


def test_symmetric_first_layer_matches_plain_layer_and_oracle():
    """GCN-decoder VAE of function/MM_Gate.py on symmetric input: symmetric_input=True must give the
    loss and every gradient of the plain model (and of the float64 oracle) at the usual bar."""
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    n = 68
    x = torch.from_numpy(fix["x"]).reshape(-1, n, n)
    x = (0.5 * (x + x.transpose(1, 2))).reshape(-1, n * n).contiguous()
    eps = torch.from_numpy(fix["eps"])
    adjacency = [list(map(int, r)) for r in fix["adjacency"]]
    results = {}
    for sym in (False, True):
        model = M.GCNVAE(R=5, n_nodes=n, k=32, n_layers=3, latent_dim=68, symmetric_input=sym)
        model.load_state_dict(params, strict=False)
        model = model.to(DEV)
        recon, mu, lv = model(x.to(DEV), adjacency, eps=eps.to(DEV))
        loss = ops.vae_loss(recon, x.to(DEV), mu, lv, "poisson_log")
        loss.backward()
        model.check_inputs()
        results[sym] = (loss.item(), {k: p.grad.clone() for k, p in model.named_parameters()})
    p = {k: v.detach().clone().double().requires_grad_(True) for k, v in params.items()}
    recon_c, mu_c, lv_c, _, _ = R.vae_forward("mm_gate", p, x.double(), eps.double(), adjacency=fix["adjacency"])
    want = R.vae_loss(recon_c, x.double(), mu_c, lv_c, "poisson_log")
    want.backward()
    for sym in (False, True):
        loss, grads = results[sym]
        assert abs(loss - want.item()) <= 1e-5 * abs(want.item()), sym
        for name, g in grads.items():
            assert _rel(g, p[name].grad) <= 1e-4, (sym, name)
    # the two first-layer gradients of the symmetric path are symmetric in (i, j) exactly
    g11 = results[True][1]["fc11.weight"].reshape(-1, n, n)
    assert torch.equal(g11, g11.transpose(1, 2))


def test_check_inputs_reports_an_asymmetric_batch():
    n = 16
    model = M.GCNVAE(R=2, n_nodes=n, k=4, n_layers=2, latent_dim=8, symmetric_input=True).to(DEV)
    x = torch.rand(5, n, n, device=DEV)
    adjacency = [[(i + 1) % n, (i + 2) % n, (i + 3) % n, (i + 4) % n] for i in range(n)]
    model(x.reshape(5, -1), adjacency)
    with pytest.raises(ValueError):
        model.check_inputs()
    model.check_inputs()                       # the flag is cleared by the report
    model((x + x.transpose(1, 2)).reshape(5, -1), adjacency)
    model.check_inputs()
