"""The CPU oracle (oracle/ref_eager.py) against the fixtures produced by the
reference's own class definitions.  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import recipe
from conftest import load_fixture, fixture_params
from oracle import ref_eager as R

LOSS_RTOL = 2e-6      # fp32 eager vs fp32 eager, same op order up to reductions
OUT_RTOL = 2e-5
GRAD_RTOL = 2e-4


def _leaf(params):
    return {k: v.clone().requires_grad_(True) for k, v in params.items()}


def _close(a, b, rtol, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b).max()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _check_grads(fix, params, skip_noise=True):
    for name, p in params.items():
        if p.grad is None:
            continue
        if skip_noise and name.startswith("ec.") and name.endswith(".bias"):
            # a conv bias in front of an instance norm has an exactly-zero gradient; both
            # sides hold rounding noise there
            wn = float(fix["gnorm/" + name[:-4] + "weight"])
            assert float(p.grad.norm()) <= 1e-5 * wn
            continue
        if name == "fcintercept.weight" and p.grad.ndim == 2 and "mask" in fix.files:
            g = torch.diagonal(p.grad)   # off-diagonal entries are masked to zero
            full = torch.zeros_like(p.grad)
            full.diagonal().copy_(g)
            recipe.check_grad(fix, name, full, GRAD_RTOL, "oracle")
            continue
        recipe.check_grad(fix, name, p.grad, GRAD_RTOL, "oracle")


@pytest.mark.parametrize("variant", ["cnn_nn_sc", "cnn_gcn_sc", "cnn_gcn_fc", "cnn_nn_fc", "fnn_gcn_fc"])
def test_conv_vae_variants(variant):
    fix = load_fixture(variant)
    p = _leaf(fixture_params(fix))
    mask = torch.from_numpy(fix["mask"]) if "mask" in fix.files else None
    if mask is not None:
        # set_mask zeroes the masked weights once (function/VAE-CNN-GCN.py:66-69)
        with torch.no_grad():
            for n in ("fc4", "fc5", "fc6", "fc7", "fc8"):
                p[n + ".weight"].mul_(mask)
            p["fcintercept.weight"].mul_(torch.eye(68 * 68))
    x = torch.from_numpy(fix["x"])
    eps = torch.from_numpy(fix["eps"])
    recon, mu, logvar, z, x_latent = R.vae_forward(variant, p, x, eps, mask=mask)
    loss = R.vae_loss(recon, x, mu, logvar, R.loss_mode(variant))
    loss.backward()
    _close(mu.detach(), fix["mu"], OUT_RTOL, "mu")
    _close(logvar.detach(), fix["logvar"], OUT_RTOL, "logvar")
    _close(z.detach(), fix["z"], OUT_RTOL, "z")
    _close(recon.detach(), fix["recon"], OUT_RTOL, "recon")
    _close(x_latent.detach(), fix["x_latent"], OUT_RTOL, "x_latent")
    assert abs(loss.item() - float(fix["loss"])) <= LOSS_RTOL * abs(float(fix["loss"]))
    # and the fp32 result sits within fp32 noise of the fp64 one
    assert abs(loss.item() - float(fix["loss64"])) <= 1e-5 * abs(float(fix["loss64"]))
    _check_grads(fix, p)


def test_mm_gate():
    fix = load_fixture("mm_gate")
    p = _leaf(fixture_params(fix))
    x = torch.from_numpy(fix["x"])
    eps = torch.from_numpy(fix["eps"])
    adj = R.knn_adjacency(fix["A_mat"], 32)
    assert np.array_equal(adj, fix["adjacency"])
    recon, mu, logvar, z, _ = R.vae_forward("mm_gate", p, x, eps, adjacency=adj)
    loss = R.vae_loss(recon, x, mu, logvar, "poisson_log")
    loss.backward()
    _close(recon.detach(), fix["recon"], OUT_RTOL, "recon")
    _close(mu.detach(), fix["mu"], OUT_RTOL, "mu")
    assert abs(loss.item() - float(fix["loss"])) <= LOSS_RTOL * abs(float(fix["loss"]))
    _check_grads(fix, p)


def test_cvae_hinge():
    fix = load_fixture("cvae")
    p = _leaf(fixture_params(fix))
    fc, sc = torch.from_numpy(fix["fc"]), torch.from_numpy(fix["sc"])
    loss, with_fc, without, recon, recon_no = R.cvae_hinge_loss(
        p, fc, sc, fix["mask_idx"], torch.from_numpy(fix["eps_a"]), torch.from_numpy(fix["eps_b"]))
    loss.backward()
    _close(recon.detach(), fix["recon"], OUT_RTOL, "recon")
    _close(recon_no.detach(), fix["recon_no"], OUT_RTOL, "recon_no")
    for got, key in ((loss, "loss"), (with_fc, "loss_with"), (without, "loss_no")):
        assert abs(got.item() - float(fix[key])) <= LOSS_RTOL * abs(float(fix[key])), key
    _check_grads(fix, p)


def test_encoder_87():
    fix = load_fixture("ec87")
    p = _leaf(fixture_params(fix))
    x = torch.from_numpy(fix["x"]).reshape(2, -1)
    feat = R.conv_features(p, x, n_nodes=87)
    mu = torch.nn.functional.linear(feat, p["fc_mean.weight"], p["fc_mean.bias"])
    lv = torch.nn.functional.linear(feat, p["fc_log_var.weight"], p["fc_log_var.bias"])
    ((mu * torch.from_numpy(fix["cot_mu"])).sum() + (lv * torch.from_numpy(fix["cot_lv"])).sum()).backward()
    _close(feat.detach(), fix["feat"], OUT_RTOL, "feat")
    _close(mu.detach(), fix["mu"], OUT_RTOL, "mu")
    _check_grads(fix, p)


@pytest.mark.parametrize("mode,rkey,xkey", [("poisson_rate", "rate", "x_cnt"), ("poisson_log", "lograte", "x_cnt"),
                                            ("mse", "r01", "x01")])
def test_loss_functions(mode, rkey, xkey):
    fix = load_fixture("losses")
    r = torch.from_numpy(fix[rkey]).requires_grad_(True)
    mu = torch.from_numpy(fix["mu"]).requires_grad_(True)
    lv = torch.from_numpy(fix["logvar"]).requires_grad_(True)
    loss = R.vae_loss(r, torch.from_numpy(fix[xkey]), mu, lv, mode)
    loss.backward()
    assert abs(loss.item() - float(fix[f"{mode}/loss"])) <= LOSS_RTOL * abs(float(fix[f"{mode}/loss"]))
    assert abs(loss.item() - float(fix[f"{mode}/loss64"])) <= 1e-5 * abs(float(fix[f"{mode}/loss64"]))
    _close(r.grad, fix[f"{mode}/grad_recon"], 1e-5, "grad_recon")
    _close(mu.grad, fix[f"{mode}/grad_mu"], 1e-6, "grad_mu")
    _close(lv.grad, fix[f"{mode}/grad_logvar"], 1e-6, "grad_logvar")


def test_fid_known_answers():
    fix = load_fixture("fid")
    # validation/FID.py run end to end with its own seeds printed these two numbers
    assert float(fix["fid_small"]) == pytest.approx(10.414893149812082, rel=1e-12)
    assert float(fix["fid_large"]) == pytest.approx(304.06124481745877, rel=1e-12)
    assert R.fid_from_features(fix["f_real"], fix["f_small"]) == pytest.approx(float(fix["fid_small"]), rel=1e-10)
    assert R.fid_from_features(fix["f_real"], fix["f_large"]) == pytest.approx(float(fix["fid_large"]), rel=1e-10)
    for d in (10, 30):
        a, b = fix[f"rand{d}/a"], fix[f"rand{d}/b"]
        want = float(fix[f"rand{d}/fid"])
        assert R.fid_from_features(a, b) == pytest.approx(want, rel=1e-10)
        sym = R.fid_symmetric(a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False))
        assert sym == pytest.approx(want, rel=1e-8)


def test_prep_masks_and_adjacency():
    fix = load_fixture("prep")
    assert np.array_equal(R.rank_mask(fix["A_mat"]), fix["rank_mask"])
    assert np.array_equal(R.knn_adjacency(fix["A_mat"], 32), fix["adjacency"])
    assert (fix["rank_mask"].sum(axis=1) == 33).all()


def test_preparation_loops_of_the_reference():
    """The oracle's preparation against rows produced by the reference's own loops."""
    fix = load_fixture("prep")
    for i in range(4):
        np.testing.assert_allclose(R.prepare_sc(fix["raw_sc0"][i].astype(np.float64) / 1.0), fix["rows_sc"][i],
                                   rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(R.prepare_log(fix["raw_sc1"][i].astype(np.float64)), fix["rows_log"][i],
                                   rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(R.prepare_fc(fix["raw_fc"][i].astype(np.float64)), fix["rows_fc"][i],
                                   rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(R.prepare_fisher(fix["raw_fc"][i].astype(np.float64)), fix["rows_fisher"][i],
                                   rtol=1e-3, atol=1e-3)


def test_graph_features_against_networkx_outputs_of_the_reference():
    fix = load_fixture("graphs")
    for tag, weighted in (("fidpy", True), ("script", False)):
        for i in range(4):
            got = R.graph_features(fix["connectomes"][i], weighted)
            np.testing.assert_allclose(got, fix[f"connectome_features_{tag}"][i], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(R.graph_features(fix["disconnected"], weighted), fix[f"disconnected_features_{tag}"],
                                   rtol=2e-5, atol=1e-7)
    bits = np.unpackbits(fix["sbm_real_bits"], axis=1)[:, :10000].reshape(-1, 100, 100).astype(np.float64)
    for i in range(0, 50, 10):
        np.testing.assert_allclose(R.graph_features(bits[i], True), fix["sbm_real_features"][i], rtol=2e-5, atol=1e-7)


def test_resnet_mlp_vae():
    """function_fc_new/ResNet.py with the constructor arguments of its main()."""
    fix = load_fixture("resnet_fc")
    p = _leaf(fixture_params(fix))
    x, eps = torch.from_numpy(fix["x"]), torch.from_numpy(fix["eps"])
    recon, mu, logvar, z = R.resnet_vae_forward(p, x, eps)
    loss = R.vae_loss(recon, x, mu, logvar, "mse")
    loss.backward()
    _close(mu.detach(), fix["mu"], OUT_RTOL, "mu")
    _close(logvar.detach(), fix["logvar"], OUT_RTOL, "logvar")
    _close(recon.detach(), fix["recon"], OUT_RTOL, "recon")
    assert abs(loss.item() - float(fix["loss"])) <= LOSS_RTOL * abs(float(fix["loss"]))
    _check_grads(fix, p)


def test_conv_encoder_with_its_own_relu_gates_passed_back_in():
    """conv_features(gates=...) with the masks of its own forward is the same function, gradients included."""
    fix = load_fixture("cnn_nn_fc")
    x = torch.from_numpy(fix["x"]).double()
    runs = []
    for gated in (False, True):
        p = {k: v.clone().requires_grad_(True) for k, v in fixture_params(fix, torch.float64).items()}
        normed = []
        f = R.conv_features(p, x, normed=normed, gates=[v > 0 for v in runs[0][2]] if gated else None)
        (f * torch.linspace(-1, 1, f.numel(), dtype=torch.float64).view_as(f)).sum().backward()
        runs.append((f.detach(), p["ec.1.weight"].grad, normed))
    assert len(runs[0][2]) == 8
    assert torch.equal(runs[0][0], runs[1][0]) and torch.allclose(runs[0][1], runs[1][1], rtol=1e-12, atol=0)
