"""Multi-branch ResNet-MLP VAE (function_fc_new/ResNet.py) against the fixture generated from the
reference class, and the latent export of tools/extract.py."""
import os

import numpy as np
import pytest
import torch

from conftest import load_fixture, fixture_params
import recipe
import mdnvae_amd as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a = a.detach().cpu().double().numpy()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_resnet_vae_against_reference_fixture():
    fix = load_fixture("resnet_fc")
    model = M.ResNetVAE()
    missing = model.load_state_dict(fixture_params(fix), strict=True)
    model = model.to(DEV)
    x = torch.from_numpy(fix["x"]).to(DEV)
    recon, mu, logvar = model(x, eps=torch.from_numpy(fix["eps"]).to(DEV))
    loss = M.loss_function_mse(recon, x, mu, logvar)
    loss.backward()
    assert _rel(mu, fix["mu"]) <= 1e-5 and _rel(logvar, fix["logvar"]) <= 1e-5
    assert _rel(recon, fix["recon"]) <= 1e-5
    assert abs(loss.item() - float(fix["loss"])) <= 1e-5 * abs(float(fix["loss"]))
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        recipe.check_grad(fix, name, p.grad, 1e-4, "hip")


def test_latent_export_format(tmp_path):
    """mu as [n, L] float32 in <prefix>.npy and {'mu': ...} in <prefix>-un.mat (tools/extract.py:47-51)."""
    import scipy.io
    torch.manual_seed(0)
    model = M.GCNVAE(R=2, n_nodes=10, k=3, n_layers=1, latent_dim=6).to(DEV)
    data = torch.rand(23, 100, device=DEV)
    mu, logvar = M.evaluate.extract_latent(model, data, batch_size=8)
    assert mu.shape == (23, 6) and logvar.shape == (23, 6)
    whole, _ = model.encode(data)
    assert torch.allclose(mu, whole, rtol=1e-5, atol=1e-6)               # batching does not change the result
    npy, mat = M.evaluate.save_latent(mu, os.path.join(tmp_path, "mu-test"))
    assert np.array_equal(np.load(npy), mu.cpu().numpy())
    assert np.array_equal(scipy.io.loadmat(mat)["mu"], mu.cpu().numpy())
