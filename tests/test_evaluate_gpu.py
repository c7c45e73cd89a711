"""Evaluation loop (reconstruction in batches of 64, /max normalisation, clamp, MSE, PSNR) against
the CPU restatement; the reconstruction itself is the forward pass already checked against the
reference fixtures, here it must be independent of how the test set is batched."""
import numpy as np
import pytest
import torch

from conftest import load_fixture, fixture_params
from oracle import ref_eager as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import mdnvae_amd
    return mdnvae_amd


@pytest.mark.parametrize("n,cols", [(1, 7), (5, 4624), (300, 4624), (3, 87 * 87)])
def test_normalise_mse_psnr(M, n, cols):
    rng = np.random.default_rng(n + cols)
    real = rng.gamma(0.3, 2.0, (n, cols)).astype(np.float32)
    gen = (real + rng.normal(0, 0.3, (n, cols))).astype(np.float32)
    gen[0, 0] = -1.0                                             # exercises the clamp
    r_want, g_want = R.normalise_for_metrics(real, gen)
    r_got, g_got = M.evaluate.normalise_for_metrics(torch.from_numpy(real).cuda(), torch.from_numpy(gen).cuda())
    assert np.array_equal(r_got.cpu().numpy(), r_want)           # one fp32 division: bit-exact
    assert np.array_equal(g_got.cpu().numpy(), g_want)
    side = int(round(cols ** 0.5)) if int(round(cols ** 0.5)) ** 2 == cols else None
    shp = (n, side, side) if side else (n, 1, cols)
    want_mse = R.compute_mse(r_want.reshape(shp).astype(np.float64), g_want.reshape(shp).astype(np.float64))
    want_psnr = R.compute_psnr(r_want.reshape(shp).astype(np.float64), g_want.reshape(shp).astype(np.float64))
    assert M.evaluate.compute_mse(r_got, g_got) == pytest.approx(want_mse, rel=1e-6)
    assert M.evaluate.compute_psnr(r_got, g_got) == pytest.approx(want_psnr, rel=1e-6)


@pytest.mark.parametrize("n_train", [1, 2, 3, 4])
def test_normalise_on_a_split_that_starts_at_any_float(M, n_train):
    """The test split of 87 x 87 rows (7569 floats each) starts 16-byte aligned only when n_train is
    a multiple of 4; the maximum behind the normalisation must take the view as it is."""
    rng = np.random.default_rng(n_train)
    rows = rng.gamma(0.3, 2.0, (7, 87 * 87)).astype(np.float32)
    dev = torch.from_numpy(rows).cuda()
    test = dev[n_train:]                          # what data.train_test_split hands back: a view
    real, gen = test, (test * 0.9).contiguous()
    r_want, g_want = R.normalise_for_metrics(rows[n_train:], rows[n_train:] * np.float32(0.9))
    r_got, g_got = M.evaluate.normalise_for_metrics(real, gen)
    assert np.array_equal(r_got.cpu().numpy(), r_want) and np.array_equal(g_got.cpu().numpy(), g_want)


def test_maximum_propagates_nan_like_numpy(M):
    x = torch.rand(3, 1001, device="cuda")
    assert float(M.evaluate._max(x)) == float(x.max())
    x[1, 17] = float("nan")
    assert np.isnan(float(M.evaluate._max(x)))
    x[1, 17] = 0.0
    x[2, 1000] = float("nan")                    # in the tail past the last whole quad
    assert np.isnan(float(M.evaluate._max(x)))


def test_identical_inputs_have_zero_mse(M):
    x = torch.rand(4, 100, device="cuda")
    assert float(M.evaluate.matrix_mse(x, x).abs().max()) == 0.0
    assert np.isnan(M.evaluate.compute_psnr(x, x))


def test_reconstruct_matches_oracle_and_is_batch_independent(M):
    fix = load_fixture("mm_gate")
    params = fixture_params(fix)
    adjacency = [list(map(int, row)) for row in fix["adjacency"]]
    net = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68)
    net.load_state_dict(params)
    net.cuda()
    x = torch.from_numpy(fix["x"])
    eps = torch.from_numpy(fix["eps"])
    x, eps = x.repeat(40, 1)[:150], eps.repeat(40, 1)[:150]      # > 2 batches of 64, ragged tail
    want = R.reconstruct("mm_gate", params, x, eps, adjacency=adjacency)
    got = M.evaluate.reconstruct(net, x.cuda(), extra=(adjacency,), eps=eps.cuda())
    assert net.training                                           # mode restored
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    got32 = M.evaluate.reconstruct(net, x.cuda(), batch_size=32, extra=(adjacency,), eps=eps.cuda())
    np.testing.assert_allclose(got32.cpu().numpy(), got.cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_sample_prior_shapes_and_determinism(M):
    net = M.GCNVAE(R=5, n_nodes=68, k=32, n_layers=3, latent_dim=68).cuda()
    adjacency = [[(u + j) % 68 for j in range(32)] for u in range(68)]
    g = torch.Generator(device="cuda").manual_seed(7)
    a = M.evaluate.sample_prior(net, 70, extra=(adjacency,), generator=g)
    g = torch.Generator(device="cuda").manual_seed(7)
    b = M.evaluate.sample_prior(net, 70, extra=(adjacency,), generator=g)
    assert a.shape == (70, 4624) and torch.equal(a, b)
    mats = a.reshape(70, 68, 68)
    assert torch.allclose(mats, mats.transpose(1, 2), rtol=1e-5, atol=1e-6)   # decoder symmetrises
