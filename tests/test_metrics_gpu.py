"""Batched evaluation metrics: top-k spectra (against np.linalg.eigvals as the reference calls it) and
SSIM (against the CPU restatement of skimage's published algorithm: skimage is absent, so that part is
parity unpinned).  Tolerances: eigenvalues 2e-5 of the largest |eigenvalue| (fp32 Jacobi against
float64 LAPACK), SSIM 1e-5 absolute."""
import numpy as np
import pytest
import torch

from conftest import load_fixture
from oracle import ref_eager as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import mdnvae_amd
    return mdnvae_amd.evaluate


def _sym(rng, count, n):
    a = rng.standard_normal((count, n, n)).astype(np.float32)
    return a + a.transpose(0, 2, 1)


@pytest.mark.parametrize("n,k", [(2, 2), (5, 3), (68, 10), (68, 30), (87, 30), (96, 96)])
def test_spectrum_matches_numpy(E, n, k):
    rng = np.random.default_rng(n * 100 + k)
    mats = _sym(rng, 7, n)
    got = E.spectral_features(torch.from_numpy(mats).cuda(), k).cpu().numpy()
    for i in range(7):
        want = R.compute_spectral_features(mats[i].astype(np.float64), k)
        assert np.max(np.abs(got[i] - want)) <= 2e-5 * np.abs(want).max()


def test_spectrum_of_real_connectomes_and_fid(E):
    """The preprocessed HCP rows of the fixture (symmetric 68x68, heavy-tailed) and the spectral FID
    of function/VAE_CNN_NN.py:318-337 between them and a perturbed copy."""
    import mdnvae_amd as M
    x = load_fixture("cnn_nn_sc")["x"].reshape(-1, 68, 68)
    rng = np.random.default_rng(3)

    def jitter(scale):
        e = rng.standard_normal((48, 68, 68)).astype(np.float32) * scale
        return e + e.transpose(0, 2, 1)

    base = np.concatenate([x * s for s in np.linspace(0.5, 1.5, 12)]).astype(np.float32)      # 48 matrices
    real = np.maximum(base + jitter(0.2), 0).astype(np.float32)
    gen = np.maximum(base * 1.05 + jitter(0.3), 0).astype(np.float32)
    fr = E.spectral_features(torch.from_numpy(real).cuda(), 10)
    fg = E.spectral_features(torch.from_numpy(gen).cuda(), 10)
    want = np.array([R.compute_spectral_features(m.astype(np.float64), 10) for m in real])
    assert np.max(np.abs(fr.cpu().numpy() - want)) <= 2e-5 * np.abs(want).max()
    wg = np.array([R.compute_spectral_features(m.astype(np.float64), 10) for m in gen])
    # the FID step on identical (device-computed) features, then the whole chain against the CPU one
    same = R.fid_from_features(fr.cpu().numpy().astype(np.float64), fg.cpu().numpy().astype(np.float64))
    fid_got = E.spectral_fid(torch.from_numpy(real).cuda(), torch.from_numpy(gen).cuda(), 10)
    assert fid_got == pytest.approx(same, rel=1e-6)
    assert fid_got == pytest.approx(R.fid_from_features(want, wg), rel=2e-3)


def test_spectrum_refuses_asymmetric_input(E):
    m = torch.randn(3, 10, 10, device="cuda")
    with pytest.raises(ValueError, match="symmetric"):
        E.spectral_features(m, 3)
    assert E.spectral_features(torch.zeros(0, 10, 10, device="cuda"), 3).shape == (0, 3)


@pytest.mark.parametrize("n", [7, 17, 68, 87])
def test_ssim_matches_published_algorithm(E, n):
    rng = np.random.default_rng(n)
    a = rng.random((5, n, n)).astype(np.float32)
    b = np.clip(a + rng.standard_normal((5, n, n)).astype(np.float32) * 0.1, 0, 1).astype(np.float32)
    b[0] = a[0]                                           # identical pair: SSIM = 1
    got = E.matrix_ssim(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    want = np.array([R.ssim(a[i], b[i]) for i in range(5)])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)
    assert got[0] == pytest.approx(1.0, abs=1e-6)
    assert E.compute_ssim_average(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()) == pytest.approx(want.mean(), abs=1e-5)


def _sbm(fix, name):
    return np.unpackbits(fix[f"sbm_{name}_bits"], axis=1)[:, :10000].reshape(-1, 100, 100).astype(np.float32)


def test_graph_features_match_the_reference_networkx_outputs(E):
    """tests/golden/graphs.npz: outputs of the reference's compute_graph_features (both variants) on
    real connectomes, on a disconnected matrix with an isolated node, and on the SBM graphs of
    validation/FID.py.  fp32 on the device against networkx's mixed float32/float64: 3e-5 relative."""
    fix = load_fixture("graphs")
    for tag, weighted in (("fidpy", True), ("script", False)):
        got = E.graph_features(torch.from_numpy(fix["connectomes"]).cuda(), weighted).cpu().numpy()
        np.testing.assert_allclose(got, fix[f"connectome_features_{tag}"], rtol=3e-5, atol=1e-7)
        got = E.graph_features(torch.from_numpy(fix["disconnected"][None]).cuda(), weighted).cpu().numpy()[0]
        np.testing.assert_allclose(got, fix[f"disconnected_features_{tag}"], rtol=3e-5, atol=1e-7)
    for name in ("real", "small", "large"):
        got = E.graph_features(torch.from_numpy(_sbm(fix, name)).cuda(), True).cpu().numpy()
        np.testing.assert_allclose(got, fix[f"sbm_{name}_features"], rtol=3e-5, atol=1e-7)


def test_fid_script_end_to_end_on_the_device(E):
    """validation/FID.py from its graphs to its two printed numbers, entirely on the device."""
    import mdnvae_amd as M
    fix = load_fixture("graphs")
    f = {name: E.graph_features(torch.from_numpy(_sbm(fix, name)).cuda(), True) for name in ("real", "small", "large")}
    assert M.calculate_fid(f["real"], f["small"]) == pytest.approx(10.414893149812082, rel=2e-3)
    assert M.calculate_fid(f["real"], f["large"]) == pytest.approx(304.06124481745877, rel=2e-3)


def test_graph_features_edge_cases(E):
    z = torch.zeros(2, 12, 12, device="cuda")                      # no edges at all
    assert torch.equal(E.graph_features(z), torch.zeros(2, 5, device="cuda"))
    assert E.graph_features(torch.zeros(0, 12, 12, device="cuda")).shape == (0, 5)
    with pytest.raises(RuntimeError, match="exceeds"):
        E.graph_features(torch.zeros(1, 129, 129, device="cuda"))
