"""FID on the GPU against the reference's own known answers and scipy-generated fixtures.

Tolerance: features enter the GPU path as fp32 (the model's output type), the reference computes
from float64 numpy arrays; casting the inputs alone moves the result by ~1e-6 relative, so the bar
is 1e-5 relative for feature inputs and 1e-9 when float64 moments are handed over directly."""
import numpy as np
import pytest
import torch

from conftest import load_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import mdnvae_amd
    return mdnvae_amd


def test_fid_script_known_answers(M):
    fix = load_fixture("fid")
    got_small = M.calculate_fid(fix["f_real"], fix["f_small"])
    got_large = M.calculate_fid(fix["f_real"], fix["f_large"])
    assert got_small == pytest.approx(10.414893149812082, rel=1e-5)
    assert got_large == pytest.approx(304.06124481745877, rel=1e-5)


@pytest.mark.parametrize("d", [10, 30])
def test_fid_from_float64_moments(M, d):
    fix = load_fixture("fid")
    a, b = fix[f"rand{d}/a"], fix[f"rand{d}/b"]
    got = M.calculate_fid(a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False))
    assert got == pytest.approx(float(fix[f"rand{d}/fid"]), rel=1e-9)
    got_f = M.calculate_fid(a, b)
    assert got_f == pytest.approx(float(fix[f"rand{d}/fid"]), rel=1e-5)


@pytest.mark.parametrize("n,d", [(2, 3), (257, 64), (1000, 65), (20000, 130)])
def test_moments_match_numpy(M, n, d):
    rng = np.random.default_rng(n * 1000 + d)
    x = (rng.standard_normal((n, d)) * rng.uniform(0.01, 3.0, d) + rng.uniform(-5, 5, d)).astype(np.float32)
    mean, cov = M.feature_stats(torch.from_numpy(x).cuda())
    x64 = x.astype(np.float64)
    want_cov = np.atleast_2d(np.cov(x64, rowvar=False))
    scale = np.sqrt(np.outer(np.diag(want_cov), np.diag(want_cov)))
    np.testing.assert_allclose(mean.cpu().numpy(), x64.mean(0), rtol=0, atol=1e-6 * np.abs(x64).max())
    assert np.max(np.abs(cov.cpu().numpy() - want_cov) / scale) < 2e-5
    # batched accumulation gives the same statistics as one call
    st = M.FeatureStats(d, torch.from_numpy(x[0]).cuda())
    for lo in range(0, n, 300):
        st.update(torch.from_numpy(x[lo:lo + 300]).cuda())
    mean2, cov2 = st.finalize()
    assert np.max(np.abs(cov2.cpu().numpy() - want_cov) / scale) < 2e-5
    np.testing.assert_allclose(mean2.cpu().numpy(), mean.cpu().numpy(), rtol=0, atol=1e-9 * np.abs(x64).max() + 1e-12)


def test_sqrtm_psd_properties(M):
    """Size-independent checks at the flattened-network dimension of the conv scripts' validation
    (68*68 = 4624 features): sqrt(A)^2 = A and FID(A, A) = 0."""
    d = 1024
    g = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn(d, 300, dtype=torch.float64, device="cuda", generator=g)
    a = q @ q.T / 300.0                      # rank-deficient PSD, like n < d covariances
    root, iters = M.sqrtm_psd(a)
    assert iters < 100
    assert float((root @ root - a).norm() / a.norm()) < 1e-9
    mu = torch.zeros(d, dtype=torch.float64)
    assert abs(M.calculate_fid(mu, a, mu, a)) < 1e-6 * float(torch.trace(a))


def test_fid_rejects_bad_input(M):
    with pytest.raises(ValueError):
        M.feature_stats(torch.zeros(1, 4, device="cuda"))
    with pytest.raises(ValueError):
        M.calculate_fid(np.zeros(3), np.eye(3), np.zeros(3), np.eye(4))
    with pytest.raises(RuntimeError, match="device memory"):
        M._lib.ptr(torch.zeros(2, 2, dtype=torch.float64), "a", torch.float64)
