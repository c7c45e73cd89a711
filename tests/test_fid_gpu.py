"""FID on the GPU against the reference's own known answers and scipy-generated fixtures.

Tolerance: features enter the GPU path as fp32 (the model's output type), the reference computes
from float64 numpy arrays; casting the inputs alone moves the result by ~1e-6 relative, so the bar
is 1e-5 relative for feature inputs and 1e-9 when float64 moments are handed over directly."""
import numpy as np
import pytest
import torch

from conftest import load_fixture
from recipe import record

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


@pytest.mark.parametrize("d", [16, 272, 1040])
def test_float64_matrix_core_products(M, d):
    """mdnvae_dgemm_sym against torch's float64 matmul: the general form (b symmetric), the symmetric form on a pair
    of commuting matrices (upper tiles only: the result is symmetric to the bit and agrees with the full product to
    rounding), and the residual epilogue 1.5 I - 0.5 a b.  Sizes: one tile, ragged tiles, several k steps."""
    from mdnvae_amd import fid
    g = torch.Generator(device="cuda").manual_seed(d)
    q = torch.randn(d, d, dtype=torch.float64, device="cuda", generator=g)
    b = (q + q.T) / (2 * d ** 0.5)
    a = torch.randn(d, d, dtype=torch.float64, device="cuda", generator=g)
    got_t = torch.full_like(a, float("nan"))
    got = fid._dgemm(a, b, torch.full_like(a, float("nan")), False, 0, got_t)
    want = a @ b
    assert float((got - want).abs().max()) <= 1e-13 * float(want.abs().max()) * d ** 0.5
    assert torch.equal(got_t, got.T)
    p2 = b @ b + 0.3 * b                                   # p2 b is symmetric
    p2 = (p2 + p2.T) / 2
    want = p2 @ b
    got = fid._dgemm(p2, b, torch.full_like(a, float("nan")), True)
    assert torch.equal(got, got.T)
    assert float((got - want).abs().max()) <= 1e-13 * float(want.abs().max()) * d ** 0.5
    got = fid._dgemm(p2, b, torch.full_like(a, float("nan")), True, 1)
    want = 1.5 * torch.eye(d, dtype=torch.float64, device="cuda") - 0.5 * want
    assert torch.equal(got, got.T)
    assert float((got - want).abs().max()) <= 1e-13 * float(want.abs().max()) * d ** 0.5
    with pytest.raises(RuntimeError):
        fid._dgemm(a[:10, :10].contiguous(), b[:10, :10].contiguous(), torch.empty(10, 10, dtype=torch.float64, device="cuda"), False)


def test_eigenvalue_estimate_and_its_fallback(M):
    """The power-iteration scale of sqrtm_psd: mdnvae_matvec_f64 against numpy, the estimate within a per cent on a
    generic covariance, and -- on a matrix built so that the fixed start vector is orthogonal to the dominant
    direction, where the estimate is 11 times too small and the first attempt diverges -- the retry under the
    rigorous norm bound still returns the root."""
    from mdnvae_amd import fid
    from mdnvae_amd import _lib
    rng = np.random.default_rng(3)
    for d in (5, 130, 700):
        a = rng.standard_normal((d, d))
        x = rng.standard_normal(d)
        at, xt = torch.from_numpy(a).cuda(), torch.from_numpy(x).cuda()
        y = torch.full((d,), float("nan"), dtype=torch.float64, device="cuda")
        _lib.call("mdnvae_matvec_f64", at.data_ptr(), xt.data_ptr(), d, y.data_ptr(), _lib.stream())
        np.testing.assert_allclose(y.cpu().numpy(), a @ x, rtol=0, atol=1e-13 * np.abs(a @ x).max() * d ** 0.5)
    cov = _psd(640, 1e-3, np.random.default_rng(640))
    est = fid._largest_eigenvalue_estimate(torch.from_numpy(cov).cuda())
    top = np.linalg.eigvalsh(cov)[-1]
    assert 0.95 * top <= est <= top * (1 + 1e-12)
    d = 128
    v0 = 1.0 + 0.5 * np.sin(0.7 * np.arange(d))
    u = rng.standard_normal(d)
    u -= v0 * (u @ v0) / (v0 @ v0)
    u /= np.linalg.norm(u)
    a = 10.0 * np.outer(u, u) + np.eye(d)
    at = torch.from_numpy((a + a.T) / 2).cuda()
    assert fid._largest_eigenvalue_estimate(at) < 1.5              # the dominant eigenvalue is 11
    root, _ = M.sqrtm_psd(at)
    got = root.cpu().numpy()
    assert np.linalg.norm(got @ got - a) / np.linalg.norm(a) < 1e-9


def test_stopping_rule_on_the_device():
    """mdnvae_ns_state_step against the rule written out on the host, on trace sequences that end each way: settled
    at rounding level, stalled below 1e-9, a landed gap, a non-finite trace; calls after the end change nothing, and
    a product given the ended record leaves its output alone."""
    from mdnvae_amd import fid, _lib

    def host_rule(traces, gaps):
        prev = last = None
        for n, (tr, gap) in enumerate(zip(traces, gaps), 1):
            if not (np.isfinite(tr) and (gap is None or np.isfinite(gap))):
                return 2.0, n
            if gap is not None and gap <= 1e-12:
                return 1.0, n
            if prev is not None:
                delta = abs(tr - prev)
                if delta <= 1e-15 * abs(tr) or (last is not None and delta >= last and delta < 1e-9 * abs(tr)):
                    return 1.0, n
                last = delta
            prev = tr
        return 0.0, len(traces)

    cases = [([1.0, 1.5, 1.75, 1.75, 9.0], None), ([1.0, 2.0, 2.0 + 1e-10, 2.0 + 3e-10, 5.0], None),
             ([1.0, 1.2, 1.3], [1e-3, 1e-13, 1e-30]), ([1.0, float("nan"), 2.0], None), ([1.0, 2.0, 3.0], [1.0, float("inf"), 0.5]),
             ([1.0, 2.0, 3.0, 4.5], None)]
    for traces, gaps in cases:
        state = fid._IterationState("cuda")
        gap = torch.zeros(1, dtype=torch.float64, device="cuda")
        for n, tr in enumerate(traces):
            y = torch.diag(torch.tensor([tr - 0.25, 0.25, 0.0], dtype=torch.float64)).cuda()
            if gaps is not None:
                gap.fill_(gaps[n])
            state.step(y, gap if gaps is not None else None)
        done, iters = state.rec[4:6].tolist()
        assert (done, int(iters)) == host_rule(traces, gaps if gaps is not None else [None] * len(traces)), (traces, gaps)
    ended = fid._IterationState("cuda")
    ended.rec[4] = 1.0
    a = torch.eye(16, dtype=torch.float64, device="cuda")
    out = torch.full((16, 16), 7.0, dtype=torch.float64, device="cuda")
    fid._dgemm(a, a, out, False, state=ended.rec)
    assert float(out.min()) == 7.0
    fid._dgemm(a, a, out, False, state=fid._IterationState("cuda").rec)
    assert torch.equal(out, a)


def _psd(d, floor, rng):
    if floor > 0:
        q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        a = (q * np.exp(rng.uniform(np.log(floor), 0.0, d))) @ q.T
    else:
        q = rng.standard_normal((d, 300))        # rank 300, like a covariance of fewer samples than features
        a = q @ q.T / 300.0
    return (a + a.T) / 2


@pytest.mark.parametrize("d,floor", [(640, 1e-6), (1000, 1e-3), (1024, 0.0)])
def test_large_principal_root_against_float64_eigendecomposition(M, d, floor):
    """sqrtm_psd against numpy's eigh on spectra that span 1/floor (floor 0: singular): root^2 against A at
    1e-9 of the Frobenius norm, the trace against the sum of root eigenvalues at 1e-10; on the singular
    matrix 1e-6 (the ridge that keeps the iteration bounded moves it by up to 1e-7 per null direction)."""
    a = _psd(d, floor, np.random.default_rng(d))
    want_tr = np.sqrt(np.clip(np.linalg.eigvalsh(a), 0, None)).sum()
    root, iters = M.sqrtm_psd(torch.from_numpy(a).cuda())
    got = root.cpu().numpy()
    err = np.linalg.norm(got @ got - a) / np.linalg.norm(a)
    tr_err = abs(np.trace(got) - want_tr) / want_tr
    tr_bar = 1e-10 if floor > 0 else 1e-6
    record("fid", "sqrtm_psd: root^2 vs A", f"d={d} spectrum floor {floor:g}", err, 1e-9, note=f"{iters} iterations")
    record("fid", "sqrtm_psd: trace vs sum of root eigenvalues", f"d={d} spectrum floor {floor:g}", tr_err, tr_bar)
    assert err < 1e-9 and tr_err < tr_bar


@pytest.mark.parametrize("d,floor,tr_bar", [(640, 0.05, 1e-9), (1000, 1e-3, 1e-5)])
def test_mixed_precision_root_factor(M, d, floor, tr_bar):
    """sqrt_factor (first iterations in split-bf16 products, float64 finish): F F^T is A to float64 rounding
    at any condition it accepts; the trace is off in second order of the fp32 phase's error, which grows
    with the condition number: 1e-9 at condition 20, 1e-5 at condition 1000."""
    a = _psd(d, floor, np.random.default_rng(d))
    want_tr = np.sqrt(np.clip(np.linalg.eigvalsh(a), 0, None)).sum()
    f, iters = M.sqrt_factor(torch.from_numpy(a).cuda())
    got = f.cpu().numpy()
    err = np.linalg.norm(got @ got.T - a) / np.linalg.norm(a)
    tr_err = abs(np.trace(got) - want_tr) / want_tr
    record("fid", "sqrt_factor: F F^T vs A", f"d={d} condition {1 / floor:g}", err, 1e-9, note=f"{iters} iterations")
    record("fid", "sqrt_factor: trace vs sum of root eigenvalues", f"d={d} condition {1 / floor:g}", tr_err, tr_bar)
    assert err < 1e-9 and tr_err < tr_bar


def test_fid_at_the_validation_size(M):
    """The conv scripts' validation at full size: two sets of 100 000 flattened 68 x 68 networks
    (4624 features).  Size-independent properties: moments accumulated in chunks, and summed over two
    halves as two data-parallel ranks would, equal the one-call moments; FID(A, A) = 0; FID(A, B) is
    symmetric; and on features with independent columns of known variance it sits just above the closed form
    |mu1 - mu2|^2 + sum (s1 - s2)^2 of the diagonal case (finite-sample bias)."""
    import time
    n, d = 100_000, 4624
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn(n, d, device="cuda", generator=g)
    b = torch.randn(n, d, device="cuda", generator=g) * 1.5 + 0.02
    shift = a[0].clone()
    whole = M.FeatureStats(d, shift).update(a)
    parts = [M.FeatureStats(d, shift), M.FeatureStats(d, shift)]
    half = n // 2 + 7
    for lo in range(0, half, 16384):
        parts[0].update(a[lo:min(lo + 16384, half)])
    parts[1].update(a[half:])
    parts[0].sum += parts[1].sum                 # what the all-reduce does
    parts[0].gram += parts[1].gram
    parts[0].n += parts[1].n
    m1, c1 = whole.finalize()
    m1p, c1p = parts[0].finalize()
    assert parts[0].n == n
    # the Gram kernel sums 1024-row chunks in fp32 before the float64 accumulator; other chunk boundaries
    # round differently: unit-variance features, so the bar is that of the moments test (2e-5)
    cov_err = float((c1 - c1p).abs().max())
    record("fid", "moments summed over chunks and ranks vs one call", "100000 x 4624", cov_err, 2e-5)
    assert cov_err < 2e-5 and float((m1 - m1p).abs().max()) < 1e-9
    del whole, parts
    m2, c2 = M.feature_stats(b)
    M.fid_from_stats(m1, c1, m1, c1)             # first call pays library start-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    same = M.fid_from_stats(m1, c1, m1, c1)
    torch.cuda.synchronize()
    t_same = time.perf_counter() - t0
    t0 = time.perf_counter()
    ab = M.fid_from_stats(m1, c1, m2, c2)
    torch.cuda.synchronize()
    t_ab = time.perf_counter() - t0
    ba = M.fid_from_stats(m2, c2, m1, c1)
    t0 = time.perf_counter()
    ab_mixed = M.fid_from_stats(m1, c1, m2, c2, mixed=True)
    torch.cuda.synchronize()
    t_mixed = time.perf_counter() - t0
    record("fid", "FID(A, B), mixed-precision roots vs float64 roots", "d=4624", abs(ab_mixed - ab) / ab, 1e-9,
           note=f"{t_mixed:.3f} s per call")
    assert abs(ab_mixed - ab) < 1e-9 * ab
    tr = float(torch.trace(c1))
    record("fid", "FID(A, A) / trace", "d=4624", abs(same) / tr, 1e-9, note=f"{t_same:.3f} s per call")
    assert abs(same) < 1e-9 * tr
    record("fid", "FID(A, B) vs FID(B, A)", "d=4624", abs(ab - ba) / ab, 1e-9, note=f"{t_ab:.3f} s per call")
    assert abs(ab - ba) < 1e-9 * ab
    # independent columns: the population FID is d * (0.02^2 + (1.5 - 1)^2); sample covariances of d / n = 0.05
    # spread their eigenvalues, which lowers tr sqrt(S1 S2) and biases the FID upward (by 14 % here)
    closed = d * (0.02 ** 2 + 0.25)
    assert closed < ab < 1.25 * closed
    assert t_ab < 1.0                            # two matrix square roots at d = 4624


def test_fid_rejects_bad_input(M):
    with pytest.raises(ValueError):
        M.feature_stats(torch.zeros(1, 4, device="cuda"))
    with pytest.raises(ValueError):
        M.calculate_fid(np.zeros(3), np.eye(3), np.zeros(3), np.eye(4))
    with pytest.raises(RuntimeError, match="device memory"):
        M._lib.ptr(torch.zeros(2, 2, dtype=torch.float64), "a", torch.float64)
