"""Connectome preparation on the device against rows produced by the reference's own loops
(tests/golden/prep.npz: raw matrices of 4 subjects and what function/VAE_CNN_NN.py:32-39,
function/Full_NN.py:106-112, function_FC/VAE_CNN_GCN.py:35-43 and function_fc_new/ResNet.py:33-41
make of them).  fp32 arithmetic on both sides; the raw inputs are stored as fp32 while the reference
adds the float64 originals before rounding, hence 1e-6 (1e-3 for Fisher z, whose slope is unbounded
at the clip)."""
import numpy as np
import pytest
import torch

from conftest import load_fixture
from oracle import ref_eager as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D():
    import mdnvae_amd
    return mdnvae_amd.data


@pytest.mark.parametrize("mode,raw_key,rows_key,crop,tol", [
    ("divide", "raw_sc0", "rows_sc", (18, 86), 1e-6), ("log1p", "raw_sc1", "rows_log", None, 1e-5),
    ("unit", "raw_fc", "rows_fc", (18, 86), 1e-6), ("fisher", "raw_fc", "rows_fisher", (18, 86), 1e-3)])
def test_preparation_matches_reference_rows(D, mode, raw_key, rows_key, crop, tol):
    fix = load_fixture("prep")
    got = D.prepare_connectomes(torch.from_numpy(fix[raw_key]).cuda(), mode, crop, 200.0).cpu().numpy()
    np.testing.assert_allclose(got, fix[rows_key], rtol=tol, atol=tol)


def test_group_mean_mask_and_adjacency(D):
    fix = load_fixture("prep")
    raw = torch.from_numpy(fix["raw_sc1"]).cuda()
    a = D.group_mean_matrix(raw)
    want = fix["raw_sc1"][:, 18:86, 18:86].astype(np.float64).mean(axis=0)
    np.testing.assert_allclose(a, want + want.T, rtol=1e-6, atol=1e-6)
    # masks / adjacency from the fixture's own A_mat (100 subjects) equal the reference's
    assert np.array_equal(D.rank_mask(fix["A_mat"]).numpy().astype(bool), fix["rank_mask"])
    assert np.array_equal(np.array(D.knn_adjacency(fix["A_mat"], 32)), fix["adjacency"])


def test_edge_cases(D):
    with pytest.raises(ValueError):
        D.prepare_connectomes(torch.zeros(2, 5, 4, device="cuda"), "divide")
    with pytest.raises(RuntimeError, match="crop"):
        D.prepare_connectomes(torch.zeros(2, 5, 5, device="cuda"), "divide", (3, 9))
    out = D.prepare_connectomes(torch.zeros(0, 87, 87, device="cuda"), "unit")
    assert out.shape == (0, 68 * 68)
    tr, te = D.train_test_split(torch.arange(10).reshape(10, 1))
    assert tr.shape[0] == 8 and te.shape[0] == 2
    x = D.synthetic_sc(3, 68, seed=5)
    m = x.reshape(3, 68, 68)
    assert torch.equal(m, m.transpose(1, 2))          # symmetric by construction
