"""Reading the reference's MATLAB files (host side: no kernel involved)."""
import numpy as np
import pytest
import torch
from scipy.io import savemat

import mdnvae_amd as M


def test_dense_tensor_layout(tmp_path):
    rng = np.random.default_rng(0)
    t = rng.poisson(50.0, (87, 87, 2, 5)).astype(np.float64)          # loaded_tensor_sub: [n, n, channels, subjects]
    path = str(tmp_path / "sc.mat")
    savemat(path, {"loaded_tensor_sub": t})
    got = M.data.load_mat_connectomes(path, device="cpu")
    assert got.shape == (5, 87, 87) and got.dtype == torch.float32
    for i in range(5):
        assert np.array_equal(got[i].numpy(), t[:, :, 0, i].astype(np.float32))
    got1 = M.data.load_mat_connectomes(path, "loaded_tensor_sub", channel=1, device="cpu")
    assert np.array_equal(got1[3].numpy(), t[:, :, 1, 3].astype(np.float32))
    with pytest.raises(KeyError):
        M.data.load_mat_connectomes(path, "all_fc", device="cpu")


def test_cell_array_layout_keeps_full_size_subjects_only(tmp_path):
    rng = np.random.default_rng(1)
    sizes = [(87, 87), (86, 86), (87, 87), (87, 87)]
    fc = np.empty((1, 4), dtype=object)
    sc = np.empty((1, 4), dtype=object)
    for i, sz in enumerate(sizes):
        fc[0, i] = rng.uniform(-1, 1, sz)
        sc[0, i] = rng.poisson(30.0, sz if i != 2 else (80, 80)).astype(np.float64)
    pf, ps = str(tmp_path / "fc.mat"), str(tmp_path / "sc.mat")
    savemat(pf, {"all_fc": fc})
    savemat(ps, {"all_sc": sc})
    got = M.data.load_mat_connectomes(pf, "all_fc", device="cpu")
    assert got.shape == (3, 87, 87)                                    # subject 1 has another size
    assert np.array_equal(got[1].numpy(), fc[0, 2].astype(np.float32))
    a, b = M.data.load_mat_pairs(pf, ps, device="cpu")
    assert a.shape == b.shape == (2, 87, 87)                           # subjects 0 and 3 are whole in both files
    assert np.array_equal(a[1].numpy(), fc[0, 3].astype(np.float32)) and np.array_equal(b[1].numpy(), sc[0, 3].astype(np.float32))
