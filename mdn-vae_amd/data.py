"""Synthetic connectomes shaped like the reference's inputs, generated on the device.

SC-like: upper-triangular counts ~ round(Gamma(0.3, scale 240)) (heavy tail, mean ~72 like
data/HCP_samples.mat), then the reference's preparation: t + t^T, diagonal := column means,
optional crop [18:86], / 200  (function/VAE_CNN_NN.py:33-38).
FC-like: correlation matrices of a rank-8 factor model, then (x + x^T with the diagonal rule,
crop) mapped to [0,1] by (x+1)/2  (function_FC/VAE_CNN_GCN.py:35-41).
"""
import numpy as np
import torch

from . import _lib

PREP_MODES = {"divide": 0, "unit": 1, "log1p": 2, "fisher": 3, "raw": 4}
CORTICAL_CROP = (18, 86)       # the 68 cortical Desikan regions of the 87-region matrices


def prepare_connectomes(raw, mode, crop=CORTICAL_CROP, offset=200.0):
    """raw [B, n, n] fp32 on the device -> prepared rows [B, w*w] (w = crop width, or n).

    Per subject: raw + raw^T, diagonal := column means, crop, scale, flatten - the loops of
    function/VAE_CNN_NN.py:32-39 ('divide': / offset), function_FC/VAE_CNN_GCN.py:35-43 ('unit':
    (x+1)/2), function/Full_NN.py:106-112 ('log1p', crop=None), function_fc_new/ResNet.py:33-41
    ('fisher'); 'raw' only crops."""
    if raw.dim() != 3 or raw.shape[1] != raw.shape[2]:
        raise ValueError(f"expected raw connectomes [B, n, n], got {tuple(raw.shape)}")
    raw = raw.contiguous()
    n = raw.shape[1]
    lo, hi = (0, n) if crop is None else crop
    out = torch.empty(raw.shape[0], (hi - lo) ** 2, dtype=torch.float32, device=raw.device)
    _lib.call("mdnvae_prep_connectome", _lib.ptr(raw, "raw"), raw.shape[0], n, lo, hi, PREP_MODES[mode], float(offset),
              _lib.ptr(out, "out"), _lib.stream())
    return out


def load_mat_connectomes(path, key=None, channel=0, shape=(87, 87), device="cuda"):
    """Raw connectomes [subjects, n, n] fp32 on `device` from one of the reference's MATLAB files,
    ready for prepare_connectomes (host side: scipy's reader, one copy to the device).

    Two layouts occur.  A dense array `[n, n, channels, subjects]` or `[n, n, subjects]`
    (`loaded_tensor_sub` of HCP_subcortical_CMData_desikan.mat, function/VAE_CNN_NN.py:16-33, which
    takes channel 0 for the networks and channel 1 for the group mean).  A 1 x S cell array of per-
    subject matrices, some of another size, of which only those of `shape` are kept (`all_fc` /
    `all_sc` of desikan_fc_all.mat / desikan_sc_all.mat, function_FC/VAE_CNN_GCN.py:21-28;
    function_FC_SC/SC-FC.py:20-33 keeps a subject only if BOTH files hold a matrix of that shape - use
    load_mat_pairs for that).  `key`: the variable name; default = the only non-private variable."""
    arr = _mat_variable(path, key)
    mats = _subject_stack(arr, channel, shape)
    return torch.from_numpy(mats).to(device)


def load_mat_pairs(path_a, path_b, key_a=None, key_b=None, shape=(87, 87), device="cuda"):
    """Two cell-array files read side by side: subject i is kept only when both hold a matrix of
    `shape` for it (function_FC_SC/SC-FC.py:26-33).  Returns the two stacks [subjects, n, n]."""
    a, b = _mat_variable(path_a, key_a), _mat_variable(path_b, key_b)
    if a.dtype != object or b.dtype != object:
        raise ValueError("load_mat_pairs expects the cell-array layout in both files")
    ca, cb = a.reshape(-1), b.reshape(-1)
    keep = [i for i in range(min(len(ca), len(cb)))
            if getattr(ca[i], "shape", None) == tuple(shape) and getattr(cb[i], "shape", None) == tuple(shape)]
    if not keep:
        raise ValueError(f"no subject has {shape} matrices in both files")
    sa = np.stack([np.asarray(ca[i], dtype=np.float32) for i in keep])
    sb = np.stack([np.asarray(cb[i], dtype=np.float32) for i in keep])
    return torch.from_numpy(sa).to(device), torch.from_numpy(sb).to(device)


def _mat_variable(path, key):
    from scipy.io import loadmat
    mat = loadmat(path)
    names = [k for k in mat if not k.startswith("__")]
    if key is None:
        if len(names) != 1:
            raise ValueError(f"{path} holds {names}: say which with key=")
        key = names[0]
    if key not in mat:
        raise KeyError(f"{path} has no variable {key!r} (it holds {names})")
    return mat[key]


def _subject_stack(arr, channel, shape):
    if arr.dtype == object:                       # cell array of per-subject matrices
        cells = [np.asarray(m, dtype=np.float32) for m in arr.reshape(-1) if getattr(m, "shape", None) == tuple(shape)]
        if not cells:
            raise ValueError(f"no {shape} matrix among the {arr.size} cells")
        return np.stack(cells)
    if arr.ndim == 4:                             # [n, n, channels, subjects]
        arr = arr[:, :, channel, :]
    if arr.ndim != 3 or arr.shape[0] != arr.shape[1]:
        raise ValueError(f"expected [n, n, (channels,) subjects], got {arr.shape}")
    return np.ascontiguousarray(np.moveaxis(arr, -1, 0), dtype=np.float32)


def prepare_sc(raw, offset=200.0, crop=CORTICAL_CROP):
    return prepare_connectomes(raw, "divide", crop, offset)


def prepare_fc(raw, crop=CORTICAL_CROP):
    return prepare_connectomes(raw, "unit", crop)


def group_mean_matrix(raw, crop=CORTICAL_CROP, symmetrise=True):
    """A_mat of the scripts: the mean over subjects of the cropped raw matrices, plus its transpose
    for the SC scripts (function/VAE_CNN_NN.py:21-22; function_FC/VAE_CNN_GCN.py:31 has no
    transpose).  Returned as float64 numpy, the form the mask / adjacency recipes take."""
    from .ops import _colsum
    rows = prepare_connectomes(raw, "raw", crop)
    w = int(round(rows.shape[1] ** 0.5))
    a = (_colsum(rows) / rows.shape[0]).reshape(w, w).double().cpu().numpy()
    return a + a.T if symmetrise else a


def rank_mask(a_mat, n_size=32):
    """0/1 mask of the n_size+1 smallest entries of every row (function/VAE-CNN-GCN.py:209); host
    side, like the reference (a 68x68 argsort done once)."""
    a_mat = np.asarray(a_mat)
    return torch.from_numpy(np.float32(np.argsort(np.argsort(a_mat, axis=-1), axis=-1) < n_size + 1))


def knn_adjacency(a_mat, k=32):
    """Per node: itself, then its k-1 strongest other neighbours (function/MM_Gate.py:55-65)."""
    a_mat = np.asarray(a_mat)
    adj = []
    for u in range(a_mat.shape[0]):
        desc = np.argsort(a_mat[u, :])[::-1][:k]
        desc = desc[desc != u][: k - 1]
        adj.append([u] + [int(v) for v in desc])
    return adj


def train_test_split(rows, train_fraction=0.8):
    """First 80 % of the subjects train, the rest test (function/VAE_CNN_NN.py:43-48)."""
    n_train = int(train_fraction * rows.shape[0])
    return rows[:n_train], rows[n_train:]


def synthetic_sc(n, n_nodes=87, seed=1234, device="cuda", offset=200.0):
    """[n, n_nodes^2] fp32 structural-connectome rows; n_nodes 87 (raw size) or 68 (cropped)."""
    g = torch.Generator(device=device).manual_seed(seed)
    raw_n = 87
    shape = torch.full((n, raw_n, raw_n), 0.3, device=device)
    counts = torch.round(torch._standard_gamma(shape, generator=g) * 240.0)
    counts = torch.triu(counts, diagonal=1)
    if n_nodes not in (68, 87):
        raise ValueError("n_nodes must be 87 (raw Desikan size) or 68 (cortical crop)")
    return prepare_sc(counts, offset, CORTICAL_CROP if n_nodes == 68 else None)


def synthetic_fc(n, n_nodes=87, seed=1234, device="cuda"):
    g = torch.Generator(device=device).manual_seed(seed)
    raw_n = 87
    u = torch.randn(n, raw_n, 8, device=device, generator=g)
    c = u @ u.transpose(1, 2) + 0.5 * torch.eye(raw_n, device=device)
    d = torch.rsqrt(torch.diagonal(c, dim1=1, dim2=2))
    c = c * d.unsqueeze(1) * d.unsqueeze(2)
    c = torch.triu(c, diagonal=1) * 0.5              # t + t^T then restores the off-diagonal value
    return prepare_fc(c, CORTICAL_CROP if n_nodes == 68 else None)


def shard_range(n, rank, world):
    """Contiguous block of subject indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
