"""Shared recipe for the golden fixtures: how parameters are filled and which
gradient entries are kept.  Used by the generator (which runs the reference's
own class definitions, only in the build container) and by the tests (which
rebuild the same parameters for the oracle and for the HIP path).

A fixture stores no weights: every parameter is uniform(-bound, +bound) drawn
from a CPU torch.Generator seeded by crc32(name), with the per-name bound kept
in the fixture.  Large gradients are kept as a seeded sample of entries plus
their sum and L2 norm.
"""
import zlib

import numpy as np
import torch

FULL_GRAD_LIMIT = 16384      # gradients up to this many elements are stored whole
SAMPLE_COUNT = 2048          # entries kept for larger gradients


def name_seed(name: str, salt: int = 0) -> int:
    return (zlib.crc32(name.encode()) + 7919 * salt) % (2 ** 31 - 1)


def fill_param(name: str, shape, bound: float, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator().manual_seed(name_seed(name))
    u = torch.rand(tuple(shape), generator=g, dtype=torch.float64)
    return ((u * 2.0 - 1.0) * float(bound)).to(dtype)


def build_params(shapes: dict, bounds: dict, dtype=torch.float32) -> dict:
    return {k: fill_param(k, shapes[k], bounds[k], dtype) for k in shapes}


def sample_index(name: str, numel: int) -> np.ndarray:
    rng = np.random.default_rng(name_seed(name, salt=1))
    return np.sort(rng.choice(numel, size=SAMPLE_COUNT, replace=False)).astype(np.int64)


def pack_grad(name: str, g: torch.Tensor) -> dict:
    """Fixture entries for one gradient tensor (float64 numpy)."""
    flat = g.detach().reshape(-1).double().numpy()
    out = {f"gsum/{name}": np.float64(flat.sum()), f"gnorm/{name}": np.float64(np.linalg.norm(flat))}
    if flat.size <= FULL_GRAD_LIMIT:
        out[f"gfull/{name}"] = flat.astype(np.float32)
    else:
        out[f"gsamp/{name}"] = flat[sample_index(name, flat.size)].astype(np.float32)
    return out


def check_grad(fix, name: str, g: torch.Tensor, rtol: float, what: str = ""):
    """Compare a gradient tensor with the fixture entries; tolerance is relative
    to the gradient's own L2 norm scaled to one element (plus elementwise rtol)."""
    flat = g.detach().reshape(-1).double().cpu().numpy()
    norm = float(fix[f"gnorm/{name}"])
    scale = norm / np.sqrt(flat.size) if norm > 0 else 1.0
    if f"gfull/{name}" in fix:
        ref = fix[f"gfull/{name}"].astype(np.float64)
        got = flat
    else:
        ref = fix[f"gsamp/{name}"].astype(np.float64)
        got = flat[sample_index(name, flat.size)]
    err = np.abs(got - ref)
    tol = rtol * (np.abs(ref) + scale)
    bad = err > tol
    assert not bad.any(), (
        f"{what} grad {name}: {bad.sum()}/{bad.size} entries off, max err {err.max():.3e} "
        f"(scale {scale:.3e}, rtol {rtol})")
    gn = float(np.linalg.norm(flat))
    assert abs(gn - norm) <= rtol * max(norm, 1e-30) * 4 + 1e-30, f"{what} grad-norm {name}: {gn} vs {norm}"
