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

import os

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


def pack_grad(name: str, g: torch.Tensor, tag: str = "g") -> dict:
    """Fixture entries for one gradient tensor.  tag 'g' = the reference's fp32 run,
    tag 'h' = the same classes run in float64 (the noise-floor reference)."""
    flat = g.detach().reshape(-1).double().numpy()
    out = {f"{tag}sum/{name}": np.float64(flat.sum()), f"{tag}norm/{name}": np.float64(np.linalg.norm(flat))}
    store = np.float32 if tag == "g" else np.float64
    if flat.size <= FULL_GRAD_LIMIT:
        out[f"{tag}full/{name}"] = flat.astype(store)
    else:
        out[f"{tag}samp/{name}"] = flat[sample_index(name, flat.size)].astype(store)
    return out


def check_grad(fix, name: str, g: torch.Tensor, rtol: float, what: str = ""):
    """Compare a gradient tensor with the fixture.

    Without a float64 entry: every kept entry within rtol*(|ref| + rms(ref)) of the
    reference's fp32 value.  With one: the reference's own fp32 run sits some distance
    (`noise`) from the float64 run of the same classes -- for gradients that have crossed
    the whole conv stack that distance is well above rtol -- so the bar becomes: no further
    from the float64 gradient than rtol or 3x that fp32 noise, whichever is larger
    (L2 over the kept entries), and no single entry further than rtol*(|ref|+rms) + 4x the
    largest fp32 deviation."""
    flat = g.detach().reshape(-1).double().cpu().numpy()
    norm = float(fix[f"gnorm/{name}"])
    scale = norm / np.sqrt(flat.size) if norm > 0 else 1.0
    full = f"gfull/{name}" in fix
    ref32 = fix[f"gfull/{name}" if full else f"gsamp/{name}"].astype(np.float64)
    got = flat if full else flat[sample_index(name, flat.size)]
    key64 = f"hfull/{name}" if full else f"hsamp/{name}"
    if key64 in fix:
        ref = fix[key64].astype(np.float64)
        noise = np.abs(ref32 - ref)
        err = np.abs(got - ref)
        l2_bar = max(rtol * np.linalg.norm(ref), 3.0 * np.linalg.norm(noise))
        record(fix, what, name, np.linalg.norm(err) / max(np.linalg.norm(ref), 1e-300), l2_bar / max(np.linalg.norm(ref), 1e-300),
               "L2 distance to the float64 gradient / its norm",
               np.linalg.norm(noise) / max(np.linalg.norm(ref), 1e-300))
        assert np.linalg.norm(err) <= l2_bar + 1e-30, (
            f"{what} grad {name}: L2 distance to the float64 gradient {np.linalg.norm(err):.3e} > bar {l2_bar:.3e} "
            f"(reference fp32 distance {np.linalg.norm(noise):.3e})")
        tol = rtol * (np.abs(ref) + scale) + 4.0 * noise.max()
        bad = err > tol
        assert not bad.any(), (
            f"{what} grad {name}: {bad.sum()}/{bad.size} entries off, max err {err.max():.3e} "
            f"(scale {scale:.3e}, reference fp32 max deviation {noise.max():.3e})")
        return
    err = np.abs(got - ref32)
    tol = rtol * (np.abs(ref32) + scale)
    record(fix, what, name, float((err / (np.abs(ref32) + scale)).max()), rtol, "worst entry vs reference fp32, relative to abs(ref) + rms")
    bad = err > tol
    assert not bad.any(), (
        f"{what} grad {name}: {bad.sum()}/{bad.size} entries off, max err {err.max():.3e} "
        f"(scale {scale:.3e}, rtol {rtol})")
    gn = float(np.linalg.norm(flat))
    assert abs(gn - norm) <= rtol * max(norm, 1e-30) * 4 + 1e-30, f"{what} grad-norm {name}: {gn} vs {norm}"


# ---- achieved errors next to the bars in force (written out by tests/conftest.py at session end) ----
REPORT = []


def record(fix, what, name, achieved, bar, note="", ref_noise=None):
    """One line of the parity report: which fixture / quantity, the error reached and the bar it
    was held to (both relative); ref_noise: how far the reference's own fp32 run is from float64 in
    the same measure, where the fixture holds both."""
    tag = fix if isinstance(fix, str) else os.path.splitext(os.path.basename(getattr(getattr(fix, "zip", None), "filename", "") or "?"))[0]
    REPORT.append((tag, what, name, float(achieved), float(bar), note, None if ref_noise is None else float(ref_noise)))
