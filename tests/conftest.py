import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_fixture(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def fixture_params(fix, dtype=None):
    """Rebuild the parameter dict a fixture was generated with (tests/golden/recipe.py)."""
    import torch
    import recipe
    shapes = {n: tuple(int(s) for s in sh.split(",")) if sh else () for n, sh in zip(fix["param_names"], fix["param_shapes"])}
    bounds = {n: float(b) for n, b in zip(fix["param_names"], fix["param_bounds"])}
    return recipe.build_params(shapes, bounds, dtype or torch.float32)


@pytest.fixture(scope="session")
def golden():
    return load_fixture


def pytest_sessionfinish(session, exitstatus):
    """Achieved errors next to the bars in force, for every gradient check that went through
    recipe.check_grad / recipe.record in this session (gpurun_out/parity_report.md; parity_report_cpu.md for a
    session without a GPU, which only checks the oracle)."""
    import recipe
    import torch
    if not recipe.REPORT:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    groups = {}
    for tag, what, name, achieved, bar, note, noise in recipe.REPORT:
        groups.setdefault((tag, what, note), []).append((name, achieved, bar, noise))
    with open(os.path.join(out, "parity_report.md" if torch.cuda.is_available() else "parity_report_cpu.md"), "w") as f:
        f.write("| fixture / case | path | checks | worst quantity (by achieved / bar) | achieved | bar in force | "
                "reference's own fp32 run | measure |\n|---|---|---|---|---|---|---|---|\n")
        for (tag, what, note), rows in sorted(groups.items()):
            name, achieved, bar, noise = max(rows, key=lambda r: r[1] / max(r[2], 1e-300))
            f.write(f"| {tag} | {what} | {len(rows)} | {name} | {achieved:.2e} | {bar:.2e} | "
                    f"{'' if noise is None else format(noise, '.2e')} | {note} |\n")
