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
