"""The entry-point smoke run (one small conv-encoder step checked against the oracle) as a test."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

pytestmark = pytest.mark.gpu


def test_entry_point_smoke_runs():
    import __graft_entry__
    __graft_entry__.smoke()
