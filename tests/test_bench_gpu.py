"""bench.py end to end at a small size: one JSON line with the contract's fields."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--batch", "512", "--also", "", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "samples/s"
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    # the line says what arithmetic the step's big products run in, not just the storage type
    assert d["dtype"].startswith("f32 (first-layer products: f16x2 forward / f16x2 weight gradients")
    assert "f16x2" in d["config"]["arithmetic"] and d["config"]["first_layer_weight_gradient"] == "f16x2"
    assert "workload" in d["config"] and d["value"] > 0
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert d["script_batch"]["graph"]["value"] > 0 and d["n68"]["value"] > 0
    # the same workload without the symmetric-input promise is reported next to the headline
    assert d["full_first_layer"]["value"] > 0
    assert d["config"]["symmetric_first_layer"] is True and len(d["config"]["workload"]) <= 128
    assert d["config"]["full_first_layer_value"] == d["full_first_layer"]["value"]
    assert not d["roofline"]["kernel"].startswith("library_")          # a hand-written kernel
    # the same step at the finest and at the fastest arithmetic, the reference's own loss kernels, and the
    # cost of the collective calls on a one-rank RCCL group
    assert d["strict"]["value"] > 0 and d["fast_wgrad"]["value"] > 0
    assert d["fused_loss_kernels"]["B=16384"]["mse_fwdbwd"]["frac"] > 0
    assert d["exchange_rehearsal"].get("ms_per_step_with", 0) > 0, d["exchange_rehearsal"]
