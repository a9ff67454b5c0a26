"""The driver's command, end to end on a real MI355X: `python bench.py` prints ONE JSON line that honours the contract
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype /
data / config) with the `roofline` and `cpu_baseline` objects, a parity figure for the plan it timed, and numbers
that are consistent with one another."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_default_command_prints_one_contract_line():
    env = dict(os.environ)
    env.pop("PLANER_HIP_STREAMS", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-iters", "1"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_rel_err"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and "synthetic" in d["data"]
    assert d["unit"] == "images/sec" and "ResNet-18" in d["metric"]
    assert d["config"]["workload"].startswith("resnet18") or "ResNet-18" in d["config"]["workload"]
    assert not any(k in d["config"] for k in ("model", "seq_len"))
    # value = images / time, both in the line
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]
    assert d["value"] > 10000                                   # an MI355X, not a fallback
    assert d["parity_rel_err"] <= 1e-4
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "mfma" and rf["peak"] == 157.3 and rf["unit"] == "TFLOP/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1
    assert abs(rf["achieved"] - rf["executed_flops_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e12) <= 0.01 * rf["achieved"]
    assert 0 < rf["whole_forward_timed_run"]["mfma_util"] < 1
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["unit"] == "images/sec" and 0 < cb["value"] < d["value"] and cb["cores"] >= 1
    # every conv of the timed plan is on record with its kernel family and launch plan
    algos = d["config"]["algos"]
    assert len([a for a in algos if a["layer"].endswith("conv+") or "conv" in a["layer"]]) >= 20 and all(a["plan"] for a in algos)
    for h in d["roofline_hbm"]:
        assert h["unit"] == "GB/s" and h["peak"] == 8000.0 and 0 < h["frac"] < 1
