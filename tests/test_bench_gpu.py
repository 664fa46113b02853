"""bench.py contract on a tiny model: the default (fixed 4-s clips) line and the mixed-length line (BASELINE configs[4])."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_small():
    d = _run("--batch", "4", "--frames", "20", "--steps", "2", "--warmup", "1", "--enc-layers", "2", "--conf-layers", "1",
             "--cpu-clips", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    p = d["parity_vs_oracle"]
    assert p["unit_ids_equal"] == p["unit_ids_compared"] > 0


def test_bench_mixed_lengths_small():
    d = _run("--mixed", "--clips", "12", "--bucket", "4", "--steps", "2", "--warmup", "1", "--enc-layers", "2",
             "--conf-layers", "1")
    assert d["value"] > 0 and d["config"]["clips_per_gpu"] == 12 and d["config"]["hipgraph"] is True
    assert 1.0 <= d["config"]["padding_overhead_rank0"] < 2.0
