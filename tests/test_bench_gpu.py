"""bench.py contract on a tiny model: the default (fixed 4-s clips) line, the mixed-length line (BASELINE configs[4]) and the
frontend-only line (BASELINE configs[1]); every line carries `roofline` and `cpu_baseline` (SURVEY 8d protocol fields)."""
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
             "--cpu-clips", "2", "--cpu-warm", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert set(c["median_s_per_clip"]) == {"frontend_s", "stage1_s", "vocoder_s"} and "1 warm-up + 2 timed" in c["sample"]
    assert c["median_s_per_clip"]["frontend_s"] < c["median_s_per_clip"]["stage1_s"]
    p = d["parity_vs_oracle"]
    assert p["clips"] == 2 and p["unit_ids_equal"] == p["unit_ids_compared"] > 0


def test_bench_mixed_lengths_small():
    d = _run("--mixed", "--clips", "12", "--bucket", "4", "--steps", "2", "--warmup", "1", "--enc-layers", "2",
             "--conf-layers", "1", "--cpu-clips", "3", "--cpu-warm", "1")
    assert d["value"] > 0 and d["config"]["clips_per_gpu"] == 12 and d["config"]["hipgraph"] is True
    assert d["config"]["streams"] == 3          # the three buckets' graphs replay side by side; the parity sample below reads their outputs
    assert 1.0 <= d["config"]["padding_overhead_rank0"] < 2.0
    r, c, p = d["roofline"], d["cpu_baseline"], d["parity_vs_oracle"]
    assert r["bound"] in ("hbm", "mfma") and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert c["kind"] == "port" and c["value"] > 0 and "timed clips" in c["sample"]
    assert p["clips"] == 3 and p["unit_ids_equal"] == p["unit_ids_compared"] > 0


def test_bench_frontend_stage_small():
    """BASELINE configs[1]: frontend kernels only; the line names the binding roof and reports both GB/s and TFLOP/s."""
    d = _run("--stage", "frontend", "--batch", "4", "--frames", "20", "--steps", "2", "--warmup", "1", "--cpu-clips", "2",
             "--cpu-warm", "1")
    assert "frontend" in d["metric"] and d["value"] > 0 and d["frames_per_sec"] > 0
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["hbm"]["unit"] == "GB/s" and r["hbm"]["achieved"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["dominant_kernel"]["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["parity_rel_max_err_clip0"] < 2e-2
