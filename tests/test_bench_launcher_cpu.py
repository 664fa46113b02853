"""`python bench.py --gpus N` must launch its own ranks (the driver's command shape) - rehearsed on CPU: world-size-2 gloo
ranks spawned by bench.py's launcher, the contract's timing harness and the static-shape all_gather on a stub step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bench_gpus2_launches_its_own_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]            # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"


def test_bench_launcher_propagates_child_failure():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--stub"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=_env(L2S_BENCH_STUB_FAIL_RANK="1"))
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_single_rank_stub_needs_no_launcher():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--stub"], capture_output=True,
                         text=True, timeout=120, cwd=ROOT, env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_bench_mixed_gpus2_ragged_bucket_gather():
    """BASELINE configs[4] at N = 2 on CPU: the real length list, dealing and buckets; every bucket round ends in the ragged
    padded all_gather (different clip counts / lengths per rank), whose rows the stub checks against the dealing."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub",
                          "--mixed", "--clips", "37", "--bucket", "8"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["clips_total"] == 74 and d["config"]["buckets_rank0"] == 5 and d["value"] > 0


def test_mixed_frame_budget_buckets_are_the_same_on_every_rank():
    """bench.mixed_buckets with a padded-frame budget (--bucket 0 --bucket-frames F): the cut is made on the global length list in
    whole serpentine rounds, so every rank holds the same number of buckets with the same clip counts (the per-bucket all_gather
    needs that), every clip is in exactly one bucket, and no bucket but a merged last one exceeds the budget."""
    import bench
    for world in (1, 2, 8):
        per_rank = [bench.mixed_buckets(40, 0, r, world, 4000) for r in range(world)]
        sizes = [[len(b) for b in pr[2]] for pr in per_rank]
        assert all(s == sizes[0] for s in sizes) and len(sizes[0]) > 1
        for lengths_all, my_lens, buckets in per_rank:
            assert sorted(x for b in buckets for x in b) == sorted(my_lens) and len(my_lens) == 40
            assert all(b == sorted(b, reverse=True) for b in buckets)
            assert all(len(b) * max(b) <= 4000 for b in buckets[:-1])
        assert sorted(x for pr in per_rank for x in pr[1]) == sorted(int(v) for v in per_rank[0][0])
    assert [len(b) for b in bench.mixed_buckets(40, 16, 0, 1, 4000)[2]] == [16, 16, 8]      # --bucket N: fixed clip counts
