"""`bench.py --gpus N` started without torch.distributed.run: it must start N rank processes itself
(before anything touches a GPU), hand rank 0's JSON line through, fail loudly when fewer than N
GPUs are visible, and end every rank when one of them dies.  The ranks of these tests meet over
gloo and stop before the GPU part (SFM_BA_BENCH_SELFTEST)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=240)


def test_launcher_spawns_ranks_strong_and_weak():
    r = _run(["--gpus", "3", "--scaling", "strong"], SFM_BA_BENCH_SELFTEST="1", SFM_BA_BENCH_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["rank_sum"] == 6.0
    assert out["scaling"] == "strong" and out["workload"] == "venice1800"
    r = _run(["--gpus", "2"], SFM_BA_BENCH_SELFTEST="1", SFM_BA_BENCH_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["workload"] == "c4_500x300k"


def test_launcher_refuses_more_ranks_than_gpus():
    import torch

    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 1 if have else 2)], SFM_BA_BENCH_SELFTEST="1")
    assert r.returncode != 0
    assert "visible GPUs" in r.stderr and not r.stdout.strip()


def test_launcher_ends_all_ranks_when_one_dies():
    r = _run(["--gpus", "2"], SFM_BA_BENCH_SELFTEST="fail1", SFM_BA_BENCH_BACKEND="gloo")
    assert r.returncode == 7
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2"], SFM_BA_BENCH_SELFTEST="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
