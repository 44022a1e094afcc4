"""bench.py --gpus N without a launcher must start N ranks itself (the flag was inert), and must
refuse — not print an N=1 line — when fewer than N devices are visible.  CPU: BH_BENCH_DRYRUN=1 swaps the
measured body for a gloo rendezvous, everything else (argument handling, self re-exec through
torch.distributed.run on 127.0.0.1, one JSON line from rank 0) is the real code path."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"BH_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["steps"] == 3 and d["warmup"] == 1


def test_gpus_2_on_a_box_without_two_devices_fails_loudly():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has two devices")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)
    assert not any(l.startswith("{") for l in r.stdout.splitlines()), "no bench line may be printed for fewer ranks than requested"


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "1"], {"BH_BENCH_DRYRUN": "1", "WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
