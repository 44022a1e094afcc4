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


def test_every_profiled_library_kernel_belongs_to_a_bench_stage():
    """bench.py sums the child run's kernel times by stage through KERNEL_STAGE's name fragments.  A kernel that matches none of them
    silently drops out of `stages` and `kernel_ms_per_step` (round 4: the new tile_bucket_kernel did, 25 us of the tile sort).  Every
    `bh::` kernel in the committed traces of the latest round must match a fragment."""
    import csv
    import glob
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    block = re.search(r"KERNEL_STAGE = \((.*?)\)\)\n", src, re.S).group(0)
    keys = re.findall(r'\("([A-Za-z_0-9<>]+)", "[A-Za-z]+"\)', block)
    assert len(keys) >= 10
    traces = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats*.csv")))
    assert traces, "no committed kernel trace"
    latest = max(int(re.search(r"/r(\d+)", t).group(1)) for t in traces)
    unmatched = set()
    for t in traces:
        if int(re.search(r"/r(\d+)", t).group(1)) != latest:
            continue
        for row in csv.DictReader(open(t)):
            name = row["Name"]
            if not name.startswith(("void bh::", "bh::")):
                continue
            short = name.replace("void bh::", "").replace("bh::", "").split("(")[0]
            if not any(k in short for k in keys):
                unmatched.add(short)
    assert not unmatched, unmatched
