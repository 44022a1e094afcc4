"""bench.py's pre-timing self-check of the N > 1 exchange path: the library's RCCL communicator — the
bench's default for N > 1 — has to prove itself (all-reduce of ones == world, two steps bit-identical across ranks and equal
to the torch.distributed path) before a single step is timed; a broken collective must be caught and the run fall back to the
hook, loudly.  One GPU: BH_FORCE_PG=1 builds the process group and a one-rank communicator, so the whole check runs."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BH_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--splats", "50000"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0]), p.stderr


def test_selfcheck_passes_and_is_recorded():
    line, _ = _bench({})
    sc = line["exchange"]["selfcheck"]
    assert sc["native"] == "ok" and sc["torch"] == "ok" and sc["timed_path"] == "native" and line["exchange"]["comm"] == "native"
    assert sc["paths_frac_beyond_1e-6"] <= 2e-2 and sc["seconds"] < 60


def test_a_broken_allreduce_is_caught_and_the_run_falls_back():
    # fault injection lives in the -DBH_TEST_HOOKS build only; BRUSH_HIP_LIB points the bench's binding at it
    line, err = _bench({"BH_BREAK_ALLREDUCE": "1", "BRUSH_HIP_LIB": os.path.join(ROOT, "brush_amd", "libbrush_hip_testhooks.so")})
    sc = line["exchange"]["selfcheck"]
    assert sc["native"].startswith("FAILED") and "all-reduce of" in sc["native"]
    assert sc["torch"] == "ok" and sc["timed_path"] == "torch" and line["exchange"]["comm"] == "torch"
    assert "FAILED its self-check" in err
