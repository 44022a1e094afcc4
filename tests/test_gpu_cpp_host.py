"""The C++ host-side mirror (include/brush_hip.hpp — the reference's host language is compiled, so the host above the
C ABI exists in C++ too) exercised by a native test program: tests/cpp/test_host.cpp renders / trains / refines /
exports through the header and checks against the oracle it dlopens, with no Python in the loop."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    import __graft_entry__ as g
    g.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s"])
    return os.path.join(ROOT, "tests", "cpp", "test_host")


def test_cpp_host_header_compiles_and_links():
    """CPU: the header and the test program compile (g++, HIP runtime API only) and link against libbrush_hip.so."""
    exe = _build()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_cpp_host_program_passes_on_the_gpu(oracle_lib):
    exe = _build()
    p = subprocess.run([exe, os.path.join(ROOT, "oracle", "libbrush_oracle.so")], cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(p.stdout[-3000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "all C++ host checks passed" in p.stdout
    for name in ("render_vs_oracle[pinhole]", "render_vs_oracle[kb4]", "render_vs_oracle[rt8]", "primitives", "training_refine_ply", "loss_optimizer", "loader_controls_comm", "errors"):
        assert "ok " + name in p.stdout, name
