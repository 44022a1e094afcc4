import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# per-tile depth cuts (include/brush_hip.h: bh_set_list_cut_threshold) are only applied to frames with >= 1.5 M intersections by
# default; the parity suite's scenes are small, and it is the mechanism that has to be covered: every context of this session
# (and of the processes it spawns) cuts whatever the frame's size.  tests/test_gpu_sliced.py checks the default threshold itself.
os.environ.setdefault("BH_CUT_MIN_PAIRS", "0")
# ... and keeps cutting however little a cut saves: the product lets a view whose last cut frame listed > 90 % of its pairs render
# complete lists for a while (option auto_exact_share) — the suite's small scenes barely saturate, so nearly every frame would.
# tests/test_gpu_options.py::test_views_render_complete_lists_when_cuts_save_nothing covers the product default.
# (brush_amd/host.py turns these into bh_set_option calls on every Context: the library itself reads no environment variable)
os.environ.setdefault("BH_OPTIONS", "auto_exact_share=0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """A fresh clone carries no binaries (*.so is git-ignored): build libbrush_hip.so (hipcc cross-compiles without a GPU)
    and the oracle before the first test.  `make` is a no-op when everything is up to date."""
    import glob
    lib = os.path.join(ROOT, "brush_amd", "libbrush_hip.so")
    orc = os.path.join(ROOT, "oracle", "libbrush_oracle.so")
    srcs = glob.glob(os.path.join(ROOT, "brush_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        glob.glob(os.path.join(ROOT, "oracle", "*.cpp"))
    # (2 s of slack: a snapshot copy may stamp binaries and sources in arbitrary order within the same moment)
    stale = not (os.path.exists(lib) and os.path.exists(orc)) or max(os.path.getmtime(f) for f in srcs) > min(os.path.getmtime(lib), os.path.getmtime(orc)) + 2.0
    if stale:
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import bo
    bo.build()
    return bo


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return torch.device("cuda:0")
