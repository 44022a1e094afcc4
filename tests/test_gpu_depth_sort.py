"""The forward's fused depth ordering (brush_amd/csrc/depth_sort.hip: one most-significant-digit split on the visible key range +
one kernel that finishes every bucket and writes the scan of the tile counts) must produce exactly what the generic stable
32-bit radix sort + prefix sum produce (BH_GENERIC_DEPTH_SORT=1 selects those): same permutation (ties in splat-id order),
same sorted depths, same cum_tiles_hit — with the split at the previous frame's depth quantiles (the default; here the
previous frame is a DIFFERENT scene every other time) and with the linear split (option dsort_splitters=0) — on ordinary scenes and on depth distributions shaped to break a range split:
all splats at one depth, two tight clusters plus a far outlier (one bucket holds nearly everything: the chunked many-pass
path), a huge dynamic range, a single visible splat, nothing visible."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from brush_amd import synth
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scenes():
    rng = np.random.default_rng(5)
    out = {}
    base = synth.make_scene(60_000, 0x5D, sh_degree=0, log_scale_range=(math.log(0.01), math.log(0.08)))
    out["ordinary"] = base
    s = {k: v.copy() for k, v in base.items()}
    s["transforms"][:, 2] = np.float32(5.0)                      # every splat at the same depth: one key, ties only
    out["one_depth"] = s
    big = synth.make_scene(200_000, 0x5E, sh_degree=0, log_scale_range=(math.log(0.01), math.log(0.05)))
    s = {k: v.copy() for k, v in big.items()}
    z = np.where(rng.random(200_000) < 0.5, 5.0 + rng.random(200_000) * 1e-3, 7.0 + rng.random(200_000) * 1e-3).astype(np.float32)
    s["transforms"][:, 2] = z
    s["transforms"][0, 0:3] = (0.0, 0.0, 5.0e5)                  # one far outlier stretches the range: each cluster (~40 k visible) lands in ONE bucket,
    s["transforms"][0, 7:10] = math.log(2.0e3)                   # beyond what the LDS-resident path holds: the chunked many-pass path
    out["clusters_and_outlier"] = s
    s = {k: v.copy() for k, v in base.items()}
    s["transforms"][:, 2] = np.exp(rng.uniform(math.log(0.02), math.log(5e4), 60_000)).astype(np.float32)   # 21 octaves
    s["transforms"][:, 0:2] *= (s["transforms"][:, 2:3] / base["transforms"][:, 2:3])
    s["transforms"][:, 7:10] += np.log(s["transforms"][:, 2:3] / base["transforms"][:, 2:3])
    out["huge_range"] = s
    s = {k: v[:300].copy() for k, v in base.items()}
    s["transforms"][1:, 2] = -3.0                                # one visible splat, the rest behind the camera
    out["single_visible"] = s
    s = {k: v[:5000].copy() for k, v in base.items()}
    s["transforms"][:, 2] = -3.0                                 # nothing visible
    out["none_visible"] = s
    return out


_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import brush_amd as ba, util
from brush_amd import synth
import test_gpu_depth_sort as T
dev = torch.device("cuda:0")
cp = synth.default_camera_params(640, 360)
res = {}
# every scene twice in a row, all in ONE context: the first frame of a scene splits its keys at the quantiles of the scene BEFORE it
# (nothing in common with its own depths), the second at its own
for name, sc in T._scenes().items():
  for rep in ("", ".again"):
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    img, aux = ba.render_splats(spl, util.hip_camera(ba, cp), (640, 360), (0, 0, 0), ba.RasterPass.Backward)
    nv = aux.num_visible
    res[name + rep + ".nv"] = np.array([nv, aux.num_intersections])
    res[name + rep + ".gfc"] = util.u32(aux.global_from_compact_gid)[:nv]
    res[name + rep + ".depths"] = aux.depths_sorted.cpu().numpy()[:nv]
    res[name + rep + ".cum"] = util.u32(aux.cum_tiles_hit)[:nv]
    res[name + rep + ".isect"] = util.u32(aux.compact_gid_from_isect)
    res[name + rep + ".img"] = img.cpu().numpy()
np.savez(sys.argv[1], **res)
"""


def test_fused_depth_order_equals_generic_sort_and_scan(dev, tmp_path):
    outs = {}
    for mode in ("fused", "linear", "generic"):
        env = dict(os.environ)
        env.pop("BH_GENERIC_DEPTH_SORT", None)
        env.pop("BH_OPTIONS", None)
        if mode == "generic":
            env["BH_GENERIC_DEPTH_SORT"] = "1"
        if mode == "linear":
            env["BH_OPTIONS"] = "dsort_splitters=0"   # the linear split of the key range for every frame
        f = str(tmp_path / (mode + ".npz"))
        r = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, ROOT), f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[mode] = np.load(f)
    a, b = outs["fused"], outs["generic"]
    for other in (outs["generic"], outs["linear"]):
        assert sorted(a.files) == sorted(other.files)
        for k in a.files:
            assert np.array_equal(a[k], other[k]), k
    for k in a.files:
        if ".again" in k:
            assert np.array_equal(a[k], a[k.replace(".again", "")]), k
    # the scenes do what they were built for
    assert a["ordinary.nv"][0] > 20_000 and a["one_depth.nv"][0] > 20_000 and a["clusters_and_outlier.nv"][0] > 60_000
    assert a["huge_range.nv"][0] > 15_000 and a["single_visible.nv"][0] == 1 and a["none_visible.nv"][0] == 0
    d = a["one_depth.depths"]
    assert np.all(d == d[0]) and np.all(np.diff(a["one_depth.gfc"].astype(np.int64)) > 0), "ties keep splat-id order"
    for name in ("ordinary", "clusters_and_outlier", "huge_range"):
        assert np.all(np.diff(a[name + ".depths"]) >= 0)


def test_first_frame_of_a_view_sorts_a_sample_first(dev):
    """A frame that finds no splitter table (a view's first frame) sorts every 64th key first and splits at THAT sample's quantiles
    (depth_sort.hip SPL_SAMPLE_STRIDE; frames of >= 131072 splats): on a scene whose depths crowd into a thin shell — the case the
    sample is there for — the first and the second frame of a fresh context equal the generic sort's, order, scan and image."""
    import brush_amd as ba
    n = 300_000
    rng = np.random.default_rng(23)
    sc = synth.make_scene(n, 0x71, sh_degree=0, log_scale_range=(math.log(0.01), math.log(0.04)))
    z = np.where(rng.random(n) < 0.97, 6.0 + rng.random(n) * 0.05, 2.0 + rng.random(n) * 10.0).astype(np.float32)
    f = z / sc["transforms"][:, 2]
    sc["transforms"][:, 0] *= f
    sc["transforms"][:, 1] *= f
    sc["transforms"][:, 2] = z
    cp = synth.default_camera_params(640, 360)
    cam = util.hip_camera(ba, cp)
    got = {}
    for name, opts in (("generic", {"generic_depth_sort": 1}), ("sampled", {}), ("linear", {"dsort_splitters": 0})):
        ctx = ba.Context(dev, options=opts)
        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
        frames = []
        for _ in range(2):
            img, aux = ba.render_splats(spl, cam, (640, 360), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx)
            nv = aux.num_visible
            frames.append((nv, util.u32(aux.global_from_compact_gid)[:nv].copy(), util.u32(aux.cum_tiles_hit)[:nv].copy(), img.cpu().numpy()))
        got[name] = frames
        ctx.close()
    assert got["generic"][0][0] > 131072 // 2
    for name in ("sampled", "linear"):
        for k in range(2):
            a, b = got["generic"][k], got[name][k]
            assert a[0] == b[0], (name, k)
            for x, y in zip(a[1:], b[1:]):
                assert np.array_equal(x, y), (name, k)
