"""Developer tool (GPU box): per-tile timeline of the forward blend from a BH_K16_TRACE variant build
(scripts/ab.sh build-one k16trace rasterize "-DBH_K16_TRACE"): where does the launch's time go — imbalance between SIMDs,
dispatch, or the tiles themselves?   BRUSH_HIP_LIB=brush_amd/variants/libbrush_hip_k16trace.so python scripts/k16_trace.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brush_amd as ba
from brush_amd import synth, _ffi

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "1m_1080p"
sc, w, h = synth.config_scene(wl, 0)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
ctx = ba.get_context(dev)
for _ in range(5):
    ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, copy=False)
torch.cuda.synchronize()
lib = C.CDLL(_ffi.LIB_PATH)
T = ((w + 15) // 16) * ((h + 15) // 16)
nb = ((T + 7) // 8) * 8
buf = np.zeros(nb * 4, np.uint64)
rc = lib.bh_debug_k16_trace(buf.ctypes.data_as(C.c_void_p), C.c_ulonglong(nb * 4))
assert rc == 0
tr = buf.reshape(nb, 4)
tr = tr[tr[:, 1] > 0]
t0, t1 = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64)
base = t0.min()
t0 -= base; t1 -= base
hw = tr[:, 2] & np.uint64(0xFFFFFFFF)
xcc = (tr[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
work = (tr[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
# HW_ID (gfx9): wave 0-3, simd 4-5, pipe 6-7, cu 8-11, sh 12, se 13-15 (+xcc)
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64)
se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
clk = 100e6   # wall_clock64: 100 MHz constant clock
print("tiles", len(tr), "span us %.1f" % ((t1.max()) / clk * 1e6), "distinct SIMDs", len(np.unique(key)))
dur = (t1 - t0) / clk * 1e6
print("tile duration us: mean %.1f p50 %.1f p90 %.1f max %.1f" % (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
print("tile start us: p50 %.1f p90 %.1f max %.1f" % (np.median(t0) / clk * 1e6, np.percentile(t0, 90) / clk * 1e6, t0.max() / clk * 1e6))
print("corr(work, duration) %.3f" % np.corrcoef(work, dur)[0, 1], "work mean %.1f max %d" % (work.mean(), work.max()))
# per SIMD: number of tiles, sum of work, last end
ks, inv = np.unique(key, return_inverse=True)
cnt = np.bincount(inv)
wsum = np.bincount(inv, weights=work)
last = np.zeros(len(ks)); np.maximum.at(last, inv, t1 / clk * 1e6)
first = np.full(len(ks), 1e18); np.minimum.at(first, inv, t0 / clk * 1e6)
print("per SIMD: tiles min %d mean %.2f max %d | work sum mean %.0f max %.0f (max/mean %.2f) | last end us p50 %.1f max %.1f" %
      (cnt.min(), cnt.mean(), cnt.max(), wsum.mean(), wsum.max(), wsum.max() / wsum.mean(), np.median(last), last.max()))
print("corr(SIMD work sum, SIMD last end) %.3f" % np.corrcoef(wsum, last)[0, 1])
# how many tiles are running at time t
for frac in (0.1, 0.25, 0.5, 0.75, 0.9):
    t = frac * t1.max()
    print("  at %.0f %% of the span: %d tiles running, %d not started" % (frac * 100, int(((t0 <= t) & (t1 > t)).sum()), int((t0 > t).sum())))
