"""Developer tool (GPU box): per-block timeline of the forward blend from an ad-hoc trace build (see profiles/EXPERIMENTS.md, split tiles):
BRUSH_HIP_LIB=brush_amd/variants/libbrush_hip_trc.so python scripts/k16_trace2.py [workload]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brush_amd as ba
from brush_amd import synth, _ffi

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "1m_1080p_centered"
sc, w, h = synth.config_scene(wl, 0)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
ctx = ba.get_context(dev)
for _ in range(5):
    ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, copy=False)
torch.cuda.synchronize()
lib = C.CDLL(_ffi.LIB_PATH)
nb = 16384
buf = np.zeros(nb * 4, np.uint64)
assert lib.bh_debug_k16_trace(buf.ctypes.data_as(C.c_void_p), C.c_ulonglong(nb * 32)) == 0
tr = buf.reshape(nb, 4)
idx = np.nonzero(tr[:, 1] > 0)[0]
tr = tr[idx]
t0 = tr[:, 0].astype(np.int64); t1 = tr[:, 1].astype(np.int64)
base = t0.min(); t0 -= base; t1 -= base
q = (tr[:, 2] >> np.uint64(32)).astype(np.int64); tile = (tr[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
listed = (tr[:, 3] >> np.uint64(32)).astype(np.int64); walked = (tr[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
us = 1e-2   # wall_clock64 ticks at 100 MHz
print("%s: %d blocks traced, launch span %.1f us (first start to last end before bookkeeping)" % (wl, len(idx), (t1.max() - t0.min()) * us))
print("split blocks %d, whole-tile blocks %d" % ((q > 0).sum(), (q == 0).sum()))
order = np.argsort(-t1)
print("last finishers: block  start_us  end_us  dur_us  quadrant  tile  listed  walked  us/entry")
for i in order[:16]:
    d = (t1[i] - t0[i]) * us
    print("   %6d  %7.1f  %7.1f  %7.1f  %d  %5d  %6d  %6d  %.3f" % (idx[i], t0[i] * us, t1[i] * us, d, q[i], tile[i], listed[i], walked[i], d / max(walked[i], 1)))
for name, m in (("split quadrant waves", q > 0), ("whole tiles", q == 0)):
    if m.sum() == 0:
        continue
    d = (t1[m] - t0[m]) * us
    wk = walked[m]
    print("%s: n %d, walked mean %.0f max %d, duration mean %.1f max %.1f us, last end %.1f us, late starters (start > 20 us): %d"
          % (name, m.sum(), wk.mean(), wk.max(), d.mean(), d.max(), t1[m].max() * us, (t0[m] * us > 20).sum()))
    big = wk >= 256
    if big.sum():
        print("   blocks that walked >= 256 entries: n %d, us per entry mean %.3f min %.3f max %.3f" % (big.sum(), (d[big] / wk[big]).mean(), (d[big] / wk[big]).min(), (d[big] / wk[big]).max()))
# occupancy over time: waves in flight per 10 us bucket (8192 slots), and when each XCD (band) finished
span = t1.max()
nb_ = int(span * us / 10) + 1
occ = np.zeros(nb_)
for b in range(nb_):
    lo_, hi_ = b * 10 / us, (b + 1) * 10 / us
    occ[b] = (np.minimum(t1, hi_) - np.maximum(t0, lo_)).clip(min=0).sum() / (10 / us)
print("waves in flight per 10 us bucket:", " ".join("%d" % o for o in occ))
band = idx & 7
print("last end per XCD band (us):", " ".join("%.0f" % (t1[band == b].max() * us) for b in range(8) if (band == b).any()))
print("busy wave-us per XCD band:", " ".join("%.0f" % ((t1[band == b] - t0[band == b]).sum() * us) for b in range(8) if (band == b).any()))
