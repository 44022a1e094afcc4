"""Developer tool (GPU box): per-block timeline of the blend backward from the trace build (scripts/k16_trace_build.sh):
BRUSH_HIP_LIB=brush_amd/variants/libbrush_hip_trc.so python scripts/k17_trace.py [workload]"""
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
v_out = torch.full((h, w, 4), 1e-6, device=dev)
for _ in range(4):
    ba.render_splats_bwd(spl, cam, (w, h), (0, 0, 0), v_out, ctx=ctx)
torch.cuda.synchronize()
lib = C.CDLL(_ffi.LIB_PATH)
nb = 32768
buf = np.zeros(nb * 4, np.uint64)
assert lib.bh_debug_k17_trace(buf.ctypes.data_as(C.c_void_p), C.c_ulonglong(nb * 32)) == 0
tr = buf.reshape(nb, 4)
idx = np.nonzero(tr[:, 0] > 0)[0]
tr = tr[idx]
t0 = tr[:, 0].astype(np.int64); t1 = np.maximum(tr[:, 1].astype(np.int64), t0)
base = t0.min(); t0 -= base; t1 -= base
jobs = tr[:, 2].astype(np.int64); ent = (tr[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64); segn = (tr[:, 3] >> np.uint64(32)).astype(np.int64)
us = 1e-2
print("%s: %d blocks, span %.1f us, jobs %d, entries %.2f M" % (wl, len(idx), t1.max() * us, jobs.sum(), ent.sum() / 1e6))
d = (t1 - t0) * us
print("block duration mean %.1f max %.1f us; start: median %.1f, p90 %.1f, max %.1f us; jobs per block mean %.2f max %d" % (d.mean(), d.max(), np.median(t0) * us, np.percentile(t0, 90) * us, t0.max() * us, jobs.mean(), jobs.max()))
span = t1.max()
nbk = int(span * us / 20) + 1
occ = []
for b in range(nbk):
    lo_, hi_ = b * 20 / us, (b + 1) * 20 / us
    occ.append((np.minimum(t1, hi_) - np.maximum(t0, lo_)).clip(min=0).sum() / (20 / us))
print("blocks in flight per 20 us bucket:", " ".join("%d" % o for o in occ))
band = idx & 7
print("last end per XCD band (us):", " ".join("%.0f" % (t1[band == b].max() * us) for b in range(8)))
print("busy block-us per XCD band:", " ".join("%.0f" % (d[band == b].sum()) for b in range(8)))
print("entries per XCD band (k):", " ".join("%.0f" % (ent[band == b].sum() / 1e3) for b in range(8)))
order = np.argsort(-t1)[:8]
print("last finishers (block, start, end, jobs, entries):", [(int(idx[i]), round(t0[i] * us, 1), round(t1[i] * us, 1), int(jobs[i]), int(ent[i])) for i in order])
hw = np.zeros(nb, np.uint64)
assert lib.bh_debug_k17_hw(hw.ctypes.data_as(C.c_void_p), C.c_ulonglong(nb * 8)) == 0
hw = hw[idx]
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64); xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
wave = hwid & 0xF; simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
took = jobs > 0
print("wave slot ids used by blocks that took a job:", np.bincount(wave[took], minlength=10).tolist())
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("distinct (xcc, se, sh, cu):", len(np.unique(key)), " distinct SIMDs:", len(np.unique(key * 4 + simd)))
print("by start time (20 us bins): n blocks with a job, mean entries, mean duration us, us per entry, share with segment > 0")
for b in range(int(t0.max() * us / 20) + 1):
    m = took & (t0 * us >= b * 20) & (t0 * us < (b + 1) * 20)
    if m.sum():
        print("   start %3d-%3d us: n %5d  entries %6.1f  dur %6.1f  us/entry %.3f  seg>0 %.2f" % (b * 20, b * 20 + 20, m.sum(), ent[m].mean(), d[m].mean(), d[m].sum() / max(ent[m].sum(), 1), (segn[m] > 0).mean()))
