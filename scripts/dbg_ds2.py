import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import brush_amd as ba
from brush_amd import synth, _ffi
dev = torch.device("cuda:0")
scene, w, h = synth.config_scene("1m_1080p", 0)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
spl = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
for _ in range(5):
    img, aux = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward)
torch.cuda.synchronize()
lib = _ffi.load()
buf = (C.c_ulonglong * (256 * 8))()
lib.bh_debug_dsort_clocks.restype = C.c_int
print("rc", lib.bh_debug_dsort_clocks(buf))
a = np.array(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
t0 = a[:255, 0].min()
ok = a[:255, 6] > 0
print("blocks", ok.sum(), "wall_clock ticks are 100 MHz (10 ns)")
rel = (a[:255, :6] - t0) * 0.01  # us
sz = a[:255, 6]
order = np.argsort(-sz)[:6]
for b in list(order) + [0, 100]:
    print("bucket %3d size %6d  start %6.1f  prefix %6.1f  loaded %6.1f  sorted %6.1f  written %6.1f  scanned %6.1f us" % ((b, sz[b]) + tuple(rel[b])))
print("latest finish %.1f us, earliest start %.1f, latest start %.1f" % (rel[ok, 5].max(), rel[ok, 0].min(), rel[ok, 0].max()))
