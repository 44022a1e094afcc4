import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import brush_amd as ba
from brush_amd import synth
dev = torch.device("cuda:0")
scene, w, h = synth.config_scene("1m_1080p", 0)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
spl = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
img, aux = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward)
d = aux.depths_sorted.cpu().numpy()
k = d.view(np.uint32).astype(np.int64)
kmin, kmax = k.min(), k.max()
r = kmax - kmin; s = 0
while (r >> s) > 254: s += 1
dig = (k - kmin) >> s
cnt = np.bincount(dig, minlength=255)
print("nv", aux.num_visible, "kmin %x kmax %x shift %d" % (kmin, kmax, s), "bucket sizes: max", cnt.max(), "mean", cnt.mean(), "n>8192:", (cnt > 8192).sum(), "n>4096", (cnt > 4096).sum())
print(np.sort(cnt)[-10:])
