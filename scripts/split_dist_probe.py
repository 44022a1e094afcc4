import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import brush_amd as ba
from brush_amd import synth
dev = torch.device("cuda:0")
sc, w, h = synth.config_scene("1m_1080p_centered", 0)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
ctx = ba.Context(dev)
v_out = torch.full((h, w, 4), 1e-6, device=dev)
for _ in range(3):
    res = ba.render_splats_bwd(spl, cam, (w, h), (0, 0, 0), v_out, ctx=ctx)
to = res["aux"].tile_offsets.to(torch.int64)
work = (to[:, 1] - to[:, 0]).clamp(min=0).cpu().numpy()
print("tiles", work.size, "mean", work.mean(), "nonzero", (work > 0).sum())
for thr in (128, 200, 256, 300, 400, 500, 600, 700, 800):
    print("work >=", thr, ":", int((work >= thr).sum()))
per = (work.size + 7) // 8
for b in range(8):
    band = np.sort(work[b * per:(b + 1) * per])[::-1]
    print("band", b, "mean %.0f" % band.mean(), "top", band[:4].tolist(), "rank64", int(band[64]) if band.size > 64 else None, "rank128", int(band[128]), "rank256", int(band[256]), ">=256:", int((band >= 256).sum()), ">=2.5mean", int((band >= max(256, 2.5 * band.mean())).sum()))
