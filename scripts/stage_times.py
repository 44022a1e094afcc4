"""Developer tool: per-stage HIP-event times of the train step at 1M/1080p for the
library selected by BRUSH_HIP_LIB (default: the in-tree build).  One line of output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brush_amd as ba
from brush_amd import synth

deg = int(os.environ.get("SH_DEGREE", "0"))
steps = int(os.environ.get("STEPS", "15"))
dev = torch.device("cuda:0")
scene, w, h = synth.config_scene(os.environ.get("WORKLOAD", "1m_1080p"), deg)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
splats = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=7).view(np.int32)).to(dev)
batch = ba.SceneBatch(gt, cam.uniforms((w, h)))
ctx = ba.get_context(dev)
tr = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=5.0, ctx=ctx)
for _ in range(4):
    tr.step(batch, splats)
torch.cuda.synchronize()
prof = os.environ.get('PROFILE', '1') == '1'
ctx.profile(prof); ctx.profile_fetch()
import time
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(batch, splats)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
st = ctx.profile_fetch()
tot = sum(ms / max(c, 1) for ms, c in st.values())
print("%s wall %.3f kern %.3f | " % (os.path.basename(os.environ.get("BRUSH_HIP_LIB", "default")), wall, tot) +
      " ".join("%s %.3f" % (k[:14], ms / max(c, 1)) for k, (ms, c) in st.items()))
