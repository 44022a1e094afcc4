"""Developer tool: host-side profile (cProfile) of the tile-partitioned train step on a 1-rank process group."""
import cProfile, math, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import brush_amd as ba
from brush_amd import synth
scene, w, h = synth.config_scene("1m_1080p", 0)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=(0, 0, 0, 1), fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
splats = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=7).view(np.int32)).to(dev)
batch = ba.SceneBatch(gt, cam.uniforms((w, h)))
tr = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=5.0, process_group=dist.group.WORLD, partition=sys.argv[1] if len(sys.argv) > 1 else "tiles")
for _ in range(5):
    tr.step(batch, splats)
torch.cuda.synchronize()
ts = []
pr = cProfile.Profile(); pr.enable()
for _ in range(24):
    t0 = time.perf_counter(); tr.step(batch, splats); ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
pr.disable()
print("per-step host ms:", " ".join("%.2f" % t for t in ts))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
dist.destroy_process_group()
