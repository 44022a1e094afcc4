"""Developer tool: distribution of per-tile work (splats actually blended before every pixel saturates) at the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brush_amd as ba
from brush_amd import synth
dev = torch.device("cuda:0")
scene, w, h = synth.config_scene("1m_1080p", 0)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
spl = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
img, aux = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward)
to = aux.tile_offsets.cpu().numpy().astype(np.int64)
work = to[:, 1] - to[:, 0]
print("tiles", len(work), "total", work.sum(), "mean %.1f" % work.mean(), "std %.1f" % work.std(), "min", work.min(), "max", work.max(),
      "p50 %d p90 %d p99 %d" % tuple(np.percentile(work, [50, 90, 99])))
# greedy in-order dispatch onto S slots (what the hardware does) vs longest-first
def makespan(order, slots):
    import heapq
    hq = [0] * slots
    for t in order:
        x = heapq.heappop(hq); heapq.heappush(hq, x + t)
    return max(hq)
for slots in (1024 * 4, 1024 * 2, 1024):
    band = work.copy()
    print("slots", slots, "ideal %.1f" % (work.sum() / slots), "in-order", makespan(list(work), slots), "longest-first", makespan(sorted(work, reverse=True), slots))
