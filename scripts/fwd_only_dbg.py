import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import brush_amd as ba
from brush_amd import synth
dev = torch.device("cuda:0")
sc, w, h = synth.config_scene("1m_1080p", 0)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
ctx = ba.get_context(dev)
def loop(tag, pass_, sliced):
    for i in range(6):
        ba.render_splats(spl, cam, (w, h), (0, 0, 0), pass_, ctx=ctx, copy=False, sliced=sliced)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ba.render_splats(spl, cam, (w, h), (0, 0, 0), pass_, ctx=ctx, copy=False, sliced=sliced)
    torch.cuda.synchronize()
    print(tag, pass_, "sliced" if sliced else "exact", round((time.perf_counter() - t0) / 50 * 1e3, 4), "ms", "share", ctx.lib.bh_last_list_share(ctx._h), "far", ctx.lib.bh_far_slices_queued(ctx._h))
loop("fresh", ba.RasterPass.Forward, False)
loop("fresh", ba.RasterPass.Forward, True)
if len(sys.argv) > 1:
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=7).view(np.int32)).to(dev)
    tr = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=5.0, ctx=ctx, seed=0xB5EED if sys.argv[1] == "noise" else None)
    batch = ba.SceneBatch(gt, cam.uniforms((w, h)))
    for _ in range(int(sys.argv[2])):
        tr.step(batch, spl)
    torch.cuda.synchronize()
    loop("trained", ba.RasterPass.Forward, False)
    loop("trained", ba.RasterPass.Forward, True)
    loop("trained", ba.RasterPass.Backward, True)
