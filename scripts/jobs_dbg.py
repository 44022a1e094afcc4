import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import brush_amd as ba
from brush_amd import synth
dev = torch.device("cuda:0")
n, w, h = 20000, 256, 256
scene = synth.make_scene(n, 0x5C, sh_degree=0, log_scale_range=(math.log(0.02), math.log(0.2)))
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
v_out = torch.rand((h, w, 4), device=dev) * 1e-3
res = {}
for name, opts in (("off", {"bwd_jobs": 0}), ("on", {"bwd_jobs": 1}), ("on2", {"bwd_jobs": 1})):
    ctx = ba.Context(dev, options=opts)
    spl = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
    for rep in range(2):
        r = ba.render_splats_bwd(spl, cam, (w, h), (0.1, 0.2, 0.3), v_out, ctx=ctx)
    to = r["aux"].tile_offsets.to(torch.int64)
    work = (to[:, 1] - to[:, 0]).clamp(min=0)
    res[name] = {k: r[k].cpu().numpy() for k in ("v_transforms", "v_sh_coeffs", "v_raw_opacities", "v_refine_weight")}
    print(name, "work max", int(work.max()), "mean", float(work.float().mean()), "pairs", r["aux"].num_intersections)
    ctx.close()
for a, b in (("off", "on"), ("on", "on2")):
    for k in res[a]:
        d = np.abs(res[a][k] - res[b][k]); m = np.abs(res[a][k]).max()
        print(a, b, k, "max rel diff %.3e" % (d.max() / max(m, 1e-30)), "frac > 1e-5*max: %.4f" % float((d > 1e-5 * m).mean()))

print("---- train steps")
gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=11).view(np.int32)).to(dev)
outs = {}
for name, opts in (("off", {"bwd_jobs": 0, "cut_min_pairs": 0}), ("on", {"bwd_jobs": 1, "cut_min_pairs": 0}), ("on_nocut", {"bwd_jobs": 1}), ("off_nocut", {"bwd_jobs": 0}), ("on_zero", {"bwd_jobs": 1, "cut_min_pairs": 0, "zero_grads": 1}), ("off_zero", {"bwd_jobs": 0, "cut_min_pairs": 0, "zero_grads": 1})):
    ctx = ba.Context(dev, options=opts)
    spl = ba.Splats(scene["transforms"].copy(), scene["sh"].copy(), scene["raw_opac"].copy(), device=dev)
    tr = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0, ctx=ctx, seed=0xB5EED)
    batch = ba.SceneBatch(gt, cam.uniforms((w, h)))
    snaps = []
    for s_ in range(3):
        tr.step(batch, spl)
        ctx.sync()
        snaps.append(torch.cat([spl.transforms.reshape(-1), spl.sh_coeffs.reshape(-1), spl.raw_opacities.reshape(-1)]).cpu().numpy())
        print(name, "step", s_ + 1, "share %.3f" % float(ctx.lib.bh_last_list_share(ctx._h)), "far", int(ctx.lib.bh_far_slices_queued(ctx._h)), "loss %.6f" % tr.stats(ctx).loss)
    outs[name] = snaps
    ctx.close()
for a, b in (("off", "on"), ("off_nocut", "on_nocut"), ("off_zero", "on_zero"), ("off", "off_nocut")):
    for s_ in range(3):
        d = np.abs(outs[a][s_] - outs[b][s_])
        print(a, b, "step", s_ + 1, "frac beyond 1e-6: %.4f max %.4g" % (float((d > 1e-6).mean()), d.max()))
