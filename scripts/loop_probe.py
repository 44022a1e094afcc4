#!/usr/bin/env python
"""Diagnostic for the per-tile-cut forecast in a realistic loop (bench.py `train_loop`): per step, which view was trained, the near
share, whether the frame needed a second attempt, and the steps since the last refine.  Prints a summary per configuration.
    python scripts/loop_probe.py --views 64 --steps 600 [--no-noise] [--refine-every 200]"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--refine-every", type=int, default=200)
    ap.add_argument("--no-noise", action="store_true")
    ap.add_argument("--exact", action="store_true")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    import brush_amd as ba
    from brush_amd import synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = ba.get_context(dev)
    scene, w, h = synth.config_scene("1m_1080p", 0)
    cp = synth.default_camera_params(w, h)
    cams = []
    for v in range(args.views):
        ang = 2.0 * math.pi * v / args.views
        pos = (cp["pos"][0] + math.cos(ang) - 1.0, cp["pos"][1] + 0.5 * math.sin(ang), cp["pos"][2])
        yaw = -math.atan2(pos[0] - cp["pos"][0], 7.0)
        cams.append(ba.Camera(position=pos, rotation=(0.0, math.sin(yaw / 2.0), 0.0, math.cos(yaw / 2.0)), fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"]))
    gts = [torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=7 + 100 * k).view(np.int32)).to(dev) for k in range(8)]
    batches = [ba.SceneBatch(gts[v % 8], c.uniforms((w, h)), view_id=v + 1) for v, c in enumerate(cams)]
    splats = ba.Splats(scene["transforms"].copy(), scene["sh"].copy(), scene["raw_opac"].copy(), device=dev)
    trainer = ba.SplatTrainer(ba.TrainConfig(exact_lists=args.exact, refine_every=args.refine_every), median_scene_scale=5.0, ctx=ctx, seed=None if args.no_noise else 0xB5EED)
    trainer.set_bounds(*ba.splat_bounds(splats, ctx=ctx))
    rng = np.random.default_rng(3)
    order = []
    while len(order) < args.steps + 4:
        order.extend(rng.permutation(args.views).tolist())
    for k in range(4):
        trainer.step(batches[order[k]], splats)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.bh_forget_views(ctx._h))
    log = []
    last_refine = 0
    last_seen = {}
    q = int(ctx.lib.bh_far_slices_queued(ctx._h))
    t0 = time.perf_counter()
    for it in range(1, args.steps + 1):
        v = order[3 + it]
        trainer.step(batches[v], splats)
        q2 = int(ctx.lib.bh_far_slices_queued(ctx._h))
        log.append((it, v, float(ctx.lib.bh_last_list_share(ctx._h)), q2 - q, it - last_refine, it - last_seen.get(v, it), last_seen.get(v, 0) <= last_refine))
        q = q2
        last_seen[v] = it
        if it % args.refine_every == 0 and it < args.steps:
            splats, _ = trainer.refine(it, splats)
            last_refine = it
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    miss = [r for r in log if r[3]]
    first_after_refine = [r for r in log if r[6] and r[0] > args.views]
    miss_first = [r for r in first_after_refine if r[3]]
    cutf = [r for r in log if r[2] < 1.0]
    print("%s views=%d steps=%d refine_every=%d noise=%s exact=%s margin=%s: %.4f ms/step | misses %d (%d of them on a view's first frame after a refine, of %d such frames) | "
          "cut frames %d mean share %.3f | complete-list frames %d" % (
              args.tag, args.views, args.steps, args.refine_every, not args.no_noise, args.exact, os.environ.get("BH_CUT_MARGIN_PCT", "150"), dt / args.steps * 1e3, len(miss), len(miss_first),
              len(first_after_refine), len(cutf), (sum(r[2] for r in cutf) / max(1, len(cutf))), len(log) - len(cutf)))
    seg = max(200, args.steps // 6)
    for a in range(0, args.steps, seg):
        part = [r for r in log if a < r[0] <= a + seg]
        cut = [r for r in part if r[2] < 1.0]
        print("   steps %5d-%5d: second attempts %3d, frames with complete lists %3d, mean share of the cut frames %.3f" % (
            a + 1, a + len(part), sum(1 for r in part if r[3]), len(part) - len(cut), sum(r[2] for r in cut) / max(1, len(cut))))
    gaps = {}
    for r in log:
        if r[5] > 0:
            b = min(r[5] // 16, 8)
            g = gaps.setdefault(b, [0, 0])
            g[0] += 1
            g[1] += 1 if r[3] else 0
    print("   misses by steps-since-the-view's-last-visit (bucket of 16: frames, misses):", {k * 16: tuple(v) for k, v in sorted(gaps.items())})


if __name__ == "__main__":
    main()
