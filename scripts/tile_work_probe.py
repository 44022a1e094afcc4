#!/usr/bin/env python
"""Diagnostic for the forward blend (K16, one wave per tile): how the blended splats are spread over the tiles of the headline
scene — the mean sets the kernel's total work, the longest tiles set the time of its last waves.
    python scripts/tile_work_probe.py [--steps 12] [--exact]
Prints the distribution of (shrunk list end - begin) per tile of the last training frame and two lower bounds of K16's time:
all SIMDs evenly loaded (sum / 1024 SIMDs) and the longest single tile running alone."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--exact", action="store_true")
    args = ap.parse_args()
    import brush_amd as ba
    from brush_amd import synth, _ffi
    from brush_amd.host import _view
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = ba.get_context(dev)
    scene, w, h = synth.config_scene("1m_1080p", 0)
    cp = synth.default_camera_params(w, h)
    cams = [ba.Camera(position=(cp["pos"][0] + 2.0 * v, cp["pos"][1], cp["pos"][2]), rotation=(0.0, 0.0, 0.0, 1.0), fov_x=cp["fov_x"], fov_y=cp["fov_y"],
                      center_uv=cp["center_uv"]) for v in range(2)]
    batches = [ba.SceneBatch(torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=7 + 100 * v).view(np.int32)).to(dev), c.uniforms((w, h)), view_id=v + 1)
               for v, c in enumerate(cams)]
    splats = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
    trainer = ba.SplatTrainer(ba.TrainConfig(exact_lists=args.exact), median_scene_scale=5.0, ctx=ctx, seed=0xB5EED)
    for s in range(args.steps):
        trainer.step(batches[s % 2], splats)
    ctx.sync()
    out = _ffi.BhRenderOut()
    ctx.check(ctx.lib.bh_last_render_out(ctx._h, C.byref(out)))
    t = out.num_tiles
    to = _view(out.tile_offsets, (t, 2), torch.int32, dev).to(torch.int64).cpu().numpy()
    work = np.clip(to[:, 1] - to[:, 0], 0, None)
    q = np.percentile(work, [50, 90, 99, 99.9])
    total = int(work.sum())
    print("tiles %d  blended pairs %d  mean %.1f  p50 %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  max %d" % (t, total, work.mean(), q[0], q[1], q[2], q[3], work.max()))
    srt = np.sort(work)[::-1]
    print("longest tiles:", srt[:16].tolist())
    for k in (64, 128, 256, 512, 1024):
        print("  tiles with more than %4d blended splats: %5d  (%.1f %% of the pairs)" % (k, int((work > k).sum()), 100.0 * work[work > k].sum() / max(1, total)))
    # the 8 tiles of a SIMD when the blocks are dealt in descending order of work over 1024 SIMDs
    simd = np.zeros(1024)
    for i, wk in enumerate(srt):
        simd[i % 1024] += wk
    print("per-SIMD pairs when dealt in descending order: mean %.0f  max %.0f  (max / mean %.3f)" % (simd.mean(), simd.max(), simd.max() / simd.mean()))


if __name__ == "__main__":
    main()
