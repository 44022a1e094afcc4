#!/usr/bin/env python
"""End-to-end check of the whole training loop on synthetic multi-view data (BASELINE.json configs[3]
stand-in: the NeRF-synthetic lego images are not in this environment).

A ground-truth splat scene is rendered from V cameras on an orbit with this library's own rasterizer
(packed rgba8, like a decoded dataset image).  Training then starts from a perturbed, under-sized copy of
the scene and runs the full brush-train loop: SceneLoader (overlapped upload) -> SplatTrainer.step
(forward, L1+SSIM, backward, Adam, noise) -> SplatTrainer.refine every --refine-every steps
(prune / split / opacity decay / 3D-filter floor), optionally data-parallel over cameras with
torch.distributed (one process per GPU, RCCL).  Reports PSNR on held-out views before / after.

    python scripts/train_synthetic.py --steps 600
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_synthetic.py --steps 600
"""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def look_at(eye, target=(0.0, 0.0, 0.0)):
    """Camera-to-world rotation as a glam quaternion (x,y,z,w): +Z forward, +Y down, +X right."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    up = np.array([0.0, -1.0, 0.0])
    r = np.cross(up, f)   # x = down x forward ... chosen so that (r, d, f) is right-handed with +Y down
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    m = np.stack([r, d, f], axis=1)  # columns = camera axes in world space
    t = np.trace(m)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = ((m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s)
    else:
        i = int(np.argmax(np.diag(m)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
        v = [0.0, 0.0, 0.0]
        v[i] = 0.25 * s
        v[j] = (m[j, i] + m[i, j]) / s
        v[k] = (m[k, i] + m[i, k]) / s
        q = (v[0], v[1], v[2], (m[k, j] - m[j, k]) / s)
    return tuple(float(x) for x in q)


def make_gt_scene(n, seed, sh_degree):
    rng = np.random.default_rng(seed)
    # a few blobs of splats inside the unit ball
    centres = rng.uniform(-0.6, 0.6, (6, 3))
    means = (centres[rng.integers(0, 6, n)] + rng.normal(scale=0.18, size=(n, 3))).astype(np.float32)
    quats = rng.normal(size=(n, 4)).astype(np.float32)
    ls = rng.uniform(math.log(0.02), math.log(0.08), (n, 3)).astype(np.float32)
    c = (sh_degree + 1) ** 2
    sh = np.zeros((n, c, 3), np.float32)
    sh[:, 0, :] = rng.uniform(-1.2, 1.6, (n, 3))
    if c > 1:
        sh[:, 1:, :] = rng.uniform(-0.15, 0.15, (n, c - 1, 3))
    op = rng.uniform(0.5, 3.0, n).astype(np.float32)
    return dict(transforms=np.concatenate([means, quats, ls], 1).astype(np.float32), sh=sh, raw_opac=op)


def orbit_cameras(ba, count, radius, fov, phase=0.0):
    cams = []
    for k in range(count):
        a = phase + 2 * math.pi * k / count
        eye = (radius * math.cos(a), 0.6 * math.sin(2 * a + 0.3), radius * math.sin(a))
        cams.append(ba.Camera(position=eye, rotation=look_at(eye), fov_x=fov, fov_y=fov))
    return cams


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 99.0 if mse <= 0 else -10.0 * math.log10(mse)


def run(args, log=print):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
        pg = dist.group.WORLD
    import brush_amd as ba
    w = h = args.res
    fov = 0.7
    gt = make_gt_scene(args.gt_splats, 1, args.sh_degree)
    gt_spl = ba.Splats(gt["transforms"], gt["sh"], gt["raw_opac"], device=dev)
    train_cams = orbit_cameras(ba, args.views, 3.0, fov)
    eval_cams = orbit_cameras(ba, 4, 3.0, fov, phase=0.37)
    bg = (0.0, 0.0, 0.0)

    def render_u8(spl, cam):
        img, _ = ba.render_splats(spl, cam, (w, h), bg, ba.RasterPass.Backward)
        return img[..., :3].clamp(0, 1)
    views = []
    for cam in train_cams:  # "decoded dataset images": RGB8 host arrays
        rgb = (render_u8(gt_spl, cam) * 255.0 + 0.5).to(torch.uint8).cpu().numpy()
        views.append((np.ascontiguousarray(rgb), cam))
    eval_ref = [render_u8(gt_spl, cam) for cam in eval_cams]

    # initial model: a random subset of the GT splats, jittered, shrunk, grey, half transparent
    rng = np.random.default_rng(2)
    keep = rng.choice(args.gt_splats, args.init_splats, replace=False)
    tr0 = gt["transforms"][keep].copy()
    tr0[:, :3] += rng.normal(scale=0.03, size=(len(keep), 3)).astype(np.float32)
    tr0[:, 3:7] = rng.normal(size=(len(keep), 4)).astype(np.float32)
    tr0[:, 7:] = math.log(0.03)
    sh0 = np.zeros((len(keep), (args.sh_degree + 1) ** 2, 3), np.float32)
    spl = ba.Splats(tr0, sh0, np.zeros(len(keep), np.float32), device=dev)

    def eval_psnr(s):
        return float(np.mean([psnr(render_u8(s, c), r) for c, r in zip(eval_cams, eval_ref)]))
    p0 = eval_psnr(spl)
    cfg = ba.TrainConfig(total_train_iters=args.steps, refine_every=args.refine_every, growth_stop_iter=int(args.steps * 0.8),
                         max_splats=args.gt_splats * 2, exact_lists=args.exact_lists)
    trainer = ba.SplatTrainer(cfg, process_group=pg)
    trainer.set_bounds(*ba.splat_bounds(spl))
    focal = ba.fov_to_focal(fov, w)
    if args.filter3d:
        trainer.set_view_cams([(c.position, focal) for c in train_cams])
        spl.with_min_scale(trainer.compute_min_scale(spl))
    loader = ba.SceneLoader(views, seed=5, slots=3, rank=rank, world=world)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)  # identical noise on every rank keeps the replicas identical
    t0 = time.perf_counter()
    stats_log = []
    ctx = ba.get_context(dev)
    shares, far0 = [], int(ctx.lib.bh_far_slices_queued(ctx._h))
    for it in range(1, args.steps + 1):
        batch = loader.next_batch()
        if args.no_view_ids:
            batch.view_id = 0
        noise = torch.randn(spl.num_splats(), 3, device=dev, generator=gen)
        trainer.step(batch, spl, noise_samples=noise)
        shares.append(float(ctx.lib.bh_last_list_share(ctx._h)))
        if it % args.refine_every == 0 and it < args.steps:
            spl, rs = trainer.refine(it, spl)
            stats_log.append((it, rs.total_splats, rs.num_pruned, rs.num_added))
            if rank == 0:
                log("iter %d refine: %d splats (+%d split/added, -%d pruned)" % (it, rs.total_splats, rs.num_added, rs.num_pruned))
    loss = trainer.stats().loss
    dt = time.perf_counter() - t0
    loader.close()
    p1 = eval_psnr(spl)
    if world > 1:
        import torch.distributed as dist
        chk = torch.stack([spl.transforms.double().sum(), spl.raw_opacities.double().sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "replicas diverged"
        dist.destroy_process_group()
    far = int(ctx.lib.bh_far_slices_queued(ctx._h)) - far0
    if rank == 0:
        log("PSNR on held-out views: %.2f dB -> %.2f dB; final loss %.5f; %d splats; %.1f steps/s x %d ranks" %
            (p0, p1, loss, spl.num_splats(), args.steps / dt, world))
        # the forward's per-tile depth cuts over a whole training run (views cycling, refine every N steps): how much of the pair
        # lists was built, how often a view's forecast missed (a far pass) and how many frames ran with complete lists
        log("lists: mean share %.3f of the pairs, %d frames with complete lists, %d far passes in %d steps" %
            (float(np.mean(shares)), sum(1 for x in shares if x >= 1.0), far, args.steps))
    return dict(psnr_before=p0, psnr_after=p1, loss=loss, splats=spl.num_splats(), refines=stats_log, steps_per_s=args.steps / dt,
                mean_list_share=float(np.mean(shares)), far_passes=far)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--res", type=int, default=160)
    ap.add_argument("--views", type=int, default=24)
    ap.add_argument("--gt-splats", type=int, default=3000)
    ap.add_argument("--init-splats", type=int, default=1500)
    ap.add_argument("--sh-degree", type=int, default=1)
    ap.add_argument("--refine-every", type=int, default=100)
    ap.add_argument("--filter3d", action="store_true")
    ap.add_argument("--exact-lists", action="store_true", help="BhTrainConfig.exact_lists: the reference's full per-tile lists (A/B)")
    ap.add_argument("--no-view-ids", action="store_true", help="do not tell the library which view a batch is (A/B)")
    return ap.parse_args(argv)


if __name__ == "__main__":
    run(parse())
