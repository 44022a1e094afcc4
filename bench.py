#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path on MI355X.

A "step" is one SplatTrainer.step (forward render -> L1+SSIM loss -> backward ->
statistics -> Adam) on ONE 1920x1080 view of the 1 M-splat synthetic scene
(BASELINE.json configs[2]), inputs already resident in HBM.
With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank
renders a different view per step and the gradients are all-reduced over RCCL
between backward and Adam (data parallel over cameras, weak scaling).

Prints ONE JSON line on rank 0 (contract: see the task statement), including
  roofline     — dominant kernel (rasterize_backward_kernel), algorithmic bytes per
                 launch / its average duration measured with HIP events on the ctx stream
  cpu_baseline — the CPU oracle (a C++ restatement of Brush's kernels; Brush has no CPU
                 backend) timed on the host cores on one full step of the same workload
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured achievable


def stage_bytes(n, nv, ni, pixels, tiles, coeffs):
    """Algorithmic HBM bytes per stage (SURVEY.md §8d / DESIGN.md §5)."""
    c = coeffs
    return {
        # K1 also stores the projected record by splat id and clears visible + the train step's gradient span on its way
        "ProjectSplats": 44 * n + 12 * n + 36 * nv + 4 * n + (48 + 12 * c) * n,
        "DepthSort": 80 * n,
        "PrefixSumGaussHits": 12 * nv,
        "ProjectVisible": (84 + 12 * c) * nv,   # separate launch only for frames without intersections
        # K5 also gathers the records into depth order (the former K4) and clears the backward's v_combined
        "MapGaussiansToIntersect": 32 * nv + 8 * ni + 76 * nv + 40 * nv,
        "TileSort": 40 * ni,
        "GetTileOffsets": 4 * ni + 8 * tiles,
        "Rasterize": 44 * ni + 16 * pixels,
        "ImageLoss": (16 + 4 + 12) * pixels,
        "ImageLossBackward": (16 + 4 + 16) * pixels,
        "ZeroGradBuffers": (48 + 12 * c) * n + 40 * nv,
        "RasterizeBackwards": 80 * ni + 32 * pixels,
        "ProjectBackwards": (88 + 12 * c) * nv + (48 + 12 * c) * nv,
        "OptimizerStep": 28 * 11 * n + (20 * 3 * c + 8) * n + 36 * n,  # Adam x3 + refine statistics, one launch
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 0.12 s of GPU time; 20 steps read 3-5 % slower (clock ramp)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sh-degree", type=int, default=0)
    ap.add_argument("--workload", default="1m_1080p")
    ap.add_argument("--splats", type=int, default=0, help="override the splat count (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm", choices=["torch", "native"], default="torch",
                    help="N>1 gradient exchange: 'torch' = torch.distributed all_reduce (backend nccl = RCCL) from the exchange hook; "
                         "'native' = the library's own RCCL communicator (bh_comm_init / built-in exchange in bh_train_step); the "
                         "unique id travels through a torch TCPStore on MASTER_ADDR:MASTER_PORT+1")
    ap.add_argument("--exchange", choices=["sparse", "dense"], default="sparse",
                    help="N>1 (cameras and tiles): 'sparse' = mask-keyed exchange (visible flags, then only the gradient rows of splats some rank "
                         "saw; dense fallback above half of the scene), 'dense' = one all-reduce of the whole exchange buffer")
    ap.add_argument("--feed", choices=["resident", "loader"], default="resident",
                    help="'resident' (the headline): the GT batch is already in HBM when the timed region starts; 'loader': every step "
                         "takes a fresh 1080p RGB8 host image through SceneLoader/BatchUploader (pinned ring + copy stream + device "
                         "packing) - the PCIe-inclusive rate quoted in DESIGN.md, never `value` of the headline line")
    ap.add_argument("--parallel", choices=["cameras", "tiles"], default="cameras",
                    help="N>1: 'cameras' = data parallel, one view per rank (weak scaling, the headline); "
                         "'tiles' = ONE view partitioned by strips of tile rows (strong scaling, BASELINE.json configs[4])")
    args = ap.parse_args()

    # the contract is ONE JSON line on stdout: keep the real stdout aside and point fd 1 at stderr, so that banners
    # printed by native libraries (RCCL prints its version to stdout when a communicator is created) cannot join it
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    # BH_FORCE_PG=1: build the RCCL process group even for one rank (developer smoke test of the exchange-hook path on a 1-GPU box)
    if world > 1 or os.environ.get("BH_FORCE_PG") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm
        pg = dist.group.WORLD

    import brush_amd as ba
    from brush_amd import synth

    scene, w, h = synth.config_scene(args.workload, args.sh_degree, n=args.splats or None)
    n = scene["transforms"].shape[0]
    coeffs = scene["sh"].shape[1]
    cp = synth.default_camera_params(w, h)
    # one view per rank: rank r looks at the scene with a small extra yaw so the ranks'
    # gradients differ (data parallel over cameras); rank 0 is the named config's camera
    tile_mode = args.parallel == "tiles" and world > 1
    yaw = 0.0 if tile_mode else 0.02 * rank
    rot = (0.0, math.sin(yaw / 2.0), 0.0, math.cos(yaw / 2.0))
    cam = ba.Camera(position=cp["pos"], rotation=rot, fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
    splats = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=7 + (0 if tile_mode else rank)).view(np.int32)).to(dev)
    batch = ba.SceneBatch(gt, cam.uniforms((w, h)))
    ctx = ba.get_context(dev)
    native = args.comm == "native" and not tile_mode and (world > 1 or os.environ.get("BH_FORCE_PG") == "1")
    if native:
        from torch.distributed import TCPStore
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29517")) + 1, world, rank == 0)
        if rank == 0:
            store.set("bh_comm_id", ba.Context.comm_unique_id())
        ctx.comm_init(rank, world, store.get("bh_comm_id"))
    trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=5.0, process_group=None if native else pg, ctx=ctx, partition=args.parallel,
                              native_comm=native, sparse_exchange=args.exchange == "sparse")

    loader = None
    if args.feed == "loader":
        rng = np.random.default_rng(1 + rank)
        host_views = [(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), cam.uniforms((w, h))) for _ in range(6)]
        loader = ba.SceneLoader(host_views, seed=rank, slots=3, ctx=ctx)

    def next_batch():
        return loader.next_batch() if loader is not None else batch

    def barrier():
        if pg is not None:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        trainer.step(next_batch(), splats)
    barrier()
    # timed region: HIP events only around the dominant kernel (2 per step); bracketing all ~15 stages
    # costs ~0.1 ms of host time per step, so the per-stage table comes from a separate untimed pass
    ctx.profile(2)
    ctx.profile_fetch()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(next_batch(), splats)
    barrier()
    dt = time.perf_counter() - t0
    dominant = ctx.profile_fetch()
    ctx.profile(1)
    for _ in range(min(args.steps, 10)):
        trainer.step(batch, splats)
    barrier()
    stages = ctx.profile_fetch()
    stages.update(dominant)   # the dominant kernel's duration is the one measured inside the timed region
    ctx.profile(0)
    st = trainer.stats()
    # how much of the per-tile lists the blend kernels actually consume before every pixel saturates (outside the
    # timed region): the forward shrinks each tile's list end to its last useful splat
    _, aux = ba.render_splats(splats, cam, (w, h), (0.0, 0.0, 0.0), ba.RasterPass.Backward, ctx=ctx)
    to = aux.tile_offsets.to(torch.int64)
    isect_blended = int((to[:, 1] - to[:, 0]).clamp(min=0).sum().item())

    if pg is not None:
        import torch.distributed as dist
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        nv, ni = st.num_visible, st.num_intersections
        pixels, tiles = w * h, ((w + 15) // 16) * ((h + 15) // 16)
        sb = stage_bytes(n, nv, ni, pixels, tiles, coeffs)
        stage_out = {}
        for name, (ms, calls) in stages.items():
            avg = ms / max(calls, 1)
            e = {"ms": round(avg, 4)}
            if name in sb and avg > 0:
                e["algo_MB"] = round(sb[name] / 1e6, 2)
                e["GBps"] = round(sb[name] / 1e9 / (avg * 1e-3), 1)
            stage_out[name] = e
        fwd_names = ["ProjectSplats", "DepthSort", "PrefixSumGaussHits", "ProjectVisible", "MapGaussiansToIntersect", "TileSort", "GetTileOffsets", "Rasterize"]
        bwd_names = ["ZeroGradBuffers", "RasterizeBackwards", "ProjectBackwards"]
        fwd_ms = sum(stage_out[k]["ms"] for k in fwd_names if k in stage_out)
        bwd_ms = sum(stage_out[k]["ms"] for k in bwd_names if k in stage_out)
        dom = "RasterizeBackwards"
        dom_ms = stage_out.get(dom, {}).get("ms", 0.0)
        dom_bytes = sb[dom]
        achieved = dom_bytes / 1e9 / (dom_ms * 1e-3) if dom_ms > 0 else 0.0
        # the committed PMC passes were taken on the headline workload: no counter figure for any other
        headline = args.workload == "1m_1080p" and args.sh_degree == 0 and not args.splats and not tile_mode
        traffic, traffic_src = pmc_traffic_bytes("rasterize_backward_kernel") if headline else (None, None)
        out = {
            "metric": "train views/sec @ 1M Gaussians, 1080p (fwd + L1/SSIM loss + bwd + Adam per view)",
            "value": round((1 if tile_mode else world) * args.steps / dt, 3),
            "unit": "views/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if tile_mode else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if loader is None else "synthetic, a fresh host RGB8 image uploaded per step (PCIe-inclusive; not the headline)",
            "config": {"workload": "%s: %d splats, %dx%d, SH degree %d, one view per rank per step (BASELINE.json configs[2])" % (args.workload, n, w, h, args.sh_degree),
                       "num_visible": nv, "num_intersections": ni, "parallelism": ("tiles%d: one view split by strips of tile rows (strip-wise loss with 21-px halo exchange + mask-keyed all-reduce of gradients)" % world if tile_mode else
                                       "dp%d over cameras (RCCL all-reduce of gradients%s)" % (world, ", library-owned communicator" if native else "")) if world > 1 else "single GPU"},
            "exchange": ({"mode": args.exchange, "rows_last_step": st.exchange_rows, "rows_total": n} if pg is not None or native else None),
            "fwd_ms": round(fwd_ms, 4),
            "fwd_bwd_ms": round(fwd_ms + bwd_ms, 4),
            "kernel_ms_per_step": round(sum(e["ms"] for e in stage_out.values()), 4),
            "roofline": {"bound": "hbm", "kernel": "rasterize_backward_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                         "intersections_blended": isect_blended,
                         "note": "achieved = SURVEY 8d algorithmic bytes (80*I + 32*P, every listed intersection) / measured duration; the kernel "
                                 "stops each tile when all its pixels saturate and blends only %d of the %d listed intersections (%.1f %%), so it is "
                                 "VALU-issue bound, not HBM bound (DESIGN.md §5): %.2f ns per blended (splat, tile), %.1f G pixel-splat evaluations/s"
                                 % (isect_blended, ni, 100.0 * isect_blended / max(ni, 1), (dom_ms * 1e6 / max(isect_blended, 1)),
                                    256.0 * isect_blended / 1e9 / (dom_ms * 1e-3) if dom_ms > 0 else 0.0)},
            "stages": stage_out,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, cp, w, h)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if loader is not None:
        loader.close()
    if pg is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of `kernel_substr` from the newest profiles/*_hbm_traffic.csv
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction applied)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.csv")))
    if not files:
        return None, None
    for r in csv.DictReader(open(files[-1])):
        if kernel_substr in r["kernel"]:
            return float(r["HBM_MB_per_launch_corrected"]) * 1e6, os.path.basename(files[-1])
    return None, None


def cpu_baseline(scene, cp, w, h):
    """One full step of the SAME workload on the CPU oracle (OpenMP over splats/tiles)."""
    import brush_amd as ba
    from brush_amd import synth
    from oracle import bo
    from oracle.trainer import OracleTrainer
    sc = {k: v.copy() for k, v in scene.items()}
    otr = OracleTrainer(bo, ba.TrainConfig(), median_scene_scale=5.0)
    gt = synth.synthetic_gt_packed(w, h, seed=7)
    steps = 3
    t = time.perf_counter()
    for _ in range(steps):
        otr.step(sc, bo.camera(**cp), gt, (0.0, 0.0, 0.0))
    dt = time.perf_counter() - t
    return {"value": round(steps / dt, 5), "unit": "views/s", "cores": bo.num_threads(), "kind": "port",
            "sample": "%d full train steps (1 view each) of the same workload; %.2f s wall" % (steps, dt),
            "what": "C++/OpenMP restatement of Brush's CubeCL kernels (oracle/brush_oracle.cpp); Brush itself has no CPU backend"}


if __name__ == "__main__":
    main()
