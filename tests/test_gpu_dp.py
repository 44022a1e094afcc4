"""Data parallel over cameras on the GPU path: two ranks (two processes sharing the one GPU of
the test box, gloo process group carrying the CUDA buffers) run SplatTrainer.step on different
views; the all-reduced update must equal the oracle's "sum of the K single-view gradients / K"
step and be identical on both ranks."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from brush_amd import synth
import util

pytestmark = pytest.mark.gpu
N, W, H = 3000, 128, 96


def _cam_params(rank):
    cp = synth.default_camera_params(W, H)
    cp["rot_xyzw"] = util.quat_from_axis_angle((0, 1, 0), 0.05 * rank)
    return cp


def _scene():
    # spread 1.8: two thirds of the splats lie outside the frustum, so the union of the ranks' contributing splats is a
    # real subset of the scene and the mask-keyed exchange takes its compact path (not the dense fallback)
    return synth.make_scene(N, 0xD0, sh_degree=1, log_scale_range=(math.log(0.03), math.log(0.25)),
                            tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * H / W), spread=1.8)


def _worker(rank, world, port, q, sparse, seed=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import brush_amd as ba
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    sc = _scene()
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    gt = torch.from_numpy(synth.synthetic_gt_packed(W, H, seed=3 + rank).view(np.int32)).to(dev)
    trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=2.0, process_group=dist.group.WORLD, sparse_exchange=sparse, seed=seed)
    batch = ba.SceneBatch(gt, util.hip_camera(ba, _cam_params(rank)))
    out, rows = [], []
    for _ in range(2):
        if seed is None:
            trainer.step(batch, spl, background=(0.1, 0.2, 0.3))
        else:
            trainer.step(batch, spl)   # background jitter + mean noise from the library's generator, keyed by (seed, step)
        rows.append(trainer.stats().exchange_rows)
        out.append((spl.transforms.cpu().numpy().copy(), spl.sh_coeffs.cpu().numpy().copy(), spl.raw_opacities.cpu().numpy().copy()))
    trainer.sync_refine_stats()  # the running maxima are rank-local until refine asks for them
    q.put((rank, out, trainer.state["vis_weight"].cpu().numpy(), trainer.state["refine_weight_norm"].cpu().numpy(),
           trainer.state["max_screen_size"].cpu().numpy(), rows))
    dist.destroy_process_group()


@pytest.mark.parametrize("sparse", [False, True])
def test_two_rank_step_matches_oracle_mean_gradient(oracle_lib, sparse):
    """sparse = the mask-keyed exchange (visible flags, then the compact block of the union's gradient rows): both modes
    must produce the oracle's mean-gradient update."""
    import brush_amd as ba
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, sparse)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, o0, vis0, norm0, scr0, rows0), (_, o1, vis1, norm1, scr1, rows1) = res
    assert rows0 == rows1
    if sparse:   # the union of two views' contributing splats: a real subset of the scene, compacted on both steps
        assert all(0 < r <= N // 2 for r in rows0), rows0
        assert rows0[0] == int((vis0 > 0).sum()) or rows0[1] == int((vis0 > 0).sum())   # rows = |union of the contributing splats|
    else:
        assert rows0 == [0, 0]
    for step in range(2):  # replicas stay bit-identical
        for a, b in zip(o0[step], o1[step]):
            assert np.array_equal(a, b)
    assert np.array_equal(vis0, vis1) and np.array_equal(norm0, norm1) and np.array_equal(scr0, scr1)
    assert vis0.max() == 4.0  # vis_weight counts views: 2 ranks x 2 steps
    # oracle: rank 0 steps with rank 1's raw gradients added, scaled by 1/2
    cfg = ba.TrainConfig()
    sc = _scene()
    bo = oracle_lib
    ot = util.OracleTrainer(bo, cfg, median_scene_scale=2.0)
    g1 = util.OracleTrainer(bo, cfg, 2.0).step({k: v.copy() for k, v in sc.items()}, bo.camera(**_cam_params(1)),
                                                 synth.synthetic_gt_packed(W, H, seed=4), (0.1, 0.2, 0.3), dry_run=True)
    ot.step(sc, bo.camera(**_cam_params(0)), synth.synthetic_gt_packed(W, H, seed=3), (0.1, 0.2, 0.3), extra_grads=[g1], world=2)
    tr, sh, op = o0[0]
    util.assert_adam_close(tr[:, 3:7], sc["transforms"][:, 3:7], cfg.lr_rotation, 1, "rotation")
    util.assert_adam_close(tr[:, 7:10], sc["transforms"][:, 7:10], cfg.lr_scale, 1, "scale")
    util.assert_adam_close(op, sc["raw_opac"], cfg.lr_opac, 1, "opacity")
    util.assert_adam_close(sh, sc["sh"], cfg.lr_coeffs_dc, 1, "sh")
    # and it is NOT the single-view update: the second view changed something
    single = util.OracleTrainer(bo, cfg, 2.0)
    sc1 = _scene()
    single.step(sc1, bo.camera(**_cam_params(0)), synth.synthetic_gt_packed(W, H, seed=3), (0.1, 0.2, 0.3))
    assert np.abs(sc1["raw_opac"] - sc["raw_opac"]).max() > 0.1 * cfg.lr_opac


def test_sparse_exchange_on_one_rank_is_the_identity(dev):
    """A 1-rank group sums nothing, so the mask-keyed path (flags -> union -> gather -> "sum" -> scatter) must leave the
    gradients exactly as it found them: same losses, same trajectory as the plain single-GPU trainer; and it falls back
    to the dense block when more than half of the scene is in the union."""
    import torch.distributed as dist
    import brush_amd as ba
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        sc = _scene()
        gt = torch.from_numpy(synth.synthetic_gt_packed(W, H, seed=3).view(np.int32)).to(dev)
        batch = ba.SceneBatch(gt, util.hip_camera(ba, _cam_params(0)))
        results = []
        for pg, sparse in ((dist.group.WORLD, True), (None, False)):
            spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
            tr = ba.SplatTrainer(ba.TrainConfig(mean_noise_weight=0.0), median_scene_scale=2.0, process_group=pg, sparse_exchange=sparse)
            losses, rows = [], []
            for _ in range(3):
                tr.step(batch, spl, background=(0.1, 0.2, 0.3))
                st = tr.stats()
                losses.append(st.loss)
                rows.append(st.exchange_rows)
            results.append((losses, rows, spl, tr.state["vis_weight"].clone()))
        (la, ra, sa, va), (lb, rb, sb, vb) = results
        assert all(0 < r <= N // 2 for r in ra) and rb == [0, 0, 0]
        assert np.allclose(la, lb, rtol=1e-6, atol=1e-7)
        assert torch.equal(va, vb)
        cfg = ba.TrainConfig()
        util.assert_adam_close(sa.transforms[:, 7:].cpu().numpy(), sb.transforms[:, 7:].cpu().numpy(), cfg.lr_scale, 3, "scale")
        util.assert_adam_close(sa.raw_opacities.cpu().numpy(), sb.raw_opacities.cpu().numpy(), cfg.lr_opac, 3, "opacity")
        # a scene where most splats are seen: the union exceeds N/2 -> dense fallback (exchange_rows = 0)
        few = synth.make_scene(400, 0xD1, sh_degree=0, log_scale_range=(math.log(0.05), math.log(0.1)), z_range=(6.0, 12.0),
                               tan_half_fov=(math.tan(math.radians(20)), math.tan(math.radians(20)) * H / W), spread=0.8)
        few["raw_opac"][:] = -3.0   # faint splats: nothing saturates, every visible splat contributes
        spl = ba.Splats(few["transforms"], few["sh"], few["raw_opac"], device=dev)
        tr = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=2.0, process_group=dist.group.WORLD, sparse_exchange=True)
        tr.step(batch, spl)
        st = tr.stats()
        assert st.exchange_rows == 0 and st.num_visible > 200
    finally:
        dist.destroy_process_group()


def test_two_ranks_with_device_noise_stay_identical():
    """The seeded step (noise drawn on the device, background jittered): both ranks pass the same seed, the gate uses the SUMMED
    visible flags and the identically updated opacity, so the replicas must stay bit-identical without exchanging a sample — and
    the noise must actually have been applied (the trajectory differs from the unseeded one)."""
    world = 2
    ctx = mp.get_context("spawn")
    runs = {}
    for seed in (0x5EED, None):
        q = ctx.Queue()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q, True, seed)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        (_, o0, vis0, *_), (_, o1, vis1, *_) = res
        for step in range(2):
            for a, b in zip(o0[step], o1[step]):
                assert np.array_equal(a, b), "replicas diverged (seed %r, step %d)" % (seed, step)
        runs[seed] = (o0[1][0], vis0)
    moved = np.any(runs[0x5EED][0][:, :3] != runs[None][0][:, :3], axis=1)
    assert moved.sum() > 20, "the seeded run shows no noise"
    assert not moved[runs[None][1] == 0].any(), "a splat no rank saw was moved"
