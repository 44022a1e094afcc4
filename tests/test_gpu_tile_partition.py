"""One frame partitioned over ranks by strips of tile rows (BASELINE.json configs[4]):
strip renders compose to the whole-image render exactly (same per-tile splat lists), partial
gradients sum to the whole-image gradients, and a 2-rank tile-partitioned train step equals the
single-GPU step."""
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


def _problem(n=6000, w=200, h=150, deg=1, seed=0xE1, spread=None):
    kw = {} if spread is None else {"spread": spread}   # spread 1.8: two thirds of the splats lie outside the frustum
    sc = synth.make_scene(n, seed, sh_degree=deg, log_scale_range=(math.log(0.02), math.log(0.25)),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w), **kw)
    return sc, synth.default_camera_params(w, h), w, h


@pytest.mark.parametrize("cuts", [(0, 4, 10), (0, 1, 2, 9, 10), (0, 10)])
def test_strips_compose_to_the_full_render_exactly(dev, cuts):
    import brush_amd as ba
    sc, cp, w, h = _problem()
    cam = util.hip_camera(ba, cp)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    bg = (0.2, 0.1, 0.3)
    full, aux = ba.render_splats(spl, cam, (w, h), bg, ba.RasterPass.Backward)
    tile_bh = (h + 15) // 16
    assert cuts[-1] == tile_bh
    composed = torch.full_like(full, float("nan"))
    ni = 0
    vis = torch.zeros_like(aux.visible)
    for b, e in zip(cuts[:-1], cuts[1:]):
        img, a = ba.render_splats(spl, cam, (w, h), bg, ba.RasterPass.Backward, tile_rows=(b, e))
        r0, r1 = b * 16, min(e * 16, h)
        composed[r0:r1] = img[r0:r1]
        ni += a.num_intersections
        vis = torch.maximum(vis, a.visible)
        # the strip's tile lists are the whole-image lists of those tiles
        to_f = util.u32(aux.tile_offsets).reshape(-1, 2).astype(np.int64)
        to_s = util.u32(a.tile_offsets).reshape(-1, 2).astype(np.int64)
        tbw = (w + 15) // 16
        gf = util.u32(aux.global_from_compact_gid)[util.u32(aux.compact_gid_from_isect)]
        gs = util.u32(a.global_from_compact_gid)[util.u32(a.compact_gid_from_isect)] if a.num_intersections else np.zeros(0, np.uint32)
        for t in range(b * tbw, e * tbw):
            assert np.array_equal(gf[to_f[t, 0]:to_f[t, 1]], gs[to_s[t, 0]:to_s[t, 1]]), "tile %d" % t
        assert not to_s[: b * tbw].any() and not to_s[e * tbw:].any()
    assert ni == aux.num_intersections            # every (splat, tile) pair belongs to exactly one strip
    assert torch.equal(composed, full)            # bit-exact image
    assert torch.equal(vis, aux.visible)


def test_strip_gradients_sum_to_the_full_gradients(dev):
    import brush_amd as ba
    sc, cp, w, h = _problem(deg=2)
    cam = util.hip_camera(ba, cp)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    bg = (0.0, 0.0, 0.0)
    rng = np.random.default_rng(5)
    v_out = torch.from_numpy(rng.normal(size=(h, w, 4)).astype(np.float32) / (h * w)).to(dev)
    full = ba.render_splats_bwd(spl, cam, (w, h), bg, v_out)
    acc = None
    for rows in ((0, 3), (3, 7), (7, 10)):
        part = ba.render_splats_bwd(spl, cam, (w, h), bg, v_out, tile_rows=rows)
        keys = ("v_transforms", "v_sh_coeffs", "v_raw_opacities")
        acc = {k: part[k].clone() for k in keys} if acc is None else {k: acc[k] + part[k] for k in keys}
    for k, v in acc.items():
        ref = full[k]
        assert float((v - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-12, k


def _worker(rank, world, port, q, partition, rebalance_every=8, steps=2, strip_loss=True, sparse=True, spread=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import brush_amd as ba
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    sc, cp, w, h = _problem(n=4000, w=160, h=112, spread=spread)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=3).view(np.int32)).to(dev)
    trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=2.0, process_group=dist.group.WORLD, partition=partition,
                              sparse_exchange=sparse)
    trainer.rebalance_every = rebalance_every
    trainer.strip_loss = strip_loss
    batch = ba.SceneBatch(gt, util.hip_camera(ba, cp))
    losses, shares = [], []
    for _ in range(steps):
        trainer.step(batch, spl, background=(0.1, 0.2, 0.3))
        st = trainer.stats()
        shares.append((st.loss, trainer._strip_loss_now, st.exchange_rows))
        losses.append(trainer.reduce_loss(st))     # strip-wise loss: the ranks' shares sum to the frame's loss
    trainer.sync_refine_stats()  # max_screen_size is strip-local until refine asks for it
    q.put((rank, spl.transforms.cpu().numpy(), spl.sh_coeffs.cpu().numpy(), spl.raw_opacities.cpu().numpy(), losses,
           trainer.state["vis_weight"].cpu().numpy(), trainer.state["max_screen_size"].cpu().numpy(),
           trainer.state["refine_weight_norm"].cpu().numpy(), trainer._row_weights, shares))
    dist.destroy_process_group()


@pytest.mark.parametrize("strip_loss,sparse", [(True, True), (False, False), (True, False)])
def test_two_rank_tile_partitioned_step_equals_single_gpu_step(dev, strip_loss, sparse):
    """strip_loss: each rank evaluates L1 + SSIM on its own strip after fetching 21-px halos from its neighbour (no
    whole-image all-gather); otherwise every rank gathers the frame and evaluates the whole loss.  sparse: the
    mask-keyed exchange (visible flags, then the compact rows — gradients AND refine weight — of the splats some strip
    saw) instead of one dense sum of the whole buffer.  All of them must follow the single-GPU trajectory."""
    import brush_amd as ba
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "tiles", 8, 2, strip_loss, sparse, 1.8)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0[8] is None                     # 2 steps < rebalance_every: equal-height strips throughout
    if strip_loss:   # both strips (64 and 48 px) are taller than the halo: the strip-wise path ran, each rank holds a share
        assert all(x[1] for x in r0[9]) and all(x[1] for x in r1[9])
        assert all(0.0 < a[0] and 0.0 < b[0] and abs(a[0] - b[0]) > 1e-6 for a, b in zip(r0[9], r1[9]))
    else:
        assert not any(x[1] for x in r0[9]) and [a[0] for a in r0[9]] == [b[0] for b in r1[9]]
    rows = [x[2] for x in r0[9]]
    assert rows == [x[2] for x in r1[9]]
    if sparse:   # the union of the two strips' contributing splats: a real subset of the scene, sent as compact rows
        assert all(0 < r <= 2000 for r in rows), rows
    else:
        assert rows == [0, 0]
    for a, b in zip(r0[1:8], r1[1:8]):  # replicas identical
        assert np.array_equal(np.asarray(a), np.asarray(b))
    # single-GPU reference on this process
    sc, cp, w, h = _problem(n=4000, w=160, h=112, spread=1.8)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=3).view(np.int32)).to(dev)
    cfg = ba.TrainConfig()
    trainer = ba.SplatTrainer(cfg, median_scene_scale=2.0)
    batch = ba.SceneBatch(gt, util.hip_camera(ba, cp))
    losses = []
    for _ in range(2):
        trainer.step(batch, spl, background=(0.1, 0.2, 0.3))
        losses.append(trainer.stats().loss)
    assert np.allclose(r0[4], losses, rtol=1e-6, atol=1e-7)      # same image -> same loss
    tr = spl.transforms.cpu().numpy()
    util.assert_adam_close(r0[1][:, 3:7], tr[:, 3:7], cfg.lr_rotation, 2, "rotation")
    util.assert_adam_close(r0[1][:, 7:10], tr[:, 7:10], cfg.lr_scale, 2, "scale")
    util.assert_adam_close(r0[3], spl.raw_opacities.cpu().numpy(), cfg.lr_opac, 2, "opacity")
    util.assert_adam_close(r0[2], spl.sh_coeffs.cpu().numpy(), cfg.lr_coeffs_dc, 2, "sh")
    assert np.mean(r0[5] != trainer.state["vis_weight"].cpu().numpy()) <= 2e-3
    # step 2 renders parameters that already carry step 1's Adam rounding differences: close, not identical
    assert np.allclose(r0[6], trainer.state["max_screen_size"].cpu().numpy(), rtol=5e-3, atol=1e-6)
    # the refine weight is a per-pixel sum: the strips' partial sums add up to the single-GPU value
    ref_norm = trainer.state["refine_weight_norm"].cpu().numpy()
    assert np.abs(r0[7] - ref_norm).max() <= 1e-3 * ref_norm.max() + 1e-12


def test_strips_rebalanced_by_blended_intersections(dev):
    """strips are cut by intersection count, not rows.  With rebalance_every = 1 the second and third
    steps run on strips re-cut from the previous frame's blended-intersection counts per tile row; the result still
    equals the single-GPU trajectory (strip composition is exact for ANY cut) and both ranks agree on the weights."""
    import brush_amd as ba
    from brush_amd.parallel import tile_rows_for_rank
    world, steps = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "tiles", 1, steps)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0[8] is not None and r0[8] == r1[8] and len(r0[8]) == 7          # 112 px = 7 tile rows, same weights on both ranks
    cuts = [tile_rows_for_rank(7, r, world, r0[8]) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[0][1] == cuts[1][0] and cuts[1][1] == 7
    loads = [sum(r0[8][b:e]) for b, e in cuts]
    assert max(loads) <= 0.75 * sum(loads)                                     # neither rank carries most of the work
    for a, b in zip(r0[1:8], r1[1:8]):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    sc, cp, w, h = _problem(n=4000, w=160, h=112)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=3).view(np.int32)).to(dev)
    cfg = ba.TrainConfig()
    trainer = ba.SplatTrainer(cfg, median_scene_scale=2.0)
    batch = ba.SceneBatch(gt, util.hip_camera(ba, cp))
    losses = []
    for _ in range(steps):
        trainer.step(batch, spl, background=(0.1, 0.2, 0.3))
        losses.append(trainer.stats().loss)
    assert np.allclose(r0[4], losses, rtol=1e-5, atol=1e-7)
    tr = spl.transforms.cpu().numpy()
    util.assert_adam_close(r0[1][:, 3:7], tr[:, 3:7], cfg.lr_rotation, steps, "rotation")
    util.assert_adam_close(r0[1][:, 7:10], tr[:, 7:10], cfg.lr_scale, steps, "scale")
    util.assert_adam_close(r0[3], spl.raw_opacities.cpu().numpy(), cfg.lr_opac, steps, "opacity")
