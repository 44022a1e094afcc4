"""The multi-rank paths at the world sizes they exist for (BASELINE.json configs[3]: 8 ranks data parallel over cameras;
configs[4]: 8 ranks, one frame cut into strips of tile rows) — rehearsed on ONE GPU: `world` processes share the device and
a gloo process group carries the buffers (the gpurun boxes have a single GPU; the collectives' arithmetic, the union
listing, uneven strips, re-balancing, vis_weight counting and the dense fall-back do not care which transport sums).
Contract: SURVEY.md §8e — all-reduced update == the oracle's step on the mean of the K single-view gradients, replicas
bit-identical; a K-strip step == the single-GPU step."""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(world, target, args, timeout=900):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=timeout) for _ in range(world)], key=lambda r: r[0])
    finally:
        for p in procs:
            p.join(timeout=120)
    for p in procs:
        assert p.exitcode == 0
    return res


# ---------------------------------------------------------------------------------------------------------------
# data parallel over cameras
# ---------------------------------------------------------------------------------------------------------------
def _dp_problem(name):
    if name == "small":
        n, w, h = 3000, 128, 96
        sc = synth.make_scene(n, 0xD0, sh_degree=1, log_scale_range=(math.log(0.03), math.log(0.25)),
                              tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w), spread=1.8)
        return sc, w, h, 2.0
    sc, w, h = synth.config_scene("1m_1080p", 0)   # BASELINE.json configs[2] / [3]'s size
    return sc, w, h, 5.0


def _dp_cam(rank, world, w, h):
    cp = synth.default_camera_params(w, h)
    cp["rot_xyzw"] = util.quat_from_axis_angle((0, 1, 0), 0.12 / world * rank)   # K nearby views: the union stays a subset of the scene
    return cp


def _dp_worker(rank, world, port, q, problem, sparse, steps, allreduce="ring"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import brush_amd as ba
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    sc, w, h, median = _dp_problem(problem)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=3 + rank).view(np.int32)).to(dev)
    trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=median, process_group=dist.group.WORLD, sparse_exchange=sparse, allreduce=allreduce)
    batch = ba.SceneBatch(gt, util.hip_camera(ba, _dp_cam(rank, world, w, h)))
    rows, first = [], None
    for _ in range(steps):
        trainer.step(batch, spl, background=(0.1, 0.2, 0.3))
        rows.append(trainer.stats().exchange_rows)
        if first is None:
            first = (spl.transforms.cpu().numpy().copy(), spl.sh_coeffs.cpu().numpy().copy(), spl.raw_opacities.cpu().numpy().copy())
    trainer.sync_refine_stats()
    last = (spl.transforms.cpu().numpy(), spl.sh_coeffs.cpu().numpy(), spl.raw_opacities.cpu().numpy())
    # (only rank 0 ships the big arrays; the others ship checksums — replicas must be bit-identical)
    import hashlib
    def digest(arrs):
        hsh = hashlib.sha256()
        for a in arrs:
            hsh.update(np.ascontiguousarray(a).tobytes())
        return hsh.hexdigest()
    state = (trainer.state["vis_weight"].cpu().numpy(), trainer.state["refine_weight_norm"].cpu().numpy(), trainer.state["max_screen_size"].cpu().numpy())
    payload = (first, state) if rank == 0 else None
    q.put((rank, digest(first), digest(last), digest(state), rows, payload))
    dist.destroy_process_group()


def _check_against_oracle_mean(bo, problem, world, first, vis):
    import brush_amd as ba
    cfg = ba.TrainConfig()
    sc, w, h, median = _dp_problem(problem)
    extra = []
    for r in range(1, world):
        g = util.OracleTrainer(bo, cfg, median).step({k: v.copy() for k, v in sc.items()}, bo.camera(**_dp_cam(r, world, w, h)),
                                                      synth.synthetic_gt_packed(w, h, seed=3 + r), (0.1, 0.2, 0.3), dry_run=True)
        extra.append(g)
    ot = util.OracleTrainer(bo, cfg, median)
    ot.step(sc, bo.camera(**_dp_cam(0, world, w, h)), synth.synthetic_gt_packed(w, h, seed=3), (0.1, 0.2, 0.3), extra_grads=extra, world=world)
    tr, sh, op = first
    util.assert_adam_close(tr[:, 3:7], sc["transforms"][:, 3:7], cfg.lr_rotation, 1, "rotation")
    util.assert_adam_close(tr[:, 7:10], sc["transforms"][:, 7:10], cfg.lr_scale, 1, "scale")
    util.assert_adam_close(op, sc["raw_opac"], cfg.lr_opac, 1, "opacity")
    util.assert_adam_close(sh, sc["sh"], cfg.lr_coeffs_dc, 1, "sh")
    return ot


@pytest.mark.parametrize("world,sparse", [(4, True), (4, False), (8, True), (8, False)])
def test_dp_over_cameras_at_4_and_8_ranks_matches_the_oracle_mean_gradient(oracle_lib, world, sparse):
    steps = 2
    res = _run(world, _dp_worker, ("small", sparse, steps))
    for key in (1, 2, 3):   # parameters after step 1, after the last step, refine statistics: identical on every rank
        assert len({r[key] for r in res}) == 1, "replicas diverged (field %d)" % key
    rows = res[0][4]
    assert all(r[4] == rows for r in res)
    first, (vis, norm, scr) = res[0][5]
    n = first[0].shape[0]
    if sparse:   # K nearby views: their union is a real subset of the scene -> compact rows on both steps
        assert all(0 < r <= n // 2 for r in rows), rows
    else:
        assert rows == [0] * steps
    assert vis.max() == float(world * steps)            # vis_weight counts (rank, step) views
    ot = _check_against_oracle_mean(oracle_lib, "small", world, first, vis)
    # after ONE step vis_weight = number of views that reached the splat: compare the two-step count's support
    assert np.array_equal(ot.state["vis"] > 0, vis > 0) or np.mean((ot.state["vis"] > 0) != (vis > 0)) <= 2e-3


@pytest.mark.parametrize("world", [2, 4, 8])
def test_dp_dense_direct_allreduce_matches_the_oracle_mean_gradient(oracle_lib, world):
    """the dense gradient block summed by the DIRECT all-reduce (reduce-scatter + all-gather over point-to-point
    messages: comm.hip comm_allreduce_direct, restated for the hook path in parallel.allreduce_direct) instead of the collective
    library's all-reduce: same update as the oracle's step on the mean gradient, replicas bit-identical, at 2 / 4 / 8 ranks."""
    steps = 2
    res = _run(world, _dp_worker, ("small", False, steps, "direct"))
    for key in (1, 2, 3):
        assert len({r[key] for r in res}) == 1, "replicas diverged (field %d)" % key
    assert res[0][4] == [0] * steps
    first, (vis, norm, scr) = res[0][5]
    assert vis.max() == float(world * steps)
    _check_against_oracle_mean(oracle_lib, "small", world, first, vis)


def test_dp_4_ranks_at_1m_1080p_matches_the_oracle_mean_gradient(oracle_lib):
    """the data-parallel step at configs[2]'s size: four 1 M-splat replicas on one GPU, mask-keyed exchange"""
    world = 4
    res = _run(world, _dp_worker, ("1m", True, 1), timeout=1500)
    for key in (1, 2, 3):
        assert len({r[key] for r in res}) == 1
    rows = res[0][4]
    first, (vis, norm, scr) = res[0][5]
    assert 0 < rows[0] <= first[0].shape[0] // 2 and rows[0] == int((vis > 0).sum())
    _check_against_oracle_mean(oracle_lib, "1m", world, first, vis)


# ---------------------------------------------------------------------------------------------------------------
# one frame over 8 ranks
# ---------------------------------------------------------------------------------------------------------------
def _tile_problem():
    n, w, h = 12000, 208, 400        # 25 tile rows over 8 ranks: strips of 4 and 3 rows before the first re-cut
    sc = synth.make_scene(n, 0xE8, sh_degree=1, log_scale_range=(math.log(0.02), math.log(0.25)),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w), spread=2.0)
    cp = synth.default_camera_params(w, h)
    return sc, cp, w, h


def _tile_worker(rank, world, port, q, strip_loss, sparse, steps):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import brush_amd as ba
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    sc, cp, w, h = _tile_problem()
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=3).view(np.int32)).to(dev)
    trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=2.0, process_group=dist.group.WORLD, partition="tiles", sparse_exchange=sparse)
    trainer.rebalance_every = 1
    trainer.strip_loss = strip_loss
    batch = ba.SceneBatch(gt, util.hip_camera(ba, cp))
    losses, info, cuts = [], [], []
    from brush_amd.parallel import tile_rows_for_rank
    for _ in range(steps):
        cuts.append(tile_rows_for_rank((h + 15) // 16, rank, world, trainer._row_weights))
        trainer.step(batch, spl, background=(0.1, 0.2, 0.3))
        st = trainer.stats()
        info.append((trainer._strip_loss_now, st.exchange_rows))
        losses.append(trainer.reduce_loss(st))
    trainer.sync_refine_stats()
    q.put((rank, spl.transforms.cpu().numpy(), spl.sh_coeffs.cpu().numpy(), spl.raw_opacities.cpu().numpy(), losses,
           trainer.state["vis_weight"].cpu().numpy(), trainer.state["refine_weight_norm"].cpu().numpy(), trainer._row_weights, info, cuts))
    dist.destroy_process_group()


@pytest.mark.parametrize("strip_loss,sparse", [(True, True), (False, True), (True, False)])
def test_one_frame_over_8_uneven_rebalanced_strips_equals_the_single_gpu_step(dev, strip_loss, sparse):
    import brush_amd as ba
    world, steps = 8, 3
    res = _run(world, _tile_worker, (strip_loss, sparse, steps))
    r0 = res[0]
    for r in res[1:]:
        for a, b in zip(r0[1:7], r[1:7]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), "replicas diverged"
        assert r[7] == r0[7]
    # the strips tile the frame at every step, are uneven, and move when they are re-cut by blended intersections
    for s in range(steps):
        spans = [r[9][s] for r in res]
        assert spans[0][0] == 0 and spans[-1][1] == 25 and all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
        assert all(e > b for b, e in spans)
    assert len({e - b for b, e in [r[9][0] for r in res]}) > 1            # 25 rows over 8 ranks
    assert [r[9][0] for r in res] != [r[9][steps - 1] for r in res]       # re-balanced
    if sparse:
        assert all(0 < x[1] for x in r0[8])
    if strip_loss:   # wherever every strip is taller than the halo the strip-wise loss ran
        assert any(x[0] for x in r0[8])
    # single-GPU reference
    sc, cp, w, h = _tile_problem()
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
    util.assert_adam_close(r0[2], spl.sh_coeffs.cpu().numpy(), cfg.lr_coeffs_dc, steps, "sh")
    assert np.mean(r0[5] != trainer.state["vis_weight"].cpu().numpy()) <= 2e-3
    ref_norm = trainer.state["refine_weight_norm"].cpu().numpy()
    assert np.abs(r0[6] - ref_norm).max() <= 2e-3 * ref_norm.max() + 1e-12


# ---------------------------------------------------------------------------------------------------------------
# configs[4] at its own size: 6 M splats, 3840x2160, SH degree 3, ONE frame over 4 strips (4 processes on one GPU)
# ---------------------------------------------------------------------------------------------------------------
def _big_tile_worker(rank, world, port, q, steps):
    import hashlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import brush_amd as ba
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    sc, w, h = synth.config_scene("6m_4k", 3)
    cp = synth.default_camera_params(w, h)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    del sc
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=3).view(np.int32)).to(dev)
    trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=5.0, process_group=dist.group.WORLD, partition="tiles", sparse_exchange=True)
    trainer.rebalance_every = 1
    batch = ba.SceneBatch(gt, util.hip_camera(ba, cp))
    losses, info = [], []
    for _ in range(steps):
        trainer.step(batch, spl, background=(0.1, 0.2, 0.3))
        st = trainer.stats()
        info.append((trainer._strip_loss_now, st.exchange_rows, st.num_visible, st.num_intersections))
        losses.append(trainer.reduce_loss(st))
    hsh = hashlib.sha256()
    for t in (spl.transforms, spl.sh_coeffs, spl.raw_opacities):
        hsh.update(t.cpu().numpy().tobytes())
    # a sample of the parameters for the comparison with the single-GPU run (every 997th splat)
    sample = (spl.transforms[::997].cpu().numpy(), spl.sh_coeffs[::997].cpu().numpy(), spl.raw_opacities[::997].cpu().numpy()) if rank == 0 else None
    q.put((rank, hsh.hexdigest(), losses, info, sample))
    dist.destroy_process_group()


def test_one_6m_4k_sh3_frame_over_4_strips_equals_the_single_gpu_step(dev):
    """BASELINE.json configs[4] at full size through the tile-partition path: strip renders, 21-px halo exchange, strip-wise loss,
    mask-keyed gradient exchange with the refine-weight column, re-balanced cuts — against the same two steps on one GPU"""
    import brush_amd as ba
    world, steps = 4, 2
    res = _run(world, _big_tile_worker, (steps,), timeout=1700)
    assert len({r[1] for r in res}) == 1, "replicas diverged"
    r0 = res[0]
    assert all(x[0] for x in r0[3]) and all(x[1] > 0 for x in r0[3])          # strip-wise loss and compact rows on every step
    assert sum(r[3][0][3] for r in res) > 100_000_000                           # the strips' pair counts add up to the frame's
    sc, w, h = synth.config_scene("6m_4k", 3)
    cp = synth.default_camera_params(w, h)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    del sc
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=3).view(np.int32)).to(dev)
    cfg = ba.TrainConfig()
    trainer = ba.SplatTrainer(cfg, median_scene_scale=5.0)
    batch = ba.SceneBatch(gt, util.hip_camera(ba, cp))
    losses = []
    for _ in range(steps):
        trainer.step(batch, spl, background=(0.1, 0.2, 0.3))
        losses.append(trainer.stats().loss)
    assert np.allclose(r0[2], losses, rtol=2e-5, atol=1e-7)
    tr, sh, op = r0[4]
    util.assert_adam_close(tr[:, 3:7], spl.transforms[::997, 3:7].cpu().numpy(), cfg.lr_rotation, steps, "rotation")
    util.assert_adam_close(tr[:, 7:10], spl.transforms[::997, 7:10].cpu().numpy(), cfg.lr_scale, steps, "scale")
    util.assert_adam_close(op, spl.raw_opacities[::997].cpu().numpy(), cfg.lr_opac, steps, "opacity")
    util.assert_adam_close(sh, spl.sh_coeffs[::997].cpu().numpy(), cfg.lr_coeffs_dc, steps, "sh")


# ---------------------------------------------------------------------------------------------------------------
# configs[3]'s code path across ranks: data parallel over cameras AND refine in one loop
# (crates/brush-train/src/train.rs:431-663; SURVEY.md §8e: "rank 0 must decide ... or all ranks share a seeded RNG")
# ---------------------------------------------------------------------------------------------------------------
_LOOP_VIEWS = 5          # the ranks cycle through V views: rank r trains view (step * K + r) % V


def _loop_view(v, w, h):
    cp = synth.default_camera_params(w, h)
    cp["rot_xyzw"] = util.quat_from_axis_angle((0, 1, 0), 0.03 * v)
    cp["pos"] = (0.15 * v, 0.0, 0.0)
    return cp, synth.synthetic_gt_packed(w, h, seed=3 + v)


def _loop_cfg(ba):
    # thresholds low enough that the 3-step RefineRecord of this small scene selects splits of every kind
    return ba.TrainConfig(growth_grad_threshold=2e-6, split_at_screen_size=0.12)


def _digest(arrs):
    import hashlib
    hsh = hashlib.sha256()
    for a in arrs:
        hsh.update(np.ascontiguousarray(a).tobytes())
    return hsh.hexdigest()


def _loop_worker(rank, world, port, q, steps_a, steps_b):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import brush_amd as ba
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    sc, w, h, median = _dp_problem("small")
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    views = []
    for v in range(_LOOP_VIEWS):
        cp, gt = _loop_view(v, w, h)
        views.append(ba.SceneBatch(torch.from_numpy(gt.view(np.int32)).to(dev), util.hip_camera(ba, cp)))
    trainer = ba.SplatTrainer(_loop_cfg(ba), median_scene_scale=median, process_group=dist.group.WORLD, sparse_exchange=True)
    bg = (0.1, 0.2, 0.3)

    def params(s):
        return [s.transforms.cpu().numpy().copy(), s.sh_coeffs.cpu().numpy().copy(), s.raw_opacities.cpu().numpy().copy()]

    def adam():
        return {k: trainer.state[k].cpu().numpy().copy() for k in ("m1_t", "m2_t", "m1_sh", "m2_sh", "m1_o", "m2_o")}

    step = 0
    for _ in range(steps_a):
        trainer.step(views[(step * world + rank) % _LOOP_VIEWS], spl, background=bg)
        step += 1
    trainer.stats()
    trainer.sync_refine_stats()      # MAX over the ranks (refine() does it again: idempotent)
    before = params(spl)
    moments_before = adam()
    record = {k: trainer.state[k].cpu().numpy().copy() for k in ("refine_weight_norm", "vis_weight", "max_screen_size")}
    new, rs = trainer.refine(steps_a, spl)      # seed derived from the iteration: the same on every rank
    plan = {k: v.cpu().numpy().copy() for k, v in trainer.last_refine_plan.items()}
    after = params(new)
    moments_after = adam()
    bounds, median_after = trainer.bounds, trainer.median_scene_scale
    for _ in range(steps_b):
        trainer.step(views[(step * world + rank) % _LOOP_VIEWS], new, background=bg)
        step += 1
    trainer.stats()
    final = params(new)
    digs = (_digest(before + list(moments_before.values()) + list(record.values())),
            _digest(after + list(moments_after.values()) + [plan["keep"], plan["split"]]), _digest(final))
    payload = dict(before=before, moments_before=moments_before, record=record, plan=plan, after=after, moments_after=moments_after, final=final,
                   stats=(rs.num_added, rs.num_pruned, rs.total_splats), bounds=bounds, median_after=median_after) if rank == 0 else None
    q.put((rank, digs, new.num_splats(), payload))
    dist.destroy_process_group()


def _oracle_dp_steps(bo, cfg, median, scene, state, step0, world, nsteps, w, h):
    """`nsteps` data-parallel steps on the oracle: every step is rank 0's view with the other ranks' single-view gradients added and
    the mean taken (OracleTrainer.step(extra_grads, world)); returns the trainer (scene is updated in place)."""
    ot = util.OracleTrainer(bo, cfg, median)
    ot.state, ot.step_count = state, step0
    bg = (0.1, 0.2, 0.3)
    for s in range(nsteps):
        step = step0 + s
        extra = []
        for r in range(1, world):
            cp, gt = _loop_view((step * world + r) % _LOOP_VIEWS, w, h)
            probe = util.OracleTrainer(bo, cfg, median)
            probe.state, probe.step_count = ot.state, ot.step_count    # dry_run: reads nothing from it, changes nothing
            extra.append(probe.step({k: v.copy() for k, v in scene.items()}, bo.camera(**cp), gt, bg, dry_run=True))
        cp, gt = _loop_view((step * world) % _LOOP_VIEWS, w, h)
        ot.step(scene, bo.camera(**cp), gt, bg, extra_grads=extra, world=world)
    return ot


@pytest.mark.parametrize("world", [2, 4])
def test_dp_steps_refine_steps_replicas_identical_and_match_the_oracle(oracle_lib, world):
    """K ranks x (N steps over cycling views -> MAX-reduce of the RefineRecord -> refine(seed) -> N more steps): the replicas are
    bit-identical before the refine, after it (parameters, Adam moments, the plan itself) and after the further steps; the
    splat count, every parameter and the RefineRecord follow the oracle trainer fed the mean gradients, and the refine follows
    the oracle's refine given the same plan."""
    import brush_amd as ba
    from oracle import refine as orf
    steps_a, steps_b = 3, 2
    res = _run(world, _loop_worker, (steps_a, steps_b))
    for k in range(3):
        assert len({r[1][k] for r in res}) == 1, "replicas diverged (%s)" % ("before refine", "after refine", "after the further steps")[k]
    assert len({r[2] for r in res}) == 1
    p = res[0][3]
    cfg = _loop_cfg(ba)
    sc, w, h, median = _dp_problem("small")
    n, C = sc["transforms"].shape[0], sc["sh"].shape[1]
    # ---- phase A vs the oracle's mean-gradient steps
    osc = {k: v.copy() for k, v in sc.items()}
    z = lambda *s: np.zeros(s, np.float32)  # noqa: E731
    ost = dict(m1_t=z(n, 10), m2_t=z(n, 10), m1_sh=z(n, C * 3), m2_sh=z(n), m1_o=z(n, 1), m2_o=z(n, 1), refine=z(n), vis=z(n), screen=z(n))
    ot = _oracle_dp_steps(oracle_lib, cfg, median, osc, ost, 0, world, steps_a, w, h)
    tr, sh, op = p["before"]
    util.assert_adam_close(tr[:, 3:7], osc["transforms"][:, 3:7], cfg.lr_rotation, steps_a, "rotation")
    util.assert_adam_close(tr[:, 7:10], osc["transforms"][:, 7:10], cfg.lr_scale, steps_a, "scale")
    util.assert_adam_close(op, osc["raw_opac"], cfg.lr_opac, steps_a, "opacity")
    util.assert_adam_close(sh, osc["sh"], cfg.lr_coeffs_dc, steps_a, "sh")
    rec = p["record"]
    assert rec["vis_weight"].max() <= float(world * steps_a) and np.mean(rec["vis_weight"] != ot.state["vis"]) <= 2e-3
    # (after the first step the two trajectories differ by Adam's +-lr sign flips on noise gradients: radii agree closely, not bitwise)
    assert np.mean(np.abs(rec["max_screen_size"] - ot.state["screen"]) > 1e-3 * np.maximum(ot.state["screen"], 1e-6)) <= 2e-3
    assert np.abs(rec["refine_weight_norm"] - ot.state["refine"]).max() <= 2e-3 * ot.state["refine"].max() + 1e-12
    # ---- the refine: decisions and tensors vs the oracle's refine, given the replicas' state and the plan they all took
    keep, split = p["plan"]["keep"].astype(bool), p["plan"]["split"].astype(bool)
    bounds0 = orf.bounds_from_pos(0.8, tr[:, :3])      # the trainer had no bounds yet: refine takes them from the splats (train.rs:485)
    mask, _bad = orf.prune_mask(tr, sh, op, bounds0[0], bounds0[1])
    assert np.array_equal(keep, ~mask)
    rcfg = dict(split_at_screen_size=cfg.split_at_screen_size, growth_grad_threshold=cfg.growth_grad_threshold,
                growth_select_fraction=cfg.growth_select_fraction, iter=steps_a, total_train_iters=cfg.total_train_iters, opac_decay=cfg.opac_decay)
    bc = orf.budget_counts(n, keep, rec["vis_weight"], rec["refine_weight_norm"], rec["max_screen_size"], rcfg)
    vis = rec["vis_weight"] > 0
    assert not (split & ~keep).any() and not (split & ~((keep & vis) | bc["oversized"] | bc["above"])).any()
    added, pruned, total = p["stats"]
    assert pruned == int(mask.sum()) and added == int(split.sum()) and total == bc["n_keep"] + added == res[0][2]
    assert added > 0, "the scene was set up so that the refine splits something"
    mb = p["moments_before"]
    state = dict(transforms=tr, sh=sh, raw_opac=op, **mb)
    ref = orf.apply(state, keep, split, rcfg, rec["max_screen_size"])
    got = dict(transforms=p["after"][0], sh=p["after"][1], raw_opac=p["after"][2], **p["moments_after"])
    for k, r in ref.items():
        assert got[k].shape == r.shape, k
        assert np.allclose(got[k], r, rtol=2e-5, atol=2e-5), (k, float(np.abs(got[k] - r).max()))
    c, e = orf.bounds_from_pos(0.8, ref["transforms"][:, :3])
    assert np.allclose(p["bounds"][0], c, atol=1e-6) and np.allclose(p["bounds"][1], e, atol=1e-6)
    # ---- phase B: the oracle continues from the refined replica state (same count, same moments, fresh RefineRecord)
    n2 = total
    osc2 = dict(transforms=p["after"][0].copy(), sh=p["after"][1].copy(), raw_opac=p["after"][2].copy())
    ma = p["moments_after"]
    ost2 = dict(m1_t=ma["m1_t"].copy(), m2_t=ma["m2_t"].copy(), m1_sh=ma["m1_sh"].reshape(n2, C * 3).copy(), m2_sh=ma["m2_sh"].copy(),
                m1_o=ma["m1_o"].reshape(n2, 1).copy(), m2_o=ma["m2_o"].reshape(n2, 1).copy(), refine=z(n2), vis=z(n2), screen=z(n2))
    _oracle_dp_steps(oracle_lib, cfg, p["median_after"], osc2, ost2, steps_a, world, steps_b, w, h)
    tr2, sh2, op2 = p["final"]
    assert tr2.shape[0] == n2
    util.assert_adam_close(tr2[:, 3:7], osc2["transforms"][:, 3:7], cfg.lr_rotation, steps_b, "rotation after refine")
    util.assert_adam_close(tr2[:, 7:10], osc2["transforms"][:, 7:10], cfg.lr_scale, steps_b, "scale after refine")
    util.assert_adam_close(op2, osc2["raw_opac"], cfg.lr_opac, steps_b, "opacity after refine")
    util.assert_adam_close(sh2, osc2["sh"], cfg.lr_coeffs_dc, steps_b, "sh after refine")
