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
    return synth.make_scene(N, 0xD0, sh_degree=1, log_scale_range=(math.log(0.03), math.log(0.25)),
                            tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * H / W))


def _worker(rank, world, port, q):
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
    trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=2.0, process_group=dist.group.WORLD)
    batch = ba.SceneBatch(gt, util.hip_camera(ba, _cam_params(rank)))
    out = []
    for _ in range(2):
        trainer.step(batch, spl, background=(0.1, 0.2, 0.3))
        trainer.stats()
        out.append((spl.transforms.cpu().numpy().copy(), spl.sh_coeffs.cpu().numpy().copy(), spl.raw_opacities.cpu().numpy().copy()))
    trainer.sync_refine_stats()  # the running maxima are rank-local until refine asks for them
    q.put((rank, out, trainer.state["vis_weight"].cpu().numpy(), trainer.state["refine_weight_norm"].cpu().numpy(),
           trainer.state["max_screen_size"].cpu().numpy()))
    dist.destroy_process_group()


def test_two_rank_step_matches_oracle_mean_gradient(oracle_lib):
    import brush_amd as ba
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, o0, vis0, norm0, scr0), (_, o1, vis1, norm1, scr1) = res
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
