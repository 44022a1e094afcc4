"""RCCL inside the library (bh_comm_*): what can be exercised on ONE GPU — RCCL binds at run time,
a 1-rank communicator initialises, collectives are identities, and a train step with the built-in exchange
equals the plain step.  (RCCL refuses two ranks on one device; the N-rank logic — which floats are summed, the 1/K
scale — is the same exchange-buffer contract the gloo and 2-process GPU tests cover through the hook.)"""
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
import util

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_and_builtin_exchange(dev):
    import brush_amd as ba
    uid = ba.Context.comm_unique_id()
    assert len(uid) == 128 and any(b != 0 for b in uid)
    ctx = ba.Context(dev)           # a private context: the shared one stays communicator-free
    assert ctx.comm_world() == 1
    with pytest.raises(ba.BrushHipError):
        ctx.allreduce_sum(torch.ones(4, device=dev))          # no communicator yet
    ctx.comm_init(0, 1, uid)
    assert ctx.comm_world() == 1
    with pytest.raises(ba.BrushHipError):
        ctx.comm_init(0, 1, uid)                              # already attached
    x = torch.arange(1000, dtype=torch.float32, device=dev)
    y = x.clone()
    ctx.allreduce_sum(y)
    ctx.allreduce_max(y)
    ctx.sync()
    assert torch.equal(x, y)
    # train step: native exchange (world 1) == plain step
    n, w, h = 3000, 128, 96
    sc = synth.make_scene(n, 0xC0, sh_degree=1, log_scale_range=(math.log(0.02), math.log(0.2)),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w))
    cp = synth.default_camera_params(w, h)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h).view(np.int32)).to(dev)
    cam = util.hip_camera(ba, cp)
    outs = []
    for native in (True, False):
        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
        tr = ba.SplatTrainer(ba.TrainConfig(mean_noise_weight=0.0), median_scene_scale=3.0, ctx=ctx, native_comm=native)
        for _ in range(2):
            tr.step(ba.SceneBatch(gt, cam), spl)
        outs.append((tr.stats(ctx).loss, spl))
        tr.sync_refine_stats()
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6 * max(1.0, abs(outs[1][0]))
    util.assert_adam_close(outs[0][1].transforms[:, 7:].cpu().numpy(), outs[1][1].transforms[:, 7:].cpu().numpy(), 5e-3, 2, "scale")
    ctx.comm_destroy()
    assert ctx.comm_world() == 1
    ctx.close()
