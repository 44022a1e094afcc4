"""The single-GPU train step does not zero-fill its gradient span (only the refine-weight vector): K18 writes the rows of the
splats that received a gradient and marks them in the sign bit of their refine weight, the update kernel takes every other row
as zero (api.hip bh_train_step, `grad_rows_marked`).  That must be invisible.  The span
is filled with NaNs before every step (bh_debug_fill_train_scratch): one read of a row nobody wrote would poison a
parameter, a moment or a statistic for good.  The results are then compared with the zero-filling step
(BH_TRAIN_ZERO_GRADS, which is also what the multi-GPU exchange path runs) — to the run-to-run tolerance of a step, not
bit for bit: the backward sums a splat's per-tile gradients with float atomics, so two runs of the SAME step already
differ in the last bits."""
import math
import os

import numpy as np
import pytest
import torch

from brush_amd import synth
import util

pytestmark = pytest.mark.gpu


def _views(w, h, k):
    cp = synth.default_camera_params(w, h)
    out = []
    for i in range(k):
        c = dict(cp)
        # swing the camera so that every view sees a different subset (splats drop out of and into the frustum)
        c["rot_xyzw"] = util.quat_from_axis_angle((0, 1, 0), math.radians(-35 + 70 * i / max(1, k - 1)))
        out.append(c)
    return out


def _run(ba, dev, zero_fill, sc, cams, gt, sh_degree, mip, min_scale_views, steps, poison=False):
    if zero_fill:
        os.environ["BH_TRAIN_ZERO_GRADS"] = "1"
    else:
        os.environ.pop("BH_TRAIN_ZERO_GRADS", None)
    try:
        # the knob is read once, at bh_create.  The NaN fill is a test hook: only the -DBH_TEST_HOOKS build exports it
        from brush_amd import _ffi
        ctx = ba.Context(dev, lib=_ffi.load_test_hooks() if poison else None)
    finally:
        os.environ.pop("BH_TRAIN_ZERO_GRADS", None)
    cfg = ba.TrainConfig()
    cfg.render_mip = bool(mip)
    trainer = ba.SplatTrainer(cfg, median_scene_scale=3.0, ctx=ctx, seed=1234)
    spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
    if min_scale_views:   # a 3D-filter floor: the fold's backward runs between K18 and the update, on written and unwritten rows alike
        spl.with_min_scale(torch.full((spl.num_splats(),), 0.05, device=dev))
    gt_t = torch.from_numpy(gt.view(np.int32)).to(dev)
    for s in range(steps):
        batch = ba.SceneBatch(gt_t, util.hip_camera(ba, cams[s % len(cams)]))
        if poison and s > 0:
            ctx.check(ctx.lib.bh_debug_fill_train_scratch(ctx._h, 0x7FC00000))   # quiet NaN in every word of the gradient scratch
        trainer.step(batch, spl)
    ctx.sync()
    out = {"transforms": spl.transforms.clone(), "sh": spl.sh_coeffs.clone(), "opac": spl.raw_opacities.clone()}
    out.update({k: v.clone() for k, v in trainer.state.items()})
    stats = trainer.stats()
    ctx.close()
    return out, stats


@pytest.mark.parametrize("sh_degree,mip,min_scale_views,n,w,h", [
    (0, False, False, 20000, 320, 192),
    (3, False, False, 12000, 256, 160),
    (2, True, True, 8000, 200, 120),
    (1, False, False, 300, 64, 48),    # fewer splats than one update block; ragged everything
])
def test_masked_rows_equal_zero_filled(dev, sh_degree, mip, min_scale_views, n, w, h):
    import brush_amd as ba
    sc = synth.make_scene(n, 0x51 + sh_degree, sh_degree=sh_degree, log_scale_range=(math.log(0.02), math.log(0.2)),
                          tan_half_fov=(math.tan(math.radians(50)), math.tan(math.radians(50)) * h / w))
    gt = synth.synthetic_gt_packed(w, h)
    cams = _views(w, h, 4)
    steps = 7
    a, sa = _run(ba, dev, False, sc, cams, gt, sh_degree, mip, min_scale_views, steps, poison=True)
    b, sb = _run(ba, dev, True, sc, cams, gt, sh_degree, mip, min_scale_views, steps)
    c, sc2 = _run(ba, dev, True, sc, cams, gt, sh_degree, mip, min_scale_views, steps)   # the yardstick: the same path twice
    assert 0 < sa.num_visible < n   # some rows are written, some are not
    for k in a:
        assert bool(torch.isfinite(a[k]).all()), k   # no NaN came through
    assert abs(sa.num_visible - sb.num_visible) <= max(2, n // 2000)
    assert abs(sa.loss - sb.loss) <= 1e-5 * max(1.0, abs(sb.loss))
    cfg = ba.TrainConfig()
    lr = {"transforms": max(cfg.lr_rotation, cfg.lr_scale), "sh": cfg.lr_coeffs_dc, "opac": cfg.lr_opac}
    for k in ("transforms", "sh", "opac"):
        # a row read as garbage (or as zero when it held a gradient) moves a parameter by about lr per step; run-to-run
        # noise moves a handful of elements (a first-step sign flip of a ~0 gradient) and the rest by far less
        d_ab = (a[k] - b[k]).abs()
        d_bc = (b[k] - c[k]).abs()
        assert float(d_ab.mean()) <= 2.0 * float(d_bc.mean()) + 1e-3 * lr[k], (k, float(d_ab.mean()), float(d_bc.mean()))
        assert float((d_ab > 0.5 * lr[k]).float().mean()) <= 2.0 * float((d_bc > 0.5 * lr[k]).float().mean()) + 1e-3, k
    for k in ("m1_t", "m2_t", "m1_sh", "m2_sh", "m1_o", "m2_o", "refine_weight_norm"):
        ref = float(b[k].abs().max())
        assert float((a[k] - b[k]).abs().max()) <= 4.0 * float((b[k] - c[k]).abs().max()) + 1e-3 * ref, k
    assert float((a["vis_weight"] != b["vis_weight"]).float().mean()) <= 2e-3
    # the scene is seen from four directions: the visibility pattern really changed between steps
    assert float((a["vis_weight"] > 0).float().mean()) > float(sa.num_visible) / n


def test_masked_and_exchange_steps_alternate(dev):
    """One context, steps alternating between the single-GPU path (masked rows, scratch poisoned with NaNs in front of them) and
    the exchange path (a hook: the whole span is zero-filled and handed over): neither leaves anything behind that the other
    trips over — the run ends where an all-zero-filling run ends."""
    import brush_amd as ba
    from brush_amd import _ffi
    n, w, h, sh_degree = 9000, 224, 144, 2
    sc = synth.make_scene(n, 0x77, sh_degree=sh_degree, log_scale_range=(math.log(0.02), math.log(0.2)),
                          tan_half_fov=(math.tan(math.radians(50)), math.tan(math.radians(50)) * h / w))
    gt = synth.synthetic_gt_packed(w, h)
    cams = _views(w, h, 4)
    steps = 8
    seen = []

    def run(alternate):
        if not alternate:
            os.environ["BH_TRAIN_ZERO_GRADS"] = "1"
        try:
            ctx = ba.Context(dev, lib=_ffi.load_test_hooks() if alternate else None)   # (the NaN fill is a test hook)
        finally:
            os.environ.pop("BH_TRAIN_ZERO_GRADS", None)
        trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0, ctx=ctx, seed=99, sparse_exchange=False)
        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
        gt_t = torch.from_numpy(gt.view(np.int32)).to(dev)

        def hook_fn(_user, _ptr, count):   # one rank: the sum over the ranks is the buffer itself
            seen.append(int(count))
            return 0
        hook = _ffi.GRAD_HOOK(hook_fn)
        for s in range(steps):
            with_hook = alternate and (s % 2 == 1)
            trainer.pg, trainer._hook, trainer._world = ("one rank", hook, 1) if with_hook else (None, None, 1)
            if alternate and not with_hook and s > 0:
                ctx.check(ctx.lib.bh_debug_fill_train_scratch(ctx._h, 0x7FC00000))
            trainer.step(ba.SceneBatch(gt_t, util.hip_camera(ba, cams[s % len(cams)])), spl)
        ctx.sync()
        out = {"transforms": spl.transforms.clone(), "sh": spl.sh_coeffs.clone(), "opac": spl.raw_opacities.clone()}
        out.update({k: v.clone() for k, v in trainer.state.items()})
        ctx.close()
        return out

    a = run(True)
    b = run(False)
    c = run(False)
    assert len(seen) == steps // 2 and all(x > n for x in seen)   # the hook saw visible + the gradient sections
    cfg = ba.TrainConfig()
    lr = {"transforms": max(cfg.lr_rotation, cfg.lr_scale), "sh": cfg.lr_coeffs_dc, "opac": cfg.lr_opac}
    for k in a:
        assert bool(torch.isfinite(a[k]).all()), k
    for k in ("transforms", "sh", "opac"):
        d_ab, d_bc = (a[k] - b[k]).abs(), (b[k] - c[k]).abs()
        assert float(d_ab.mean()) <= 2.0 * float(d_bc.mean()) + 1e-3 * lr[k], (k, float(d_ab.mean()), float(d_bc.mean()))
        assert float((d_ab > 0.5 * lr[k]).float().mean()) <= 2.0 * float((d_bc > 0.5 * lr[k]).float().mean()) + 1e-3, k


@pytest.mark.parametrize("sh_degree", [0, 2])
def test_dormant_splats_are_skipped_without_changing_a_bit(dev, sh_degree):
    """The update kernel leaves out splats whose Adam moments are all zero and that neither received a gradient nor were reached
    by the view (optim.hip: the mark is the sign of their m2_sh, -0.0).  On a ONE-tile image every splat has at most one (splat,
    tile) pair, so the backward's float atomics add each gradient once into a zero: the step is deterministic, and a run with the
    skip must equal a run without it (BH_UPDATE_NO_DORMANT) bit for bit — parameters, every moment (the sign of a zero m2_sh
    aside), the refine statistics — while most splats are dormant (outside the tiny frustum or behind the saturated front)."""
    import brush_amd as ba
    n, w, h = 6000, 16, 16
    sc = synth.make_scene(n, 0xD0A, sh_degree=sh_degree, log_scale_range=(math.log(0.05), math.log(0.4)),
                          tan_half_fov=(math.tan(math.radians(50)), math.tan(math.radians(50))))
    cams = _views(w, h, 3)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h).view(np.int32)).to(dev)
    steps = 9
    runs = {}
    for key in ("skip", "full"):
        if key == "full":
            os.environ["BH_UPDATE_NO_DORMANT"] = "1"
        try:
            ctx = ba.Context(dev)
        finally:
            os.environ.pop("BH_UPDATE_NO_DORMANT", None)
        try:
            spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
            tr = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0, ctx=ctx, seed=77)   # the default stochastic step (device noise)
            for s in range(steps):
                tr.step(ba.SceneBatch(gt, util.hip_camera(ba, cams[s % len(cams)])), spl)
            ctx.sync()
            out = {"transforms": spl.transforms.clone(), "sh": spl.sh_coeffs.clone(), "opac": spl.raw_opacities.clone()}
            out.update({k: v.clone() for k, v in tr.state.items()})
            runs[key] = out
        finally:
            ctx.close()
    a, b = runs["skip"], runs["full"]
    for k in a:
        if k == "m2_sh":
            assert torch.equal(a[k], b[k]), k                                   # by value: -0.0 == +0.0
        else:
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), k   # bit for bit
    marks = (a["m2_sh"].view(torch.int32) == -2147483648)
    moments_zero = (a["m1_t"].abs().sum(1) == 0) & (a["m2_t"].abs().sum(1) == 0) & (a["m1_sh"].reshape(n, -1).abs().sum(1) == 0) & (a["m1_o"] == 0) & (a["m2_o"] == 0) & (a["m2_sh"] == 0)
    assert torch.equal(marks, moments_zero), "the mark must say exactly: every moment of this splat is zero"
    assert 0.25 < float(marks.float().mean()) < 1.0, float(marks.float().mean())         # many splats dormant, some trained
