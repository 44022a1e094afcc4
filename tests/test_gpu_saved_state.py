"""The backward takes its forward's SAVED state: SplatBwdOps::{rasterize_bwd, project_bwd} receive the tensors
RenderBackwards saved (crates/brush-render/src/bwd/burn_glue.rs:62-92, 336-371, consumed at :121-182), so two render nodes may be
alive in one autodiff graph.  Through the C ABI (ctypes):

  forward A, forward B, backward_saved(A)                -> BH_ERR_STATE (A's buffers now hold B), never B's gradients;
  forward A, retain(A), forward B, backward_saved(A / B) -> each the oracle's gradients of ITS view; release recycles the blocks."""
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
import util
from test_gpu_backward import assert_grads_match

pytestmark = pytest.mark.gpu


def _two_views(w, h):
    a = synth.default_camera_params(w, h)
    b = dict(a)
    b["pos"] = (1.2, -0.3, -1.0)
    b["rot_xyzw"] = util.quat_from_axis_angle((0, 1, 0), math.radians(-9.0))
    return a, b


def _oracle(bo, sc, cp, w, h, bg, v_out):
    p = {k: v for k, v in cp.items() if k not in ("img_w", "img_h")}
    ref = bo.Render().forward(bo.camera(img_w=w, img_h=h, **p), sc["transforms"], sc["sh"], sc["raw_opac"], bg=bg, flags=bo.FLAG_BWD_INFO)
    ref.backward(v_out)
    return ref


def _as_res(node, grads, aux_of):
    from brush_amd import host
    ctx = node.ctx
    vc = host._view(ctx.lib.bh_last_v_combined(ctx._h), (max(node.out.num_listed_splats, 1), 10), torch.float32, node.splats.device).clone()
    return dict(grads, aux=aux_of, v_combined=vc)


@pytest.mark.parametrize("sliced", [False, True])
def test_forward_a_forward_b_backward_a(dev, oracle_lib, sliced):
    import brush_amd as ba
    from brush_amd import host
    bo = oracle_lib
    n, w, h = 20000, 256, 160
    sc = synth.make_scene(n, 0x5A7ED, sh_degree=1, log_scale_range=(math.log(0.03), math.log(0.3)),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w))
    cpa, cpb = _two_views(w, h)
    bg = (0.3, 0.1, 0.2)
    rng = np.random.default_rng(5)
    v_out = rng.normal(size=(h, w, 4)).astype(np.float32)
    v_dev = torch.from_numpy(v_out).to(dev)
    ref = {"A": _oracle(bo, sc, cpa, w, h, bg, v_out), "B": _oracle(bo, sc, cpb, w, h, bg, v_out)}
    assert util.rel_linf(ref["A"].get("v_transforms"), ref["B"].get("v_transforms")) > 1e-2   # the views differ: a mix-up would show
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    ctx = ba.Context(dev)
    try:
        cams = {"A": util.hip_camera(ba, cpa), "B": util.hip_camera(ba, cpb)}
        if sliced:   # seed both views' per-tile cut tables, so that the retained forwards below are cut frames
            for k in ("A", "B"):
                ba.render_splats(spl, cams[k], (w, h), bg, ba.RasterPass.Backward, ctx=ctx, sliced=True, copy=False)

        def aux_of(node):
            return host._aux_from(node.out, n, w, h, dev, True, None)

        # ---- not retained: stale -> a loud error, and the live forward is unharmed
        A = ba.render_splats_diff(spl, cams["A"], (w, h), bg, ctx=ctx, sliced=sliced)
        B = ba.render_splats_diff(spl, cams["B"], (w, h), bg, ctx=ctx, sliced=sliced)
        assert B.out.generation == A.out.generation + 1
        with pytest.raises(ba.BrushHipError, match="stale"):
            A.backward(v_dev)
        auxB = aux_of(B)
        assert_grads_match(_as_res(B, B.backward(v_dev), auxB), ref["B"])
        # ---- retained: both alive, either order, and again
        A = ba.render_splats_diff(spl, cams["A"], (w, h), bg, ctx=ctx, retain=True, sliced=sliced)
        imgA = A.img.clone()
        auxA = aux_of(A)
        B = ba.render_splats_diff(spl, cams["B"], (w, h), bg, ctx=ctx, retain=True, sliced=sliced)
        auxB = aux_of(B)
        C_ = ba.render_splats(spl, cams["B"], (w, h), bg, ba.RasterPass.Forward, ctx=ctx)   # an eval render in between
        assert torch.equal(A.img, imgA), "a retained forward's image must survive later forwards"
        assert float(np.abs(imgA.cpu().numpy() - ref["A"].image()).max()) <= 1e-6
        assert_grads_match(_as_res(A, A.backward(v_dev), auxA), ref["A"])
        assert_grads_match(_as_res(B, B.backward(v_dev), auxB), ref["B"])
        assert_grads_match(_as_res(A, A.backward(v_dev), auxA), ref["A"])
        with pytest.raises(ba.BrushHipError):   # retained once
            ctx.check(ctx.lib.bh_render_retain(ctx._h, A.out))
        A.release()
        B.release()
        with pytest.raises(ba.BrushHipError):
            A.backward(v_dev)
        del C_
        # ---- the plain "last forward" shorthand still works, and refuses when there is no live forward
        res = ba.render_splats_bwd(spl, cams["A"], (w, h), bg, v_dev, ctx=ctx, sliced=sliced)
        assert_grads_match(res, ref["A"])
    finally:
        ctx.close()


def test_retain_release_cycle_allocates_nothing_in_steady_state(dev):
    """every step: forward, retain, (another forward), backward, release — released blocks go back to the arena / its pool"""
    import brush_amd as ba
    n, w, h = 50000, 512, 320
    sc = synth.make_scene(n, 0x77, sh_degree=0, log_scale_range=(math.log(0.03), math.log(0.3)),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w))
    cpa, cpb = _two_views(w, h)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    v = torch.full((h, w, 4), 1e-4, device=dev)
    ctx = ba.Context(dev)
    try:
        ca, cb = util.hip_camera(ba, cpa), util.hip_camera(ba, cpb)
        free = []
        for i in range(40):
            A = ba.render_splats_diff(spl, ca, (w, h), ctx=ctx, retain=True)
            ba.render_splats(spl, cb, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, copy=False)
            A.backward(v)
            A.release()
            if i in (9, 39):
                ctx.sync()
                torch.cuda.synchronize(dev)
                free.append(torch.cuda.mem_get_info(dev)[0])
        assert free[0] - free[1] < (8 << 20), "device memory kept shrinking over 30 retain/release cycles: %d -> %d" % (free[0], free[1])
    finally:
        ctx.close()
