"""Backward parity through the C ABI: rasterize_bwd + project_bwd vs the oracle.

Tolerance: the per-(splat,tile) partial sums are accumulated across pixels by a wave
reduction and across tiles by float atomics, so gradients are compared relative to the
largest gradient of each tensor: |d| <= 1e-4 * max|g| (north_star: 1e-4), in practice ~1e-6."""
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
import util

pytestmark = pytest.mark.gpu
GRAD_TOL = 1e-4
# Per-ELEMENT mixed tolerance beside the per-tensor one (the reference's own gradient criterion is mixed absolute + relative per
# element: crates/brush-bench-test/tests/finite_diff.rs:218-256, abs 5e-5 + 1 % of max(|num|, |an|) on gradients of O(0.05), i.e. an
# absolute floor of ~1e-3 of the largest gradient).  Here, HIP vs oracle: |d| <= GRAD_ABS * max|g| + GRAD_REL * |ref| for every entry —
# the floor is the float-atomic / wave-reduction ordering noise of sums whose terms are as large as the largest gradient
# (2^-23 x a few terms), the relative part bounds every gradient that stands clear of that floor to 0.1 %: a small gradient
# can no longer be 100 % wrong and pass.
GRAD_ABS = 1e-5
GRAD_REL = 1e-3


def mixed_violations(a, b, scale=None, abs_frac=GRAD_ABS, rel=GRAD_REL):
    """Entries of a (HIP) outside abs_frac * max|b| + rel * |b| of b (oracle): (count, worst excess ratio)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = float(np.abs(b).max()) if scale is None else scale
    tol = abs_frac * max(scale, 1e-30) + rel * np.abs(b)
    d = np.abs(a - b)
    bad = d > tol
    return int(bad.sum()), float((d / tol).max()) if d.size else 0.0


def run_both(ba, bo, dev, scene, cp, w, h, v_out, bg=(0.0, 0.0, 0.0), pass_=None, mip=False):
    pass_ = pass_ or ba.RasterPass.Backward
    spl = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], render_mip=mip, device=dev)
    res = ba.render_splats_bwd(spl, util.hip_camera(ba, cp), (w, h), bg, torch.from_numpy(v_out).to(dev), pass_)
    flags = bo.FLAG_BWD_INFO | (bo.FLAG_SMOOTH_CUTOFF if pass_.smooth_cutoff() else 0) | (bo.FLAG_MIP if mip else 0)
    p = {k: v for k, v in cp.items() if k not in ("img_w", "img_h")}
    ref = bo.Render().forward(bo.camera(img_w=w, img_h=h, **p), scene["transforms"], scene["sh"], scene["raw_opac"], bg=bg, flags=flags)
    ref.backward(v_out)
    return res, ref


def _v_combined_in_oracle_rows(res, ref):
    """The HIP accumulator in the oracle's row numbering.  A frame rendered with per-tile cuts numbers only the splats that own a
    listed pair (BhRenderOut.num_listed_splats, a sub-sequence of the full depth order); the others blended nowhere: zero rows."""
    vc = res["v_combined"].cpu().numpy().reshape(-1, 10)
    aux = res["aux"]
    nv = ref.num_visible
    if getattr(aux, "num_listed_splats", nv) == nv and vc.shape[0] == max(nv, 1):
        return vc.reshape(-1)
    gfc_h = util.u32(aux.global_from_compact_gid)
    gfc_o = ref.get("global_from_compact_gid")[:nv]
    pos = np.full(int(gfc_o.max()) + 1 if nv else 1, -1, np.int64)
    pos[gfc_o] = np.arange(nv)
    rows = pos[gfc_h]
    assert np.all(rows >= 0) and np.all(np.diff(rows) > 0), "listed splats are not a sub-sequence of the depth order"
    full = np.zeros((max(nv, 1), 10), np.float32)
    full[rows] = vc[:gfc_h.size]
    return full.reshape(-1)


def assert_grads_match(res, ref, tol=GRAD_TOL):
    n = res["v_transforms"].shape[0]
    pairs = [("v_combined", _v_combined_in_oracle_rows(res, ref), ref.get("v_combined")),
             ("v_transforms", res["v_transforms"].cpu().numpy().reshape(-1), ref.get("v_transforms")),
             ("v_sh", res["v_sh_coeffs"].cpu().numpy().reshape(-1), ref.get("v_coeffs")),
             ("v_raw_opac", res["v_raw_opacities"].cpu().numpy(), ref.get("v_raw_opac")),
             ("v_refine", res["v_refine_weight"].cpu().numpy(), ref.get("v_refine"))]
    for name, a, b in pairs:
        assert np.isfinite(a).all(), name
        if name == "v_combined":
            a, b = a.reshape(-1, 10), b.reshape(-1, 10)
            for lane in range(10):  # lanes have very different magnitudes
                assert util.rel_linf(a[:, lane], b[:, lane]) <= tol, (name, lane, util.rel_linf(a[:, lane], b[:, lane]))
                assert mixed_violations(a[:, lane], b[:, lane])[0] == 0, (name, lane, "per-element abs+rel", mixed_violations(a[:, lane], b[:, lane]))
        elif name == "v_transforms":
            a, b = a.reshape(n, 10), b.reshape(n, 10)
            for sl in (slice(0, 3), slice(3, 7), slice(7, 10)):
                assert util.rel_linf(a[:, sl], b[:, sl]) <= tol, (name, sl, util.rel_linf(a[:, sl], b[:, sl]))
                assert mixed_violations(a[:, sl], b[:, sl])[0] == 0, (name, sl, "per-element abs+rel", mixed_violations(a[:, sl], b[:, sl]))
        else:
            assert util.rel_linf(a, b) <= tol, (name, util.rel_linf(a, b))
            assert mixed_violations(a, b)[0] == 0, (name, "per-element abs+rel", mixed_violations(a, b))
    # zero pattern: splats without gradient are exactly zero in both
    za = res["v_transforms"].cpu().numpy().reshape(n, 10)
    zb = ref.get("v_transforms").reshape(n, 10)
    assert np.array_equal(np.all(za == 0, axis=1), np.all(zb == 0, axis=1))


@pytest.mark.parametrize("sh_degree,mip,smooth", [(0, False, False), (3, False, False), (2, True, False), (1, False, True), (4, False, False)])
def test_config0_grads_vs_oracle(dev, oracle_lib, sh_degree, mip, smooth):
    import brush_amd as ba
    scene, w, h = synth.config_scene("10k_256", sh_degree)
    cp = synth.default_camera_params(w, h)
    rng = np.random.default_rng(sh_degree)
    v_out = (rng.uniform(-1, 1, (h, w, 4)) / (h * w)).astype(np.float32)
    pass_ = ba.RasterPass.BackwardSmoothCutoff if smooth else ba.RasterPass.Backward
    res, ref = run_both(ba, oracle_lib, dev, scene, cp, w, h, v_out, bg=(0.2, 0.3, 0.1), pass_=pass_, mip=mip)
    assert np.abs(res["img"].cpu().numpy() - ref.image()).max() <= 1e-6
    assert_grads_match(res, ref)


def test_mean_image_grad_like_reference_bench(dev, oracle_lib):
    """benches.rs:193: gradient of mean(out_img)."""
    import brush_amd as ba
    scene, w, h = synth.config_scene("10k_256", 0)
    cp = synth.default_camera_params(w, h)
    v_out = np.full((h, w, 4), 1.0 / (h * w * 4), np.float32)
    res, ref = run_both(ba, oracle_lib, dev, scene, cp, w, h, v_out)
    assert_grads_match(res, ref)


@pytest.mark.parametrize("w,h", [(1, 1), (17, 33), (123, 82), (300, 200)])
def test_grads_ragged_sizes_rotated_camera(dev, oracle_lib, w, h):
    import brush_amd as ba
    scene = synth.make_scene(2500, 0x91, sh_degree=1, log_scale_range=(math.log(0.03), math.log(0.4)))
    cp = dict(pos=(0.3, 0.2, -0.6), rot_xyzw=util.quat_from_axis_angle((1.0, 0.3, -0.2), -0.2), fov_x=0.9, fov_y=0.8, center_uv=(0.52, 0.47))
    rng = np.random.default_rng(w * h)
    v_out = (rng.uniform(-1, 1, (h, w, 4)) / (h * w)).astype(np.float32)
    res, ref = run_both(ba, oracle_lib, dev, scene, cp, w, h, v_out, bg=(0.7, 0.1, 0.4))
    assert_grads_match(res, ref)


def test_finite_difference_on_gpu(dev):
    """finite_diff.rs:210-273 run against the HIP path itself (smooth cutoff, eps 3e-4)."""
    import brush_amd as ba
    scene = util.base_scene()
    cam = util.hip_camera(ba, util.STD_CAM)
    w = h = 32
    v_out = torch.full((h, w, 4), 1.0 / (h * w * 4), device=dev)

    def value(sc):
        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
        img, _ = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.BackwardSmoothCutoff)
        return float(img.double().mean().item())
    res = ba.render_splats_bwd(ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev), cam, (w, h), (0, 0, 0), v_out,
                               ba.RasterPass.BackwardSmoothCutoff)
    vt = res["v_transforms"].cpu().numpy(); vo = res["v_raw_opacities"].cpu().numpy(); vs = res["v_sh_coeffs"].cpu().numpy()
    eps = 3e-4
    for kind, i, c in [("tr", 0, 0), ("tr", 0, 2), ("tr", 1, 1), ("tr", 0, 3), ("tr", 1, 5), ("tr", 0, 7), ("tr", 1, 8), ("sh", 0, 0), ("sh", 2, 2), ("op", 0, 0), ("op", 2, 0)]:
        def pert(d):
            s = {k: v.copy() for k, v in scene.items()}
            if kind == "tr":
                s["transforms"][i, c] += np.float32(d)
            elif kind == "sh":
                s["sh"][i, 0, c] += np.float32(d)
            else:
                s["raw_opac"][i] += np.float32(d)
            return value(s)
        num = (pert(eps) - pert(-eps)) / (2 * eps)
        an = float(vt[i, c] if kind == "tr" else (vs[i, 0, c] if kind == "sh" else vo[i]))
        assert abs(num - an) <= 2e-4 + 0.02 * max(abs(num), abs(an)), (kind, i, c, num, an)


def test_backward_requires_bwd_forward(dev):
    """State errors surface as error codes (bwd/burn_glue.rs:281-284 assert)."""
    import brush_amd as ba
    sc = util.base_scene()
    ctx = ba.Context(dev)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    ba.render_splats(spl, util.hip_camera(ba, util.STD_CAM), (32, 32), (0, 0, 0), ba.RasterPass.Forward, ctx=ctx)
    z = torch.zeros(32 * 32 * 4, device=dev)
    g = torch.zeros(64, device=dev)
    import ctypes as C
    rc = ctx.lib.bh_render_backward(ctx._h, C.c_void_p(z.data_ptr()), C.c_void_p(spl.transforms.data_ptr()), C.c_void_p(spl.sh_coeffs.data_ptr()),
                                    C.c_void_p(spl.raw_opacities.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(g.data_ptr()),
                                    C.c_void_p(g.data_ptr()), C.c_void_p(g.data_ptr()))
    assert rc < 0 and b"forward" in ctx.lib.bh_last_error(ctx._h)


def test_fuzz_bwd_gradients_finite(dev):
    """fuzz.rs:494-560 on the HIP path."""
    import brush_amd as ba
    rng = np.random.default_rng(123)
    for it in range(20):
        n = int(rng.integers(4, 256))
        w, h = int(rng.integers(16, 128)), int(rng.integers(16, 128))
        sc = synth.make_scene(n, 700 + it, sh_degree=int(rng.integers(0, 4)), log_scale_range=(-4.0, 2.0))
        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], render_mip=(it % 3 == 0), device=dev)
        res = ba.render_splats_bwd(spl, util.hip_camera(ba, synth.default_camera_params(w, h)), (w, h), (0, 0, 0),
                                   torch.full((h, w, 4), 1.0 / (h * w * 4), device=dev))
        for k in ("v_transforms", "v_sh_coeffs", "v_raw_opacities", "v_refine_weight"):
            assert bool(torch.isfinite(res[k]).all()), (it, k)
    cam = util.hip_camera(ba, util.STD_CAM)
    for ls_val in (-20.0, 0.0, 15.0, 40.0):
        for mag in (0.1, 1e6, 3.4028235e38 / 2):
            n = 8
            tr = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, ls_val, ls_val, ls_val], np.float32), (n, 1))
            sh = np.tile(np.array([mag, -mag, mag], np.float32), (n, 1, 1))
            spl = ba.Splats(tr, sh, np.full(n, 2.0, np.float32), device=dev)
            res = ba.render_splats_bwd(spl, cam, (64, 64), (0, 0, 0), torch.full((64, 64, 4), 1.0 / (64 * 64 * 4), device=dev))
            assert res["aux"].num_visible == n
            for k in ("v_transforms", "v_sh_coeffs", "v_raw_opacities", "v_refine_weight"):
                assert bool(torch.isfinite(res[k]).all()), (ls_val, mag, k)
