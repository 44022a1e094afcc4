"""Camera models: Kannala-Brandt 4, radial-tangential 8, thin-prism fisheye.

The reference pins these with (a) fov<->focal round trips and finite renders
(crates/brush-render/src/tests/mod.rs:711-871) and (b) finite-difference fuzzing of the backward
(crates/brush-bench-test/tests/finite_diff.rs:723-800, 1170-1240) — there are no golden images
for the distorted lenses.  The same checks validate the ORACLE restatement here (the HIP kernels
are then compared with the oracle in tests/test_gpu_camera_models.py), plus two independent pins:
a float64 numpy restatement of the three projection laws, and the analytic projection Jacobian
against central differences of the projection."""
import math

import numpy as np
import pytest

from oracle import bo
import util

FLAGS = bo.FLAG_BWD_INFO | bo.FLAG_SMOOTH_CUTOFF
EPS = 3e-4  # finite_diff.rs:821


# ---- fov <-> focal (tests/mod.rs:711-790) ------------------------------------------------------
def test_pinhole_focal_to_fov_and_back():
    fov = bo.focal_to_fov(800.0, 1920)
    assert abs(bo.fov_to_focal(fov, 1920) - 800.0) < 1e-9


def test_kb4_focal_to_fov_and_back_no_distortion():
    fov = bo.focal_to_fov(300.0, 1024, "kb4", (0, 0, 0, 0))
    assert abs(fov - 1024 / 300.0) < 1e-9  # zero-distortion KB4: r_pix = f * theta
    assert abs(bo.fov_to_focal(fov, 1024, "kb4", (0, 0, 0, 0)) - 300.0) < 1e-9


@pytest.mark.parametrize("model,dist,f,pixels", [
    ("kb4", (-0.01, 0.003, -0.0005, 0.00002), 280.0, 1024),
    ("rt8", (-0.2, 0.05, -0.001, 0.0, 0.0, 0.0, 0.0, 0.0), 900.0, 1920),
    ("tpf", (-0.01, 0.003, -0.0005, 0.00002, 1e-3, -2e-3, 5e-4, -5e-4), 280.0, 1024),
])
def test_distorted_focal_to_fov_and_back(model, dist, f, pixels):
    fov = bo.focal_to_fov(f, pixels, model, dist)
    assert abs(bo.fov_to_focal(fov, pixels, model, dist) - f) < 1e-6


def test_rt8_clamp_limits_collapse_to_pinhole_for_a_tiny_distortion():
    """camera.rs:228-245: with a near-pinhole lens the undistorted bound equals the pinhole one."""
    p = dict(util.STD_CAM)
    a = bo.camera(img_w=64, img_h=48, **p)
    b = bo.camera(img_w=64, img_h=48, model="rt8", dist=(1e-9, 0, 0, 0, 0, 0, 0, 0), **p)
    for f in ("lim_pos_x", "lim_pos_y", "lim_neg_x", "lim_neg_y"):
        assert abs(getattr(a, f) - getattr(b, f)) < 1e-5
    c = bo.camera(img_w=64, img_h=48, model="kb4", dist=(0.01, 0, 0, 0), **p)
    assert (c.lim_pos_x, c.lim_pos_y, c.lim_neg_x, c.lim_neg_y) == (0.0, 0.0, 0.0, 0.0)  # fisheye: unclamped
    assert abs(a.half_max_render_fov - np.float32(np.hypot(np.float32(0.6), np.float32(0.6)) * np.float32(1.05)) * 0.5) < 1e-6


# ---- the projection laws against an independent float64 restatement ----------------------------
def _project_f64(model, dist, fx, fy, cx, cy, p):
    x, y, z = (float(v) for v in p)
    if model == "pinhole":
        return fx * x / z + cx, fy * y / z + cy
    if model == "rt8":
        k1, k2, k3, k4, k5, k6, p1, p2 = dist
        a, b = x / z, y / z
        r2 = a * a + b * b
        d = (1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3) / (1 + k4 * r2 + k5 * r2 ** 2 + k6 * r2 ** 3)
        return (fx * (a * d + 2 * p1 * a * b + p2 * (r2 + 2 * a * a)) + cx,
                fy * (b * d + 2 * p2 * a * b + p1 * (r2 + 2 * b * b)) + cy)
    k1, k2, k3, k4 = dist[:4]
    r = math.hypot(x, y)
    th = math.atan2(r, z)
    d = th * (1 + k1 * th ** 2 + k2 * th ** 4 + k3 * th ** 6 + k4 * th ** 8)
    u, v = fx * d * x / r + cx, fy * d * y / r + cy
    if model == "tpf":
        p1, p2, sx1, sy1 = dist[4:8]
        r2 = x * x + y * y
        u += fx * (2 * p1 * x * y + p2 * (3 * x * x + y * y) + sx1 * r2) / (z * z)
        v += fy * (2 * p2 * x * y + p1 * (x * x + 3 * y * y) + sy1 * r2) / (z * z)
    return u, v


@pytest.mark.parametrize("lens", ["kb4", "rt8", "tpf"])
def test_projected_means_match_float64_restatement(lens):
    """One tiny isotropic splat per probe point: its projected record's xy is the lens law."""
    model, dist = util.REF_LENSES[lens]
    cam = bo.camera(img_w=96, img_h=80, model=model, dist=dist, pos=(0.0, 0.0, 0.0), rot_xyzw=(0, 0, 0, 1), fov_x=0.9, fov_y=0.8,
                    center_uv=(0.48, 0.53))
    rng = np.random.default_rng(11)
    n = 64
    pts = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(-0.7, 0.7, n), rng.uniform(1.5, 4.0, n)], 1).astype(np.float32)
    tr = np.concatenate([pts, np.tile(np.array([[1, 0, 0, 0]], np.float32), (n, 1)), np.full((n, 3), -4.0, np.float32)], 1)
    r = bo.Render().forward(cam, tr, np.full((n, 1, 3), 0.5, np.float32), np.full(n, 3.0, np.float32))
    assert r.num_visible > n // 2
    proj = r.get("projected").reshape(-1, 9)
    gids = r.get("global_from_compact_gid")
    for row, gid in zip(proj, gids):
        eu, ev = _project_f64(model, dist, cam.fx, cam.fy, cam.cx, cam.cy, pts[gid])
        assert abs(row[0] - eu) < 2e-3 and abs(row[1] - ev) < 2e-3, (gid, row[:2], eu, ev)


def test_atan2_polynomial_accuracy():
    """The fixed atan2 polynomial shared by oracle and HIP kernels is within 2 ulp-ish of libm on the
    domain the fisheye laws use (r >= 0, any z): probed through theta = acos of a KB4 zero-distortion
    projection, u = fx * theta * x / r."""
    cam = bo.camera(img_w=64, img_h=64, model="kb4", dist=(0, 0, 0, 0), pos=(0, 0, 0), rot_xyzw=(0, 0, 0, 1), fov_x=2.6, fov_y=2.6)
    rng = np.random.default_rng(5)
    n = 512
    pts = np.stack([rng.uniform(-3, 3, n), np.zeros(n), rng.uniform(0.2, 3.0, n)], 1).astype(np.float32)
    pts[np.abs(pts[:, 0]) < 1e-3, 0] = 0.5
    tr = np.concatenate([pts, np.tile(np.array([[1, 0, 0, 0]], np.float32), (n, 1)), np.full((n, 3), -5.0, np.float32)], 1)
    r = bo.Render().forward(cam, tr, np.full((n, 1, 3), 0.5, np.float32), np.full(n, 3.0, np.float32))
    proj = r.get("projected").reshape(-1, 9)
    gids = r.get("global_from_compact_gid")
    assert len(gids) > 200
    th = (proj[:, 0].astype(np.float64) - cam.cx) / cam.fx * np.sign(pts[gids, 0])
    ref = np.arctan2(np.abs(pts[gids, 0].astype(np.float64)), pts[gids, 2].astype(np.float64))
    assert np.abs(th - ref).max() < 5e-7


# ---- analytic projection Jacobian vs central differences of the projection ----------------------
@pytest.mark.parametrize("lens", ["kb4", "rt8", "tpf"])
def test_cov2d_jacobian_consistent_with_projection(lens):
    """The 2x2 screen covariance of a tiny isotropic splat is s^2 J J^T: compare the record's conic
    (inverse of cov + 0.3 I) with J from central differences of the float64 lens law."""
    model, dist = util.REF_LENSES[lens]
    cam = bo.camera(img_w=96, img_h=96, model=model, dist=dist, pos=(0, 0, 0), rot_xyzw=(0, 0, 0, 1), fov_x=0.9, fov_y=0.9)
    rng = np.random.default_rng(2)
    n = 32
    pts = np.stack([rng.uniform(-0.6, 0.6, n), rng.uniform(-0.6, 0.6, n), rng.uniform(1.5, 3.0, n)], 1).astype(np.float32)
    s = 0.05
    tr = np.concatenate([pts, np.tile(np.array([[1, 0, 0, 0]], np.float32), (n, 1)), np.full((n, 3), math.log(s), np.float32)], 1)
    r = bo.Render().forward(cam, tr, np.full((n, 1, 3), 0.5, np.float32), np.full(n, 3.0, np.float32))
    proj = r.get("projected").reshape(-1, 9)
    gids = r.get("global_from_compact_gid")
    assert len(gids) >= n // 2
    h = 1e-5
    for row, gid in zip(proj, gids):
        p = pts[gid].astype(np.float64)
        J = np.zeros((2, 3))
        for k in range(3):
            dp = np.zeros(3)
            dp[k] = h
            a = _project_f64(model, dist, cam.fx, cam.fy, cam.cx, cam.cy, p + dp)
            b = _project_f64(model, dist, cam.fx, cam.fy, cam.cx, cam.cy, p - dp)
            J[:, k] = [(a[0] - b[0]) / (2 * h), (a[1] - b[1]) / (2 * h)]
        cov = s * s * J @ J.T + 0.3 * np.eye(2)
        conic = np.linalg.inv(cov)
        got = np.array([[row[2], row[3]], [row[3], row[4]]], np.float64)
        assert np.abs(got - conic).max() <= 2e-3 * np.abs(conic).max(), (gid, got, conic)


# ---- renders are finite (tests/mod.rs:792-871) ---------------------------------------------------
@pytest.mark.parametrize("lens", ["kb4", "rt8", "tpf"])
def test_renders_finite_with_model(lens):
    model, dist = util.REF_LENSES[lens]
    cam = bo.camera(img_w=48, img_h=48, model=model, dist=dist, pos=(0.0, 0.0, -3.0), rot_xyzw=(0, 0, 0, 1), fov_x=0.7, fov_y=0.7)
    rng = np.random.default_rng(9)
    n = 64
    tr = np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(-1, 1, (n, 4)), rng.uniform(-3.0, -1.5, (n, 3))], 1).astype(np.float32)
    r = bo.Render().forward(cam, tr, rng.uniform(0, 1, (n, 1, 3)).astype(np.float32), rng.uniform(1, 3, n).astype(np.float32))
    img = r.image()
    assert np.isfinite(img).all() and img[..., 3].max() > 0.1 and r.num_visible > n // 2


def test_fisheye_culls_on_view_angle_not_depth():
    """project_forward.rs:47-61: pinhole culls z < 0.01; the other models cull theta > half_max_render_fov."""
    p = dict(pos=(0, 0, 0), rot_xyzw=(0, 0, 0, 1), fov_x=1.0, fov_y=1.0)
    tr = np.array([[0.0, 0.0, 2.0, 1, 0, 0, 0, -3, -3, -3],      # on axis
                   [1.9, 0.0, 1.0, 1, 0, 0, 0, -3, -3, -3],      # theta = 1.086 > 0.7425 -> culled by angle
                   [0.0, 0.0, -2.0, 1, 0, 0, 0, -3, -3, -3]], np.float32)  # behind
    sh, op = np.full((3, 1, 3), 0.5, np.float32), np.full(3, 3.0, np.float32)
    cam = bo.camera(img_w=64, img_h=64, model="kb4", dist=(0, 0, 0, 0), **p)
    r = bo.Render().forward(cam, tr, sh, op)
    assert list(r.get("global_from_compact_gid")) == [0]
    assert abs(cam.half_max_render_fov - 0.5 * 1.05 * math.hypot(1.0, 1.0)) < 1e-6


# ---- backward: finite differences (finite_diff.rs:723-800 and :1170-1240) -------------------------
def _value(scene, cam):
    r = bo.Render().forward(cam, scene["transforms"], scene["sh"], scene["raw_opac"], flags=FLAGS)
    return float(r.image().astype(np.float64).mean())


def _analytic(scene, cam):
    r = bo.Render().forward(cam, scene["transforms"], scene["sh"], scene["raw_opac"], flags=FLAGS)
    h, w = cam.img_h, cam.img_w
    r.backward(np.full((h, w, 4), 1.0 / (h * w * 4), np.float32))
    n = scene["transforms"].shape[0]
    return r.get("v_transforms").reshape(n, 10), r.get("v_coeffs").reshape(n, -1, 3), r.get("v_raw_opac")


def _fd_rows(scene, cam, lanes):
    vt, vsh, vop = _analytic(scene, cam)
    rows = []
    for kind, i, c in lanes:
        def pert(d):
            s = {k: v.copy() for k, v in scene.items()}
            if kind == "tr":
                s["transforms"][i, c] += np.float32(d)
            elif kind == "sh":
                s["sh"][i, 0, c] += np.float32(d)
            else:
                s["raw_opac"][i] += np.float32(d)
            return _value(s, cam)
        num = (pert(EPS) - pert(-EPS)) / (2 * EPS)
        an = float(vt[i, c] if kind == "tr" else (vsh[i, 0, c] if kind == "sh" else vop[i]))
        rows.append((kind, i, c, num, an))
    return rows


def _assert_clean(rows, rel_tol, abs_tol, label):
    bad = [r for r in rows if abs(r[3] - r[4]) > abs_tol + rel_tol * max(abs(r[3]), abs(r[4]), 1e-8)]
    assert not bad, (label, bad)


def test_fuzz_finite_diff_camera_models():
    """finite_diff.rs:779-794: 20 seeds, random lens per seed, 3..8 splats, 32x32; here EVERY mean lane
    plus one random lane of the other groups per seed (the reference probes one random lane)."""
    seen = set()
    for seed in range(20):
        cp = util.random_camera_with_model(seed)
        seen.add(cp["model"])
        rng = util.Sm64((seed + 0xC0DEBEEF) & 0xFFFFFFFFFFFFFFFF)
        n = rng.usize_in(3, 9)
        scene = util.random_scene(seed, n)
        cam = bo.camera(img_w=32, img_h=32, **cp)
        pick = util.Sm64(seed * 0xA5A55A5A + 0x1234)
        s = pick.usize_in(0, n)
        lanes = [("tr", s, 0), ("tr", s, 1), ("tr", s, 2), ("tr", pick.usize_in(0, n), 3 + pick.usize_in(0, 4)),
                 ("tr", pick.usize_in(0, n), 7 + pick.usize_in(0, 3)), ("sh", pick.usize_in(0, n), pick.usize_in(0, 3)), ("op", pick.usize_in(0, n), 0)]
        _assert_clean(_fd_rows(scene, cam, lanes), 0.02, 2e-4, "cam-models seed %d %s" % (seed, cp["model"]))
    assert seen == {"pinhole", "kb4", "rt8", "tpf"}


def test_fuzz_heavy_distortion():
    """finite_diff.rs:1170-1240: strong KB4 / RT8 / thin-prism distortion, 48x48, tolerance 3 % + 5e-4."""
    for seed in range(20):
        cp = util.heavy_distortion_camera(seed)
        rng = util.Sm64((seed * 0xB1B1) & 0xFFFFFFFFFFFFFFFF)
        n = rng.usize_in(3, 8)
        scene = util.random_scene(seed, n)
        for i in range(n):
            scene["transforms"][i, 0] = rng.uniform(-0.5, 0.5)
            scene["transforms"][i, 1] = rng.uniform(-0.5, 0.5)
            scene["transforms"][i, 2] = rng.uniform(-0.5, 0.5)
        cam = bo.camera(img_w=48, img_h=48, **cp)
        pick = util.Sm64(seed * 0xA5A55A5A + 0x1234)
        s = pick.usize_in(0, n)
        lanes = [("tr", s, 0), ("tr", s, 1), ("tr", s, 2), ("tr", pick.usize_in(0, n), 7 + pick.usize_in(0, 3)), ("tr", pick.usize_in(0, n), 3 + pick.usize_in(0, 4))]
        _assert_clean(_fd_rows(scene, cam, lanes), 0.03, 5e-4, "heavy-dist seed %d %s" % (seed, cp["model"]))


def test_rt8_clamped_jacobian_region():
    """A splat whose x/z lies beyond the RT8 Jacobian clamp (but inside the view-angle gate).  The
    reference's VJP there differentiates the CLAMPED surrogate point (radial_tangential_8.rs:196-260:
    J_eff zeroes the x column), while its forward projects the unclamped point (:23-67) — so the mean
    gradient is by construction not the finite difference of the forward in this region (the reference's
    own fuzz keeps splats out of it, finite_diff.rs:1105-1110).  What must hold: the restatement
    reproduces that routing (d/dx through the mean2d path is exactly 0), everything is finite, and the
    lanes that do not pass through the projection VJP (scales, opacity) still match finite differences."""
    model, dist = util.REF_LENSES["rt8"]
    cam = bo.camera(img_w=48, img_h=48, model=model, dist=dist, pos=(0.0, 0.0, -3.0), rot_xyzw=(0, 0, 0, 1), fov_x=0.7, fov_y=0.7)
    assert cam.lim_pos_x > 0 and cam.lim_neg_x < 0
    x_off = 3.0 * (cam.lim_pos_x + 0.05)   # x/z beyond the positive clamp, inside the view-angle gate
    tr = np.array([[x_off, 0.1, 0.0, 0.9, 0.1, 0.05, 0.03, -0.2, -0.6, -0.9],
                   [0.1, -0.2, 0.2, 0.7, 0.2, 0.3, 0.1, -1.4, -1.5, -1.6]], np.float32)
    scene = dict(transforms=tr, sh=np.array([[[0.4, 0.5, 0.6]], [[0.6, 0.4, 0.3]]], np.float32), raw_opac=np.array([2.0, 2.5], np.float32))
    r = bo.Render().forward(cam, scene["transforms"], scene["sh"], scene["raw_opac"], flags=FLAGS)
    assert r.num_visible == 2
    vt, _, _ = _analytic(scene, cam)
    assert np.isfinite(vt).all() and vt[0, 0] == 0.0 and vt[0, 2] != 0.0
    lanes = [("tr", 0, 7), ("tr", 0, 8), ("tr", 0, 9), ("op", 0, 0), ("tr", 1, 0), ("tr", 1, 2)]
    _assert_clean(_fd_rows(scene, cam, lanes), 0.03, 5e-4, "rt8-clamped")
