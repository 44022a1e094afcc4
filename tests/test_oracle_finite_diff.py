"""The reference has no golden numbers for gradients: its backward
is validated by central finite differences (crates/brush-bench-test/tests/finite_diff.rs).
The same method validates the ORACLE's backward here, with the same scenes,
eps and tolerances, using the C^1 smooth-cutoff pass (finite_diff.rs:25-27)."""
import math

import numpy as np
import pytest

from oracle import bo
import util

FLAGS = bo.FLAG_BWD_INFO | bo.FLAG_SMOOTH_CUTOFF
EPS = 3e-4
REL_TOL = 0.02
ABS_TOL = 2e-4  # finite_diff.rs uses 5e-5..2e-4 abs + 1-2 % rel depending on the case


def _value(scene, cam, weights=None, bg=(0.0, 0.0, 0.0), flags=FLAGS):
    r = bo.Render().forward(cam, scene["transforms"], scene["sh"], scene["raw_opac"], bg=bg, flags=flags)
    img = r.image().astype(np.float64)
    return float(img.mean() if weights is None else (img * weights).sum())


def _analytic(scene, cam, weights=None, bg=(0.0, 0.0, 0.0), flags=FLAGS):
    r = bo.Render().forward(cam, scene["transforms"], scene["sh"], scene["raw_opac"], bg=bg, flags=flags)
    h, w = cam.img_h, cam.img_w
    v = np.full((h, w, 4), 1.0 / (h * w * 4), np.float32) if weights is None else weights.astype(np.float32)
    r.backward(v)
    n = scene["transforms"].shape[0]
    return r.get("v_transforms").reshape(n, 10), r.get("v_coeffs").reshape(n, -1, 3), r.get("v_raw_opac")


def _check(scene, cam, cases, weights=None, bg=(0.0, 0.0, 0.0), eps=EPS, flags=FLAGS):
    vt, vsh, vop = _analytic(scene, cam, weights, bg, flags)
    bad = []
    for kind, i, c in cases:
        def pert(d):
            s = {k: v.copy() for k, v in scene.items()}
            if kind == "tr":
                s["transforms"][i, c] += np.float32(d)
            elif kind == "sh":
                s["sh"][i, c // 3, c % 3] += np.float32(d)
            else:
                s["raw_opac"][i] += np.float32(d)
            return _value(s, cam, weights, bg, flags)
        num = (pert(eps) - pert(-eps)) / (2 * eps)
        an = float(vt[i, c] if kind == "tr" else (vsh[i, c // 3, c % 3] if kind == "sh" else vop[i]))
        tol = ABS_TOL + REL_TOL * max(abs(num), abs(an), 1e-8)
        if abs(num - an) > tol:
            bad.append((kind, i, c, num, an))
    assert not bad, bad


def test_finite_difference_gradient_broad():
    """finite_diff.rs:210-273"""
    cam = bo.camera(img_w=32, img_h=32, **util.STD_CAM)
    cases = [("tr", 0, 0), ("tr", 0, 2), ("tr", 1, 1), ("tr", 0, 3), ("tr", 1, 5), ("tr", 0, 7), ("tr", 1, 8),
             ("sh", 0, 0), ("sh", 1, 1), ("sh", 2, 2), ("op", 0, 0), ("op", 2, 0)]
    _check(util.base_scene(), cam, cases)


def test_finite_diff_all_lanes_weighted_sum():
    """finite_diff.rs:457 (random-weighted sum instead of the mean), every transform lane."""
    cam = bo.camera(img_w=32, img_h=32, **util.STD_CAM)
    rng = np.random.default_rng(3)
    wts = (rng.uniform(-1, 1, (32, 32, 4)) / (32 * 32)).astype(np.float32)
    cases = [("tr", s, c) for s in range(4) for c in range(10)] + [("op", s, 0) for s in range(4)] + [("sh", s, c) for s in range(4) for c in range(3)]
    _check(util.base_scene(), cam, cases, weights=wts, bg=(0.2, 0.4, 0.6))


def test_finite_diff_mip_mode():
    """finite_diff.rs:357"""
    cam = bo.camera(img_w=32, img_h=32, **util.STD_CAM)
    cases = [("tr", 0, 0), ("tr", 1, 7), ("tr", 2, 4), ("op", 1, 0), ("sh", 3, 1)]
    _check(util.base_scene(), cam, cases, flags=FLAGS | bo.FLAG_MIP)


def test_finite_diff_rotated_camera_offcentre_nonsquare():
    """finite_diff.rs:903,950,989"""
    p = dict(util.STD_CAM)
    p["rot_xyzw"] = util.quat_from_axis_angle((0.1, 1.0, 0.2), 0.25)
    p["pos"] = (0.9, 0.1, -2.8)
    p["center_uv"] = (0.42, 0.57)
    p["fov_x"], p["fov_y"] = 0.8, 0.5
    cam = bo.camera(img_w=48, img_h=30, **p)
    cases = [("tr", s, c) for s in range(4) for c in (0, 1, 2, 4, 8)] + [("op", 0, 0)]
    _check(util.base_scene(), cam, cases)


def test_finite_diff_sh_degree3_and_viewdir_path():
    """finite_diff.rs:1253,1306: SH degree 3 incl. the viewdir -> mean path."""
    sc = util.base_scene()
    rng = np.random.default_rng(5)
    sh = np.zeros((4, 16, 3), np.float32)
    sh[:, 0, :] = sc["sh"][:, 0, :]
    sh[:, 1:, :] = rng.uniform(-0.3, 0.3, (4, 15, 3)).astype(np.float32)
    sc["sh"] = sh
    cam = bo.camera(img_w=32, img_h=32, **util.STD_CAM)
    cases = [("tr", 0, 0), ("tr", 1, 1), ("tr", 2, 2), ("sh", 0, 5), ("sh", 1, 20), ("sh", 2, 40), ("sh", 3, 47)]
    _check(sc, cam, cases)


def test_finite_diff_anisotropic_and_near_far():
    """finite_diff.rs:1043,1089,1137"""
    sc = util.base_scene()
    sc["transforms"][:, 7:10] = np.array([[-0.8, -2.2, -1.6], [-2.0, -1.0, -1.4], [-1.2, -1.9, -2.4], [-1.5, -1.5, -0.9]], np.float32)
    sc["transforms"][0, 2] = -1.8   # close to the camera
    sc["transforms"][3, 2] = 6.0    # far
    cam = bo.camera(img_w=40, img_h=40, **util.STD_CAM)
    cases = [("tr", 0, 2), ("tr", 0, 7), ("tr", 3, 0), ("tr", 3, 9), ("tr", 1, 3), ("tr", 2, 6)]
    _check(sc, cam, cases)
