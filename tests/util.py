"""Shared helpers for the parity tests (scenes of the reference's own tests, camera
construction for both the oracle and the HIP host mirror)."""
import math
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def base_scene():
    """crates/brush-bench-test/tests/finite_diff.rs:43-72 (literal constants)."""
    means = np.array([0.20, -0.10, 0.00, -0.30, 0.40, 0.20, 0.10, 0.30, -0.30, -0.20, -0.20, 0.10], np.float32).reshape(4, 3)
    rots = np.array([0.90, 0.10, 0.05, 0.03, 0.70, 0.20, 0.30, 0.10, 0.50, 0.40, 0.30, 0.20, 0.80, 0.10, 0.10, 0.20], np.float32).reshape(4, 4)
    ls = np.array([-1.4, -1.5, -1.6, -1.5, -1.4, -1.3, -1.7, -1.5, -1.4, -1.3, -1.6, -1.5], np.float32).reshape(4, 3)
    sh = np.array([0.45, 0.55, 0.50, 0.60, 0.40, 0.30, 0.35, 0.50, 0.65, 0.50, 0.45, 0.55], np.float32).reshape(4, 1, 3)
    op = np.array([2.5, 2.0, 2.2, 2.4], np.float32)
    return dict(transforms=np.concatenate([means, rots, ls], axis=1), sh=sh, raw_opac=op)


STD_CAM = dict(pos=(0.0, 0.0, -3.0), rot_xyzw=(0.0, 0.0, 0.0, 1.0), fov_x=0.6, fov_y=0.6, center_uv=(0.5, 0.5))  # finite_diff.rs:74-83


def golden_case(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    tr = np.concatenate([d["means"], d["quats"], d["scales"]], axis=1).astype(np.float32)
    return dict(transforms=tr, sh=d["coeffs"].astype(np.float32), raw_opac=d["opacities"].astype(np.float32)), d["out_img"]


def golden_camera_params(w, h):
    """crates/brush-bench-test/src/reference.rs:113-122"""
    fov = math.pi * 0.5
    focal = (w / 2.0) / math.tan(fov / 2.0)
    return dict(pos=(0.123, 0.456, -8.0), rot_xyzw=(0.0, 0.0, 0.0, 1.0), fov_x=2.0 * math.atan((w / 2.0) / focal),
                fov_y=2.0 * math.atan((h / 2.0) / focal), center_uv=(0.5, 0.5), img_w=w, img_h=h)


def oracle_camera(bo, params, w=None, h=None):
    p = dict(params)
    if w is not None:
        p["img_w"], p["img_h"] = w, h
    return bo.camera(**p)


def hip_camera(ba, params):
    return ba.Camera(position=params["pos"], rotation=params["rot_xyzw"], fov_x=params["fov_x"], fov_y=params["fov_y"],
                     center_uv=params["center_uv"])


def quat_from_axis_angle(axis, angle):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    s = math.sin(angle / 2.0)
    return (float(a[0] * s), float(a[1] * s), float(a[2] * s), float(math.cos(angle / 2.0)))


def u32(t):
    """torch int32 tensor -> numpy uint32"""
    return t.detach().cpu().numpy().view(np.uint32)


def rel_linf(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


from oracle.trainer import OracleTrainer  # noqa: E402,F401


def packed_from_rgba(rgba_u8):
    a = rgba_u8.astype(np.uint32)
    return (a[..., 0] | (a[..., 1] << 8) | (a[..., 2] << 16) | (a[..., 3] << 24)).astype(np.uint32)
