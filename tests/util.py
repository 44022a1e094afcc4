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
                     center_uv=params["center_uv"], camera_model=params.get("model", "pinhole"), dist=tuple(params.get("dist", ())))


# Lens parameters of the reference's own camera-model tests
# (crates/brush-render/src/tests/mod.rs:735-871): name -> (model, dist in struct order)
REF_LENSES = {
    "kb4": ("kb4", (-0.05, 0.01, -0.001, 5e-5)),
    "rt8": ("rt8", (-0.2, 0.05, -0.001, 0.0, 0.0, 0.0, 1e-3, -1e-3)),
    "tpf": ("tpf", (-0.05, 0.01, -0.001, 5e-5, 1e-3, -1e-3, 5e-4, -5e-4)),
}


class Sm64:
    """SplitMix64 as crates/brush-bench-test/tests/finite_diff.rs (Sm64) / brush-render tests use it."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def f01(self):  # finite_diff.rs:566-568
        return np.float32(float(self.next_u64()) / float(0xFFFFFFFFFFFFFFFF))

    def uniform(self, lo, hi):  # f32 arithmetic, finite_diff.rs:569-571
        return float(np.float32(lo) + self.f01() * (np.float32(hi) - np.float32(lo)))

    def usize_in(self, lo, hi):
        return lo + int(self.next_u64() % (hi - lo))


def random_scene(seed, n):
    """crates/brush-bench-test/tests/finite_diff.rs:577-590 (same SplitMix64 stream)."""
    rng = Sm64((seed * 0x517CC1B727220A95) & 0xFFFFFFFFFFFFFFFF)
    means = np.array([rng.uniform(-1.0, 1.0) for _ in range(n * 3)], np.float32).reshape(n, 3)
    rots = np.array([rng.uniform(-1.0, 1.0) for _ in range(n * 4)], np.float32).reshape(n, 4)
    ls = np.array([rng.uniform(-2.5, 0.0) for _ in range(n * 3)], np.float32).reshape(n, 3)
    sh = np.array([rng.uniform(0.2, 0.8) for _ in range(n * 3)], np.float32).reshape(n, 1, 3)
    op = np.array([rng.uniform(0.5, 3.0) for _ in range(n)], np.float32)
    return dict(transforms=np.concatenate([means, rots, ls], axis=1), sh=sh, raw_opac=op)


def random_camera_with_model(seed):
    """finite_diff.rs:730-776: random pose / fov and a randomly chosen lens with mild distortion."""
    rng = Sm64(((seed * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF) ^ 0xC0DE)
    dist = rng.uniform(2.5, 5.0)
    pos = (rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), -dist)
    fov = rng.uniform(0.5, 1.0)
    which = rng.usize_in(0, 4)
    if which == 0:
        model, d = "pinhole", ()
    elif which == 1:
        model, d = "kb4", (rng.uniform(-0.05, 0.05), rng.uniform(-0.02, 0.02), rng.uniform(-0.01, 0.01), rng.uniform(-0.005, 0.005))
    elif which == 2:
        k1, k2, k3 = rng.uniform(-0.1, 0.1), rng.uniform(-0.05, 0.05), rng.uniform(-0.01, 0.01)
        p1, p2 = rng.uniform(-0.01, 0.01), rng.uniform(-0.01, 0.01)
        model, d = "rt8", (k1, k2, k3, 0.0, 0.0, 0.0, p1, p2)
    else:
        kb = (rng.uniform(-0.05, 0.05), rng.uniform(-0.02, 0.02), rng.uniform(-0.005, 0.005), rng.uniform(-0.001, 0.001))
        model, d = "tpf", kb + (rng.uniform(-0.01, 0.01), rng.uniform(-0.01, 0.01), rng.uniform(-0.005, 0.005), rng.uniform(-0.005, 0.005))
    return dict(pos=pos, rot_xyzw=(0.0, 0.0, 0.0, 1.0), fov_x=fov, fov_y=fov, center_uv=(0.5, 0.5), model=model, dist=d)


def heavy_distortion_camera(seed):
    """finite_diff.rs:1180-1225: strong distortion, KB4 / RT8 / thin-prism fisheye cycled per seed."""
    rng = Sm64((seed * 0xF15EBEEF) & 0xFFFFFFFFFFFFFFFF)
    dist = rng.uniform(3.0, 5.0)
    fov = rng.uniform(0.5, 0.9)
    m = seed % 3
    if m == 0:
        model, d = "kb4", (rng.uniform(-0.3, 0.3), rng.uniform(-0.15, 0.15), rng.uniform(-0.05, 0.05), rng.uniform(-0.02, 0.02))
    elif m == 1:
        k = [rng.uniform(-0.4, 0.4), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05), rng.uniform(-0.01, 0.01), rng.uniform(-0.005, 0.005), 0.0]
        model, d = "rt8", tuple(k) + (rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05))
    else:
        kb = (rng.uniform(-0.2, 0.2), rng.uniform(-0.1, 0.1), rng.uniform(-0.02, 0.02), rng.uniform(-0.005, 0.005))
        model, d = "tpf", kb + (rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03), rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02))
    return dict(pos=(0.0, 0.0, -dist), rot_xyzw=(0.0, 0.0, 0.0, 1.0), fov_x=fov, fov_y=fov, center_uv=(0.5, 0.5), model=model, dist=d)


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


def assert_adam_close(a, b, lr, steps=1, what="", extra_abs=0.0):
    """Parameters after `steps` Adam steps computed two ways (HIP vs oracle, 2 ranks vs 1, ...).  Adam's update is
    lr * m / (sqrt(v) + 1e-15): where a gradient is summation-order noise (|g| ~ 1e-14) its sign, hence a full +-lr
    step, is arbitrary; everywhere else the two must agree to a small fraction of the learning rate.  So: at most
    0.1 % of the entries may differ by more than 2 % of the accumulated lr, none by more than the +-lr flips allow."""
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    tol = 0.02 * lr * steps + extra_abs
    frac = float(np.mean(d > tol))
    assert frac <= 1e-3, (what, "fraction beyond 2%% of lr: %.2e" % frac, float(d.max()))
    assert float(d.max()) <= 2.1 * lr * steps + extra_abs, (what, float(d.max()))


def random_camera(seed):
    """crates/brush-bench-test/tests/finite_diff.rs:592-606 (pinhole, identity rotation, same SplitMix64 stream)."""
    rng = Sm64(((seed * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF) ^ 0xCAFE)
    dist = rng.uniform(2.5, 5.0)
    pos = (rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), -dist)
    fov = rng.uniform(0.4, 0.9)
    return dict(pos=pos, rot_xyzw=(0.0, 0.0, 0.0, 1.0), fov_x=fov, fov_y=fov, center_uv=(0.5, 0.5))
