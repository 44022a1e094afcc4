"""Deterministic synthetic scenes for parity tests and bench.py.

SplitMix64 exactly as the reference's test generator
(crates/brush-render/src/tests/mod.rs:168-186) so a scene is reproducible from
(seed, n) in C++/Python/Rust alike. Pure numpy; no GPU, no oracle.
"""
import math

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_unit(seed, count, offset=0):
    """`count` successive next() values in [0,1] as float32 (draws offset.. of the stream)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z.astype(np.float64) / float(2 ** 64 - 1)).astype(np.float32)


def make_scene(n, seed, sh_degree=0, log_scale_range=(math.log(0.005), math.log(0.05)),
               z_range=(2.0, 12.0), tan_half_fov=(math.tan(math.radians(30.0)),) * 2, spread=1.1, opacity_range=(0.05, 0.95)):
    """Random splats filling a pyramid `spread`x the view frustum (tan of the half
    field of view per axis) looking down +Z.

    Returns dict(transforms [n,10] = mean(3) quat wxyz(4) log-scale(3), sh [n,C,3], raw_opac [n]).
    """
    C = (sh_degree + 1) ** 2
    k = 3 + 4 + 3 + 1 + 3 * C
    r = splitmix64_unit(seed, n * k).reshape(n, k)

    def uni(col, lo, hi):
        return (np.float32(lo) + col * np.float32(hi - lo)).astype(np.float32)

    z = uni(r[:, 0], *z_range)
    tx = np.float32(spread * tan_half_fov[0])
    ty = np.float32(spread * tan_half_fov[1])
    x = uni(r[:, 1], -1.0, 1.0) * z * tx
    y = uni(r[:, 2], -1.0, 1.0) * z * ty
    quat = uni(r[:, 3:7], -1.0, 1.0)
    ls = uni(r[:, 7:10], *log_scale_range)
    p = uni(r[:, 10], *opacity_range).astype(np.float64)
    raw_opac = np.log(p / (1.0 - p)).astype(np.float32)
    sh = r[:, 11:].reshape(n, C, 3)
    sh_out = np.empty_like(sh)
    sh_out[:, 0, :] = uni(sh[:, 0, :], -1.0, 1.7)
    if C > 1:
        sh_out[:, 1:, :] = uni(sh[:, 1:, :], -0.25, 0.25)
    transforms = np.concatenate([x[:, None], y[:, None], z[:, None], quat, ls], axis=1).astype(np.float32)
    return {"transforms": np.ascontiguousarray(transforms), "sh": np.ascontiguousarray(sh_out), "raw_opac": raw_opac}


# Named workloads of BASELINE.json / SURVEY.md §8d.
CONFIGS = {
    # configs[0]: plumbing
    "10k_256": dict(n=10_000, w=256, h=256, seed=0xB0, log_scale_range=(math.log(0.02), math.log(0.2))),
    # configs[1]/[2]: 1M splats, 1080p
    "1m_1080p": dict(n=1_000_000, w=1920, h=1080, seed=0xB1, log_scale_range=(math.log(0.005), math.log(0.05))),
    # NOT a BASELINE.json config: the configs[2] scene with opacities U(0.02, 0.1) instead of U(0.05, 0.95), so that a tile's
    # pixels do not saturate after the first tenth of its list — the blend kernels then consume most of every list
    # (bench.py's second, non-headline measurement: their throughput on a scene that does not flatter them)
    "1m_1080p_lowopac": dict(n=1_000_000, w=1920, h=1080, seed=0xB1, log_scale_range=(math.log(0.005), math.log(0.05)), opacity_range=(0.02, 0.1)),
    # NOT a BASELINE.json config: an object-centric frame — the configs[2] splats squeezed into the central half of the frustum, so
    # that the outer tiles are empty (they never saturate: what a NeRF-synthetic view with a blank background looks like to the
    # list builder)
    "1m_1080p_centered": dict(n=1_000_000, w=1920, h=1080, seed=0xB1, log_scale_range=(math.log(0.005), math.log(0.05)), spread=0.5),
    # the SURVEY 8d "heavy" variant: scales U(ln 0.01, ln 0.1), I ~ 37 M
    "1m_1080p_heavy": dict(n=1_000_000, w=1920, h=1080, seed=0xB1, log_scale_range=(math.log(0.01), math.log(0.1))),
    # configs[4]: 6M splats, 4K
    "6m_4k": dict(n=6_000_000, w=3840, h=2160, seed=0xB5, log_scale_range=(math.log(0.005), math.log(0.05))),
}


def config_scene(name, sh_degree=0, n=None):
    cfg = CONFIGS[name]
    cam = default_camera_params(cfg["w"], cfg["h"])
    tans = (math.tan(cam["fov_x"] / 2.0), math.tan(cam["fov_y"] / 2.0))
    scene = make_scene(n or cfg["n"], cfg["seed"], sh_degree=sh_degree, log_scale_range=cfg["log_scale_range"],
                       tan_half_fov=tans, opacity_range=cfg.get("opacity_range", (0.05, 0.95)), spread=cfg.get("spread", 1.1))
    return scene, cfg["w"], cfg["h"]


def default_camera_params(w, h, fov_x_deg=60.0):
    """Origin, identity rotation (+Z forward), pinhole, square pixels."""
    fov_x = math.radians(fov_x_deg)
    fx = (w / 2.0) / math.tan(fov_x / 2.0)
    fov_y = 2.0 * math.atan((h / 2.0) / fx)
    return dict(pos=(0.0, 0.0, 0.0), rot_xyzw=(0.0, 0.0, 0.0, 1.0), fov_x=fov_x, fov_y=fov_y,
                center_uv=(0.5, 0.5), img_w=w, img_h=h)


def synthetic_gt_packed(w, h, seed=7):
    """Smooth RGBA8 pattern packed as u32 [h,w] (r in bits 0-7 ... a in 24-31), opaque alpha."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    r = 0.5 + 0.5 * np.sin(xx * 0.013 + seed)
    g = 0.5 + 0.5 * np.cos(yy * 0.017 + 0.3 * seed)
    b = 0.5 + 0.5 * np.sin((xx + yy) * 0.007)
    def q(v):
        return np.clip(v * 255.0, 0, 255).astype(np.uint32)
    return (q(r) | (q(g) << 8) | (q(b) << 16) | (np.uint32(255) << 24)).astype(np.uint32)
