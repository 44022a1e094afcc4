"""Invariants the reference tests as properties (crates/brush-render/src/tests/mod.rs,
crates/brush-bench-test/tests/fuzz.rs), checked on the oracle."""
import math

import numpy as np
import pytest

from brush_amd import synth
from oracle import bo
import util

POISON = [float("nan"), -float("nan"), float("inf"), -float("inf"), 0.0, -0.0, 1.17549435e-38, 5.877e-39, 1e-40,
          1.1920929e-07, 1e38, -1e38, 3.4028235e38, -3.4028235e38, 1e20, -1e20, 1.0, -1.0, 0.01, 1e10, 1.0 / 255.0, 16.0]  # fuzz.rs:61-86


def small_scene(n=300, seed=11, sh_degree=0):
    return synth.make_scene(n, seed, sh_degree=sh_degree, log_scale_range=(math.log(0.03), math.log(0.3)))


def test_empty_render():
    """tests/mod.rs:20"""
    cam = bo.camera(img_w=33, img_h=17, **util.STD_CAM)
    r = bo.Render().forward(cam, np.zeros((0, 10), np.float32), np.zeros((0, 1, 3), np.float32), np.zeros((0,), np.float32), bg=(0.25, 0.5, 0.75))
    assert r.num_visible == 0 and r.num_intersections == 0
    img = r.image()
    assert np.allclose(img[..., :3], [0.25, 0.5, 0.75]) and np.all(img[..., 3] == 0)


def test_counts_invariants_and_tile_lists():
    """render_aux.rs:30-45; map/PF agreement tests/mod.rs:454,519"""
    sc = small_scene(2000)
    cam = bo.camera(**synth.default_camera_params(200, 120))
    r = bo.Render().forward(cam, sc["transforms"], sc["sh"], sc["raw_opac"])
    assert r.num_visible <= 2000 and r.num_intersections <= r.num_visible * r.num_tiles
    cum = r.get("cum_tiles_hit")
    assert cum[-1] == r.num_intersections
    tiles = r.get("tile_id_from_isect")
    assert tiles.max() < r.num_tiles, "no sentinel rows: count and emit walks agree"
    assert np.all(np.diff(tiles.astype(np.int64)) >= 0)
    offs = r.get("tile_offsets_pre").reshape(-1, 2)
    gids = r.get("compact_gid_from_isect")
    for t in range(r.num_tiles):
        lo, hi = offs[t]
        assert np.all(tiles[lo:hi] == t)
        assert np.all(np.diff(gids[lo:hi].astype(np.int64)) > 0), "depth order inside a tile"
    d = r.get("depths_sorted")
    assert np.all(np.diff(d) >= 0)


def test_culled_splats_do_not_perturb():
    """tests/mod.rs:315,360"""
    sc = small_scene(500)
    cam = bo.camera(**synth.default_camera_params(96, 64))
    base = bo.Render().forward(cam, sc["transforms"], sc["sh"], sc["raw_opac"]).image()
    extra = small_scene(200, seed=99)
    extra["transforms"][:, 2] = -5.0  # behind the camera
    tr = np.concatenate([sc["transforms"], extra["transforms"]])
    sh = np.concatenate([sc["sh"], extra["sh"]])
    op = np.concatenate([sc["raw_opac"], extra["raw_opac"]])
    img = bo.Render().forward(cam, tr, sh, op).image()
    assert np.array_equal(img, base)


def test_zero_quat_and_nan_are_culled():
    """tests/mod.rs:676; fuzz.rs:332"""
    sc = small_scene(50)
    cam = bo.camera(**synth.default_camera_params(64, 64))
    for col, val in ((3, 0.0), (0, float("nan")), (7, float("inf")), (2, 1e11)):
        tr = sc["transforms"].copy()
        if col == 3:
            tr[:, 3:7] = 0.0
        else:
            tr[:, col] = val
        r = bo.Render().forward(cam, tr, sc["sh"], sc["raw_opac"])
        assert r.num_visible == 0 and r.num_intersections == 0
    op = np.full(50, float("nan"), np.float32)
    assert bo.Render().forward(cam, sc["transforms"], sc["sh"], op).num_visible == 0


def test_fuzz_poisoned_scenes_keep_invariants():
    """fuzz.rs:229-330: poisoned inputs (forward only) keep the count invariants and never
    leak NaN/Inf into the image."""
    rng = np.random.default_rng(2024)
    sizes = [(1, 1), (16, 16), (17, 31), (64, 48), (257, 257)]
    for it in range(40):
        n = int(rng.integers(1, 80))
        sc = small_scene(n, seed=1000 + it, sh_degree=int(rng.integers(0, 3)))
        for arr in (sc["transforms"], sc["sh"], sc["raw_opac"]):
            flat = arr.reshape(-1)
            k = max(1, int(0.05 * flat.size))
            idx = rng.integers(0, flat.size, k)
            flat[idx] = np.array(POISON, np.float32)[rng.integers(0, len(POISON), k)]
        w, h = sizes[it % len(sizes)]
        cam = bo.camera(**synth.default_camera_params(w, h))
        r = bo.Render().forward(cam, sc["transforms"], sc["sh"], sc["raw_opac"], bg=(0.1, 0.1, 0.1))
        assert r.num_visible <= n and r.num_intersections <= max(r.num_visible, 1) * r.num_tiles
        assert np.isfinite(r.image()).all()


def test_fuzz_bwd_finite_scenes_and_extreme_inputs_have_finite_grads():
    """fuzz.rs:494-560: backward gradients stay finite on finite random scenes (both render
    modes) and on extreme-but-valid log-scales / colours."""
    rng = np.random.default_rng(77)
    for it in range(25):
        n = int(rng.integers(4, 200))
        w, h = int(rng.integers(16, 128)), int(rng.integers(16, 128))
        sc = small_scene(n, seed=500 + it, sh_degree=int(rng.integers(0, 4)))
        sc["transforms"][:, 7:10] = rng.uniform(-4.0, 2.0, (n, 3)).astype(np.float32)
        cam = bo.camera(**synth.default_camera_params(w, h))
        flags = bo.FLAG_BWD_INFO | (bo.FLAG_MIP if it % 3 == 0 else 0)
        r = bo.Render().forward(cam, sc["transforms"], sc["sh"], sc["raw_opac"], flags=flags)
        r.backward(np.full((h, w, 4), 1.0 / (h * w * 4), np.float32))
        for name in ("v_transforms", "v_coeffs", "v_raw_opac", "v_refine"):
            assert np.isfinite(r.get(name)).all(), (it, name)
    cam = bo.camera(img_w=64, img_h=64, **util.STD_CAM)
    for ls_val in (-20.0, -5.0, 0.0, 5.0, 15.0, 30.0, 40.0):
        for mag in (0.1, 10.0, 1e6, 3.4028235e38 / 2):
            n = 8
            tr = np.tile(np.array([0, 0, 3.0, 1, 0, 0, 0, ls_val, ls_val, ls_val], np.float32), (n, 1))
            sh = np.tile(np.array([mag, -mag, mag], np.float32), (n, 1, 1))
            op = np.full(n, 2.0, np.float32)
            r = bo.Render().forward(cam, tr, sh, op)
            assert r.num_visible == n, "log_scale=%g over-culled" % ls_val  # fuzz.rs:452-480
            r.backward(np.full((64, 64, 4), 1.0 / (64 * 64 * 4), np.float32))
            for name in ("v_transforms", "v_coeffs", "v_raw_opac", "v_refine"):
                assert np.isfinite(r.get(name)).all(), (ls_val, mag, name)


def test_loss_structural():
    """brush-loss/tests/reference.rs:57,81,105: SSIM(x,x) = 1, range, finite non-zero backward."""
    rng = np.random.default_rng(0)
    h, w = 40, 52
    gt8 = rng.integers(0, 256, (h, w, 4), dtype=np.uint32)
    packed = gt8[..., 0] | (gt8[..., 1] << 8) | (gt8[..., 2] << 16) | (gt8[..., 3] << 24)
    same = (gt8[..., :3].astype(np.float32) / 255.0).transpose(2, 0, 1).copy()
    lm = bo.image_loss_forward(same, packed, 0.0, 1.0)
    assert np.allclose(lm, 1.0, atol=1e-5)
    pred = rng.uniform(0, 1, (3, h, w)).astype(np.float32)
    lm = bo.image_loss_forward(pred, packed, 0.0, 1.0)
    assert lm.min() >= -1.0 and lm.max() <= 1.0
    g = bo.image_loss_backward(pred, packed, np.full((3, h, w), 1.0 / (3 * h * w), np.float32), 0.8, -0.2)
    assert np.isfinite(g).all() and np.abs(g).max() > 0


def test_loss_backward_matches_finite_difference():
    rng = np.random.default_rng(1)
    h, w = 24, 28
    gt8 = rng.integers(0, 256, (h, w, 4), dtype=np.uint32)
    packed = gt8[..., 0] | (gt8[..., 1] << 8) | (gt8[..., 2] << 16) | (gt8[..., 3] << 24)
    pred = rng.uniform(0.05, 0.95, (4, h, w)).astype(np.float32)
    wts = rng.uniform(-1, 1, (4, h, w)).astype(np.float32)
    for bg, mask in ((None, False), ((0.3, 0.5, 0.2), False), (None, True)):
        g = bo.image_loss_backward(pred, packed, wts, 0.8, -0.2, bg=bg, mask=mask)
        for (c, y, x) in [(0, 3, 4), (1, 12, 14), (2, 23, 27), (3, 5, 5), (0, 0, 0)]:
            def f(d):
                p = pred.copy()
                p[c, y, x] += d
                return float((bo.image_loss_forward(p, packed, 0.8, -0.2, bg=bg, mask=mask).astype(np.float64) * wts).sum())
            num = (f(1e-3) - f(-1e-3)) / 2e-3
            assert abs(num - g[c, y, x]) < 2e-3 + 0.02 * abs(num), (c, y, x, num, g[c, y, x])


def test_adam_matches_closed_form():
    """adam_scaled.rs:93-147 against a float64 restatement."""
    rng = np.random.default_rng(4)
    rows, rl = 7, 6
    p = rng.normal(size=(rows, rl)).astype(np.float32)
    p64 = p.astype(np.float64)
    m1 = np.zeros_like(p); m2 = np.zeros_like(p)
    m1r = np.zeros_like(p); m2r = np.zeros(rows, np.float32); pr = p.copy()
    M1 = np.zeros_like(p64); M2 = np.zeros_like(p64)
    scale = np.array([1, 1, 1, 0.1, 0.1, 0.1], np.float32)
    for t in range(1, 6):
        g = rng.normal(size=(rows, rl)).astype(np.float32)
        bo.adam_step(p, g, m1, m2, 0.01, t, col_scale=scale)
        bo.adam_step(pr, g, m1r, m2r, 0.01, t, col_scale=scale, reduce_m2=True)
        M1 = 0.9 * M1 + 0.1 * g if t > 1 else 0.1 * g
        M2 = 0.999 * M2 + 0.001 * g.astype(np.float64) ** 2 if t > 1 else 0.001 * g.astype(np.float64) ** 2
        upd = (M1 / (1 - 0.9 ** t)) / (np.sqrt(M2 / (1 - 0.999 ** t)) + 1e-15)
        p64 = p64 - upd * (scale.astype(np.float64) * 0.01)
        assert np.allclose(p, p64, rtol=2e-5, atol=2e-6)
    assert np.isfinite(pr).all() and not np.allclose(pr, p)
