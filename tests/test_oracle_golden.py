"""The oracle is pinned by the reference's own known-answer vectors before it is
trusted as the checker."""
import math

import numpy as np
import pytest

from oracle import bo
import util


@pytest.mark.parametrize("name", ["tiny_case", "basic_case"])
def test_forward_matches_reference_golden(name):
    """crates/brush-bench-test/src/reference.rs:80-151: gsplat-CUDA render of 4 / 16
    SH-degree-3 splats at 123x82; tolerance 1e-5 abs + 1e-2 rel (reference.rs:50-51)."""
    scene, ref = util.golden_case(name)
    h, w, _ = ref.shape
    cam = bo.camera(**util.golden_camera_params(w, h))
    for flags in (bo.FLAG_BWD_INFO, bo.FLAG_BWD_INFO | bo.FLAG_SMOOTH_CUTOFF):
        r = bo.Render().forward(cam, scene["transforms"], scene["sh"], scene["raw_opac"], flags=flags)
        img = r.image()
        assert not np.isnan(img).any()
        if flags & bo.FLAG_SMOOTH_CUTOFF:
            # the C^1 cutoff shifts edge pixels by < 1/255 (gaussian_splats.rs:44-47)
            assert np.abs(img - ref).max() < 1.0 / 255.0
        else:
            tol = 1e-5 + 1e-2 * np.abs(ref)
            assert (np.abs(img - ref) < tol).all(), "max |d| = %g" % np.abs(img - ref).max()


@pytest.mark.parametrize("name", ["tiny_case", "basic_case"])
def test_packed_output_matches_float(name):
    """kernels/rasterize.rs:173-179: rgba8 packing of the same blend."""
    scene, ref = util.golden_case(name)
    h, w, _ = ref.shape
    cam = bo.camera(**util.golden_camera_params(w, h))
    rf = bo.Render().forward(cam, scene["transforms"], scene["sh"], scene["raw_opac"], flags=bo.FLAG_BWD_INFO)
    rp = bo.Render().forward(cam, scene["transforms"], scene["sh"], scene["raw_opac"], flags=0)
    img = rf.image()
    packed = rp.get("out_packed").reshape(h, w)
    for c in range(4):
        expect = np.clip(img[..., c] * np.float32(255.0), 0, 255).astype(np.uint32)
        assert np.array_equal((packed >> (8 * c)) & 0xFF, expect)


def test_exp_log_accuracy():
    L = bo.lib()
    xs = np.linspace(-87.0, 88.0, 20001)
    e = np.array([L.bo_expf(float(np.float32(x))) for x in xs], np.float64)
    ref = np.exp(np.float32(xs).astype(np.float64))
    assert np.max(np.abs(e - ref) / ref) < 2.5e-7
    xs = np.exp(np.linspace(-80.0, 80.0, 20001))
    l = np.array([L.bo_logf(float(np.float32(x))) for x in xs], np.float64)
    ref = np.log(np.float32(xs).astype(np.float64))
    assert np.max(np.abs(l - ref) / np.maximum(np.abs(ref), 1e-2)) < 3e-7
    assert L.bo_expf(float("-inf")) == 0.0 and math.isinf(L.bo_expf(float("inf"))) and math.isnan(L.bo_expf(float("nan")))
    assert math.isinf(L.bo_logf(0.0)) and math.isnan(L.bo_logf(-1.0)) and L.bo_logf(1.0) == 0.0


def test_sort_reference_vectors():
    """brush-sort/src/lib.rs:154-201 test_sorting: 15 keys x 128 variants vs stable argsort."""
    for i in range(128):
        keys = np.array([5 + i * 4, i, 6, 123, 74657, 123, 999, 2 ** 24 + 123, 6, 7, 8, 0, i * 2, 16 + i, 128 * i], np.uint32)
        vals = keys * 2 + 5
        ok, ov = bo.radix_argsort(keys, vals, 32)
        idx = np.argsort(keys, kind="stable")
        assert np.array_equal(ok, keys[idx]) and np.array_equal(ov, vals[idx])


def test_sort_low_bits_only():
    rng = np.random.default_rng(1)
    keys = rng.integers(0, 2 ** 32, 5000, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(5000, dtype=np.uint32)
    ok, ov = bo.radix_argsort(keys, vals, 13)
    idx = np.argsort(keys & 0x1FFF, kind="stable")
    assert np.array_equal(ov, vals[idx]) and np.array_equal(ok, keys[idx])


def test_prefix_sum_reference_vectors():
    """brush-prefix-sum/src/lib.rs:105-160"""
    assert np.array_equal(bo.prefix_sum(np.array([1, 1, 1, 1], np.uint32)), [1, 2, 3, 4])
    data = (90 + np.arange(1024)).astype(np.uint32)
    assert np.array_equal(bo.prefix_sum(data), np.cumsum(data, dtype=np.uint64).astype(np.uint32))
    it = np.arange(512 * 16 + 123)
    data = np.stack([2 + it, 0 * it, 32 + 0 * it, 512 + 0 * it, 30965 + 0 * it], axis=1).reshape(-1).astype(np.uint32)
    assert np.array_equal(bo.prefix_sum(data), np.cumsum(data, dtype=np.uint64).astype(np.uint32))


def test_camera_roundtrip_and_uniforms():
    """camera.rs fov<->focal round trips (tests/mod.rs:711-789) and the +Z-forward pose."""
    L = bo.lib()
    for fov in (0.3, 1.0, 2.0):
        for px in (64, 1920):
            assert abs(L.bo_focal_to_fov(L.bo_fov_to_focal(fov, px), px) - fov) < 1e-12
    cam = bo.camera(pos=(1.0, 2.0, 3.0), fov_x=1.0, fov_y=0.8, img_w=200, img_h=100)
    assert np.allclose(list(cam.vm)[:9], [1, 0, 0, 0, 1, 0, 0, 0, 1])
    assert np.allclose(list(cam.vm)[9:], [-1, -2, -3])
    assert abs(cam.fx - 100.0 / math.tan(0.5)) < 1e-3 and cam.cx == 100.0 and cam.cy == 50.0
    assert abs(cam.lim_pos_x - (1.15 * 200 - 100) / cam.fx) < 1e-6
    # rotated camera: world_to_local = inverse(rotation, translation)
    q = util.quat_from_axis_angle((0.2, 1.0, -0.3), 0.7)
    cam = bo.camera(pos=(0.5, -1.0, 2.0), rot_xyzw=q, img_w=64, img_h=64)
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    vm = np.array(list(cam.vm)).reshape(4, 3).T  # 3x4
    assert np.allclose(vm[:, :3], R.T, atol=1e-6)
    assert np.allclose(vm[:, 3], -R.T @ np.array([0.5, -1.0, 2.0]), atol=1e-6)
