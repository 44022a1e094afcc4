"""Forward parity of the HIP pipeline through the C ABI: golden tensors, stage-wise
bit-exactness against the oracle (tile assignment / sort order / counts are integer
work -> exact; projected records and images are f32 on the same operation sequence ->
exact as well, asserted with a tight tolerance), and the reference's property tests."""
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
import util

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-6  # L-inf on [0,1] images vs the oracle (north_star allows 1e-4)


def render_both(ba, bo, dev, scene, cam_params, w, h, bg=(0.0, 0.0, 0.0), pass_=None, mip=False):
    pass_ = pass_ or ba.RasterPass.Backward
    spl = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], render_mip=mip, device=dev)
    img, aux = ba.render_splats(spl, util.hip_camera(ba, cam_params), (w, h), bg, pass_)
    flags = (bo.FLAG_BWD_INFO if pass_.bwd_info() else 0) | (bo.FLAG_SMOOTH_CUTOFF if pass_.smooth_cutoff() else 0) | (bo.FLAG_MIP if mip else 0)
    p = {k: v for k, v in cam_params.items() if k not in ("img_w", "img_h")}
    ref = bo.Render().forward(bo.camera(img_w=w, img_h=h, **p), scene["transforms"], scene["sh"], scene["raw_opac"], bg=bg, flags=flags)
    return img, aux, ref


def assert_stagewise_exact(aux, ref):
    assert aux.num_visible == ref.num_visible and aux.num_intersections == ref.num_intersections
    assert np.array_equal(util.u32(aux.intersect_counts), ref.get("intersect_counts"))
    assert np.array_equal(util.u32(aux.global_from_compact_gid), ref.get("global_from_compact_gid")), "depth order"
    assert np.array_equal(util.u32(aux.cum_tiles_hit), ref.get("cum_tiles_hit"))
    assert np.array_equal(util.u32(aux.tile_id_from_isect), ref.get("tile_id_from_isect")), "tile assignment"
    assert np.array_equal(util.u32(aux.compact_gid_from_isect), ref.get("compact_gid_from_isect")), "per-tile sort order"
    assert np.array_equal(util.u32(aux.tile_offsets).reshape(-1), ref.get("tile_offsets"))
    assert np.array_equal(aux.depths_sorted.cpu().numpy(), ref.get("depths_sorted"))
    assert np.array_equal(aux.max_radius.cpu().numpy(), ref.get("max_radius"))
    proj = aux.projected_splats.cpu().numpy().reshape(-1)
    assert np.array_equal(proj, ref.get("projected")), "projected records (xy, conic, alpha, rgb)"
    if aux.visible is not None:
        assert np.array_equal(aux.visible.cpu().numpy(), ref.get("visible"))


@pytest.mark.parametrize("name", ["tiny_case", "basic_case"])
def test_reference_golden_images(dev, oracle_lib, name):
    """crates/brush-bench-test/src/reference.rs:80-151 through the HIP path."""
    import brush_amd as ba
    scene, ref_img = util.golden_case(name)
    h, w, _ = ref_img.shape
    cp = util.golden_camera_params(w, h)
    img, aux, ref = render_both(ba, oracle_lib, dev, scene, cp, w, h)
    img = img.cpu().numpy()
    tol = 1e-5 + 1e-2 * np.abs(ref_img)
    assert (np.abs(img - ref_img) < tol).all()
    assert np.abs(img - ref.image()).max() <= IMG_TOL
    assert_stagewise_exact(aux, ref)


@pytest.mark.parametrize("sh_degree,mip", [(0, False), (3, False), (1, True), (4, False)])
def test_config0_10k_256_exact_vs_oracle(dev, oracle_lib, sh_degree, mip):
    """BASELINE.json configs[0]: 10k random splats, 256x256."""
    import brush_amd as ba
    scene, w, h = synth.config_scene("10k_256", sh_degree)
    cp = synth.default_camera_params(w, h)
    img, aux, ref = render_both(ba, oracle_lib, dev, scene, cp, w, h, bg=(0.1, 0.2, 0.3), mip=mip)
    assert_stagewise_exact(aux, ref)
    assert np.abs(img.cpu().numpy() - ref.image()).max() <= IMG_TOL
    aux.validate(10_000)


def test_forward_only_packed_output(dev, oracle_lib):
    import brush_amd as ba
    scene, w, h = synth.config_scene("10k_256", 0)
    cp = synth.default_camera_params(w, h)
    img, aux, ref = render_both(ba, oracle_lib, dev, scene, cp, w, h, bg=(0.3, 0.1, 0.6), pass_=ba.RasterPass.Forward)
    assert np.array_equal(util.u32(img).reshape(-1), ref.get("out_packed"))
    assert aux.visible is None


def test_smooth_cutoff_pass(dev, oracle_lib):
    import brush_amd as ba
    scene, w, h = synth.config_scene("10k_256", 0, n=3000)
    cp = synth.default_camera_params(w, h)
    img, aux, ref = render_both(ba, oracle_lib, dev, scene, cp, w, h, pass_=ba.RasterPass.BackwardSmoothCutoff)
    assert np.abs(img.cpu().numpy() - ref.image()).max() <= IMG_TOL


@pytest.mark.parametrize("w,h", [(1, 1), (15, 17), (16, 16), (123, 82), (257, 257), (640, 360)])
def test_odd_image_sizes_rotated_offcentre_camera(dev, oracle_lib, w, h):
    """tests/mod.rs + finite_diff.rs:903-989 camera variety; ragged tiles."""
    import brush_amd as ba
    scene = synth.make_scene(1500, 0x77, sh_degree=2, log_scale_range=(math.log(0.03), math.log(0.3)))
    cp = dict(pos=(0.4, -0.3, -0.5), rot_xyzw=util.quat_from_axis_angle((0.2, 1.0, 0.1), 0.3), fov_x=1.0, fov_y=0.7, center_uv=(0.45, 0.56))
    img, aux, ref = render_both(ba, oracle_lib, dev, scene, cp, w, h, bg=(0.5, 0.5, 0.5))
    assert_stagewise_exact(aux, ref)
    assert np.abs(img.cpu().numpy() - ref.image()).max() <= IMG_TOL


@pytest.mark.parametrize("w,h", [(20000, 48), (48, 18000)])
def test_panoramic_image_wider_than_1023_tiles(dev, oracle_lib, w, h):
    """No reference limit on the image size: the walk's box records hold 16-bit tile coordinates and its candidate index ->
    (row, column) split is corrected to exact integer division, so tile grids beyond 1023 tiles per side (round 1's limit)
    and splats whose boxes span more than a thousand tiles are walked like any other."""
    import brush_amd as ba
    scene = synth.make_scene(300, 0x5A, sh_degree=0, log_scale_range=(math.log(0.02), math.log(0.2)))
    scene["transforms"][:12, 7:10] = math.log(30.0)   # a dozen giants: boxes over the whole strip
    scene["raw_opac"][:12] = -3.0
    fov_long, fov_short = 2.4, 2.4 * min(w, h) / max(w, h)
    cp = dict(pos=(0.0, 0.0, -1.5), rot_xyzw=(0.0, 0.0, 0.0, 1.0), fov_x=fov_long if w > h else fov_short, fov_y=fov_short if w > h else fov_long,
              center_uv=(0.5, 0.5))
    img, aux, ref = render_both(ba, oracle_lib, dev, scene, cp, w, h, bg=(0.2, 0.1, 0.0))
    assert aux.num_intersections > 5000 and max(w, h) // 16 > 1023
    assert_stagewise_exact(aux, ref)
    assert np.abs(img.cpu().numpy() - ref.image()).max() <= IMG_TOL


def test_few_splats_under_a_large_tile_table(dev, oracle_lib):
    """K1 clears the tile table and the visible flags on its way, one word per splat thread; with 40 splats under a
    1920x1080 frame (16 448 table words) its grid cannot cover them and the launcher has to fall back to plain fills.
    A dense render first leaves every table entry and flag dirty."""
    import brush_amd as ba
    w, h = 1920, 1080
    cp = synth.default_camera_params(w, h)
    dense = synth.make_scene(30000, 0x91, sh_degree=0, log_scale_range=(math.log(0.05), math.log(0.3)))
    render_both(ba, oracle_lib, dev, dense, cp, w, h)
    few = synth.make_scene(40, 0x92, sh_degree=0, log_scale_range=(math.log(0.05), math.log(0.4)))
    for _ in range(2):   # twice: both halves of the counter ping-pong
        img, aux, ref = render_both(ba, oracle_lib, dev, few, cp, w, h, bg=(0.1, 0.2, 0.3))
        assert_stagewise_exact(aux, ref)
        assert np.abs(img.cpu().numpy() - ref.image()).max() <= IMG_TOL
    # and an empty scene right after a dirty one: no K1 at all, fills only
    spl = ba.Splats(np.zeros((0, 10), np.float32), np.zeros((0, 1, 3), np.float32), np.zeros((0,), np.float32), device=dev)
    img, aux = ba.render_splats(spl, util.hip_camera(ba, cp), (w, h), (0.0, 0.0, 0.0), ba.RasterPass.Backward)
    assert aux.num_visible == 0 and aux.num_intersections == 0 and float(img.abs().max()) == 0.0
    assert int(aux.tile_offsets.to(torch.int64).abs().sum()) == 0


def test_empty_render_and_zero_size(dev):
    """tests/mod.rs:20; render.rs:50-53 assert -> error."""
    import brush_amd as ba
    spl = ba.Splats(np.zeros((0, 10), np.float32), np.zeros((0, 1, 3), np.float32), np.zeros((0,), np.float32), device=dev)
    cam = util.hip_camera(ba, util.STD_CAM)
    img, aux = ba.render_splats(spl, cam, (33, 17), (0.25, 0.5, 0.75), ba.RasterPass.Backward)
    img = img.cpu().numpy()
    assert aux.num_visible == 0 and aux.num_intersections == 0
    assert np.allclose(img[..., :3], [0.25, 0.5, 0.75]) and np.all(img[..., 3] == 0)
    with pytest.raises(ba.BrushHipError):
        ba.render_splats(spl, cam, (0, 10), (0, 0, 0))


def test_determinism_bit_exact(dev):
    """tests/mod.rs:288: repeated renders are bit-identical (ours also across equal depths)."""
    import brush_amd as ba
    scene = synth.make_scene(20000, 0x21, log_scale_range=(math.log(0.02), math.log(0.2)))
    scene["transforms"][::3, 2] = 5.0  # many exactly equal depths
    spl = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
    cam = util.hip_camera(ba, synth.default_camera_params(320, 200))
    a, aux_a = ba.render_splats(spl, cam, (320, 200), (0, 0, 0), ba.RasterPass.Backward)
    for _ in range(3):
        b, aux_b = ba.render_splats(spl, cam, (320, 200), (0, 0, 0), ba.RasterPass.Backward)
        assert torch.equal(a, b) and torch.equal(aux_a.compact_gid_from_isect, aux_b.compact_gid_from_isect)


def test_culled_splats_do_not_perturb(dev):
    """tests/mod.rs:315,360"""
    import brush_amd as ba
    sc = synth.make_scene(3000, 0x31, log_scale_range=(math.log(0.03), math.log(0.3)))
    cam = util.hip_camera(ba, synth.default_camera_params(128, 96))
    base, _ = ba.render_splats(ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev), cam, (128, 96), (0, 0, 0), ba.RasterPass.Backward)
    ex = synth.make_scene(1000, 0x32)
    ex["transforms"][:, 2] = -4.0
    tr = np.concatenate([sc["transforms"], ex["transforms"]]); sh = np.concatenate([sc["sh"], ex["sh"]]); op = np.concatenate([sc["raw_opac"], ex["raw_opac"]])
    img, aux = ba.render_splats(ba.Splats(tr, sh, op, device=dev), cam, (128, 96), (0, 0, 0), ba.RasterPass.Backward)
    assert torch.equal(img, base)
    assert float(aux.visible[3000:].sum()) == 0.0


def test_fullscreen_splats_no_dropped_tile(dev, oracle_lib):
    """tests/mod.rs:394-451: many screen-filling splats; every tile gets its full list."""
    import brush_amd as ba
    n = 3000
    sc = synth.make_scene(n, 0x41, log_scale_range=(math.log(0.8), math.log(1.5)))
    cp = synth.default_camera_params(160, 96)
    img, aux, ref = render_both(ba, oracle_lib, dev, sc, cp, 160, 96)
    assert_stagewise_exact(aux, ref)
    offs = util.u32(aux.tile_offsets).reshape(-1, 2)
    assert (offs[:, 1] > offs[:, 0]).all()
    assert np.abs(img.cpu().numpy() - ref.image()).max() <= IMG_TOL


def test_fuzz_poisoned_inputs_match_oracle(dev, oracle_lib):
    """fuzz.rs:61-330: NaN/Inf/denormal/huge values in every slot; counts and the visible
    set must match the oracle exactly and the image stays finite."""
    import brush_amd as ba
    from test_oracle_properties import POISON
    rng = np.random.default_rng(99)
    sizes = [(1, 1), (16, 16), (17, 31), (64, 48), (257, 257)]
    for it in range(30):
        n = int(rng.integers(1, 200))
        sc = synth.make_scene(n, 3000 + it, sh_degree=int(rng.integers(0, 3)), log_scale_range=(math.log(0.03), math.log(0.3)))
        for arr in (sc["transforms"], sc["sh"], sc["raw_opac"]):
            flat = arr.reshape(-1)
            k = max(1, int(0.05 * flat.size))
            flat[rng.integers(0, flat.size, k)] = np.array(POISON, np.float32)[rng.integers(0, len(POISON), k)]
        w, h = sizes[it % len(sizes)]
        cp = synth.default_camera_params(w, h)
        img, aux, ref = render_both(ba, oracle_lib, dev, sc, cp, w, h, bg=(0.1, 0.1, 0.1))
        assert aux.num_visible == ref.num_visible and aux.num_intersections == ref.num_intersections
        assert np.array_equal(util.u32(aux.compact_gid_from_isect), ref.get("compact_gid_from_isect"))
        im = img.cpu().numpy()
        assert np.isfinite(im).all()
        assert np.allclose(im, ref.image(), atol=1e-5, equal_nan=False)


def test_bad_geometry_is_fully_culled(dev):
    """fuzz.rs:332-446"""
    import brush_amd as ba
    sc = synth.make_scene(64, 0x51, log_scale_range=(math.log(0.03), math.log(0.3)))
    cam = util.hip_camera(ba, synth.default_camera_params(64, 64))
    for col, val in ((3, 0.0), (0, float("nan")), (7, float("inf")), (2, 1e11), (2, -1.0)):
        tr = sc["transforms"].copy()
        if col == 3:
            tr[:, 3:7] = 0.0
        else:
            tr[:, col] = val
        _, aux = ba.render_splats(ba.Splats(tr, sc["sh"], sc["raw_opac"], device=dev), cam, (64, 64), (0, 0, 0), ba.RasterPass.Backward)
        assert aux.num_visible == 0 and aux.num_intersections == 0
