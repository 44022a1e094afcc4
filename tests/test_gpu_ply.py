"""PLY export / import on the MI355X: rows packed / unpacked by HIP kernels,
byte-exact against the oracle's numpy restatement, and the reference's round-trip tests
(brush-serde/src/export.rs:305-349) through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import ply
import util

pytestmark = pytest.mark.gpu


def _splats(n, deg, seed=0):
    rng = np.random.default_rng(seed + deg)
    c = (deg + 1) ** 2
    return (rng.normal(size=(n, 10)).astype(np.float32), rng.normal(size=(n, c, 3)).astype(np.float32), rng.normal(size=n).astype(np.float32))


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("n", [1, 100, 4097])
def test_export_bytes_equal_oracle(dev, deg, n):
    import brush_amd as ba
    tr, sh, op = _splats(n, deg)
    spl = ba.Splats(tr, sh, op, render_mip=(deg % 2 == 1), device=dev)
    up = (0.0, 0.5, -1.0) if deg == 2 else None
    got = ba.splat_to_ply(spl, up_axis=up)
    want = ply.splat_to_ply(tr, sh, op, render_mip=(deg % 2 == 1), up_axis=up)
    assert got == want


def test_export_bakes_the_floor(dev, oracle_lib):
    import brush_amd as ba
    tr, sh, op = _splats(5000, 1)
    tr[:, 7:] = np.random.default_rng(1).uniform(-6, -2, (5000, 3))
    f = np.random.default_rng(2).uniform(0.001, 0.05, 5000).astype(np.float32)
    spl = ba.Splats(tr, sh, op, device=dev, min_scale=f)
    assert ba.splat_to_ply(spl) == ply.splat_to_ply(tr, sh, op, min_scale=f)
    assert spl.min_scale is not None and np.array_equal(spl.transforms.cpu().numpy(), tr)  # export does not modify the live splats


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_roundtrip_through_the_device(dev, deg):
    """export.rs:305-349 (test_roundtrip_sh_coefficient_ordering, test_export_roundtrip_multiple_splats)."""
    import brush_amd as ba
    tr, sh, op = _splats(100, deg)
    spl = ba.Splats(tr, sh, op, render_mip=True, device=dev)
    back, meta = ba.load_splat_from_ply(ba.splat_to_ply(spl, up_axis=(0.0, 0.0, 1.0)), device=dev)
    assert back.num_splats() == 100 and back.sh_degree() == deg and back.render_mip and meta.up_axis == (0.0, 0.0, 1.0)
    assert torch.equal(back.sh_coeffs, spl.sh_coeffs) and torch.equal(back.raw_opacities, spl.raw_opacities)
    want = ply.load_splat_from_ply(ply.splat_to_ply(tr, sh, op))
    assert np.array_equal(back.transforms.cpu().numpy(), want["transforms"])


def test_import_arbitrary_column_order_and_defaults(dev):
    import brush_amd as ba
    rng = np.random.default_rng(4)
    n = 777
    props = ["nx", "opacity", "z", "f_dc_2", "x", "rot_1", "rot_0", "y", "f_dc_0", "rot_3", "rot_2", "f_dc_1", "extra"]
    rows = rng.normal(size=(n, len(props))).astype("<f4")
    data = ("ply\nformat binary_little_endian 1.0\ncomment splatrendermode: MIP\nelement vertex %d\n" % n).encode() + \
        b"".join(("property float %s\n" % p).encode() for p in props) + b"element face 0\nproperty list uchar int vertex_indices\nend_header\n" + rows.tobytes()
    spl, meta = ba.load_splat_from_ply(data, device=dev)
    want = ply.load_splat_from_ply(data)
    assert meta.render_mode == "mip" and spl.render_mip and meta.sh_degree == 0
    assert np.array_equal(spl.transforms.cpu().numpy(), want["transforms"]) and (want["transforms"][:, 7:] == -4.0).all()
    assert np.array_equal(spl.sh_coeffs.cpu().numpy(), want["sh"]) and np.array_equal(spl.raw_opacities.cpu().numpy(), want["raw_opac"])
    pts = np.arange(12, dtype="<f4").reshape(4, 3)
    bare = b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nend_header\n" + pts.tobytes()
    spl, _ = ba.load_splat_from_ply(bare, device=dev)
    assert (spl.transforms.cpu().numpy()[:, 3:7] == [1, 0, 0, 0]).all() and (spl.sh_coeffs.cpu().numpy() == 0.5).all() and (spl.raw_opacities.cpu().numpy() == 0).all()


def test_exported_scene_renders_identically_after_reimport(dev, oracle_lib):
    """A trained-looking scene written and read back renders the same image (quaternions are only
    renormalised, which the projection does anyway) — the end-to-end meaning of the round trip."""
    import brush_amd as ba
    from brush_amd import synth
    scene, w, h = synth.config_scene("10k_256", 2)
    cp = synth.default_camera_params(w, h)
    spl = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
    img0, _ = ba.render_splats(spl, util.hip_camera(ba, cp), (w, h), (0, 0, 0), ba.RasterPass.Backward)
    back, _ = ba.load_splat_from_ply(ba.splat_to_ply(spl), device=dev)
    img1, _ = ba.render_splats(back, util.hip_camera(ba, cp), (w, h), (0, 0, 0), ba.RasterPass.Backward)
    assert (img0 - img1).abs().max().item() <= 1e-4  # renormalised quaternions round differently: within the 1e-4 image bar


@pytest.mark.parametrize("deg,n", [(0, 1), (0, 255), (1, 257), (2, 4097), (3, 100000)])
def test_compressed_import_equals_oracle(dev, deg, n):
    """SuperSplat-compressed files (import.rs:407-600, quant.rs): packed words decoded by one HIP thread per splat.
    Everything is f32 arithmetic in the reference's order -> bit-exact, except the logit of the alpha byte (device
    logf: 2 ulp) and NaN payloads of impossible quaternions."""
    import brush_amd as ba
    data = ply.make_compressed_ply(n, deg, seed=deg + n, chunk_order=list(reversed(ply.CHUNK_PROPS)) if deg == 1 else None,
                                   extra_chunk_props=("pad0",) if deg == 2 else ())
    want = ply.load_compressed_ply(data)
    spl, meta = ba.load_splat_from_ply(data, device=dev)
    assert meta.compressed and meta.total_splats == n and meta.sh_degree == deg
    tr, sh, op = spl.transforms.cpu().numpy(), spl.sh_coeffs.cpu().numpy(), spl.raw_opacities.cpu().numpy()
    assert np.array_equal(tr[:, :3], want["transforms"][:, :3]) and np.array_equal(tr[:, 7:], want["transforms"][:, 7:])
    assert np.array_equal(np.isnan(tr[:, 3:7]), np.isnan(want["transforms"][:, 3:7]))
    assert np.allclose(tr[:, 3:7], want["transforms"][:, 3:7], rtol=0, atol=2e-7, equal_nan=True)   # sqrt / div: correctly rounded both sides
    assert np.array_equal(sh[:, 1:], want["sh"][:, 1:])
    assert np.allclose(sh[:, 0], want["sh"][:, 0], rtol=2e-7, atol=1e-7)                          # the division by SH_C0
    fin = np.isfinite(want["raw_opac"])
    assert np.array_equal(op[~fin], want["raw_opac"][~fin])                                        # alpha 0 / 255 -> -inf / +inf
    assert np.allclose(op[fin], want["raw_opac"][fin], rtol=3e-7, atol=3e-7)


def test_compressed_import_renders(dev, oracle_lib):
    """The decoded splats go straight into the hot path: GPU render of the imported splats == oracle render of the same rows."""
    import brush_amd as ba
    data = ply.make_compressed_ply(3000, 1, seed=11)
    spl, _ = ba.load_splat_from_ply(data, device=dev)
    tr, sh, op = spl.transforms.cpu().numpy(), spl.sh_coeffs.cpu().numpy(), spl.raw_opacities.cpu().numpy()
    ok = np.isfinite(tr).all(axis=1) & np.isfinite(op)
    assert ok.sum() > 2800
    scene = dict(transforms=tr[ok], sh=sh[ok], raw_opac=op[ok])
    cp = dict(pos=(0.5, 0.5, -6.0), rot_xyzw=(0.0, 0.0, 0.0, 1.0), fov_x=0.9, fov_y=0.7, center_uv=(0.5, 0.5))
    w, h = 96, 64
    spl2 = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
    img, aux = ba.render_splats(spl2, util.hip_camera(ba, cp), (w, h), (0.0, 0.0, 0.0), ba.RasterPass.Backward)
    ref = oracle_lib.Render().forward(oracle_lib.camera(img_w=w, img_h=h, **cp), scene["transforms"], scene["sh"], scene["raw_opac"])
    assert aux.num_visible == ref.num_visible > 1000 and aux.num_intersections == ref.num_intersections
    assert np.abs(img.cpu().numpy() - ref.image()).max() <= 1e-6


@pytest.mark.parametrize("colour", ["uchar", "ushort", "float", None])
def test_mixed_type_rows_equal_oracle(dev, colour):
    """Vertex rows of mixed scalar types and the quantised colour override (ply_gaussian.rs:36-99): every value is a cast or one
    division / one rgb_to_sh -> bit-exact against the oracle."""
    import brush_amd as ba
    from test_oracle_ply import _mixed_ply
    data, _ = _mixed_ply(5000, seed=5, colour=colour)
    want = ply.load_splat_from_ply(data)
    spl, meta = ba.load_splat_from_ply(data, device=dev)
    assert meta.total_splats == 5000 and meta.sh_degree == 0
    assert np.array_equal(spl.transforms.cpu().numpy(), want["transforms"])
    assert np.array_equal(spl.raw_opacities.cpu().numpy(), want["raw_opac"])
    got_sh = spl.sh_coeffs.cpu().numpy()
    assert np.allclose(got_sh, want["sh"], rtol=2e-7, atol=1e-7) if colour else np.array_equal(got_sh, want["sh"])


def _expected_pick(n, subsample_points, max_splats):
    """Row indices the reference keeps: parse (import.rs:346-349) then SplatData::subsample (import.rs:49-74)."""
    s = subsample_points or 1
    idx = np.arange(s - 1, (n // s) * s, s)
    if max_splats and idx.size > max_splats:
        step = -(-idx.size // max_splats)
        idx = idx[::step]
    return idx


@pytest.mark.parametrize("kind", ["plain", "compressed", "mixed"])
@pytest.mark.parametrize("subsample_points,max_splats", [(None, None), (2, None), (7, None), (None, 3), (None, 333), (3, 100), (None, 10 ** 9)])
def test_import_with_subsample(dev, kind, subsample_points, max_splats):
    """import.rs:651-670 (test_import_with_subsample) and :672-715 (test_splat_data_subsample: step = ceil(10 / 3) = 4 -> rows 0, 4, 8)."""
    import brush_amd as ba
    n = 1000
    if kind == "plain":
        data = ply.splat_to_ply(*_splats(n, 2))
        full = ply.load_splat_from_ply(data)
    elif kind == "compressed":
        data = ply.make_compressed_ply(n, 2, seed=9)
        full = ply.load_compressed_ply(data)
    else:
        from test_oracle_ply import _mixed_ply
        data = _mixed_ply(n, seed=2, colour="uchar")[0]
        full = ply.load_splat_from_ply(data)
    idx = _expected_pick(n, subsample_points, max_splats)
    spl, meta = ba.load_splat_from_ply(data, device=dev, subsample_points=subsample_points, max_splats=max_splats)
    assert meta.total_splats == idx.size == spl.transforms.shape[0]
    if max_splats:
        assert idx.size <= max_splats
    assert np.allclose(spl.transforms.cpu().numpy(), full["transforms"][idx], rtol=0, atol=2e-7, equal_nan=True)
    assert np.allclose(spl.sh_coeffs.cpu().numpy(), full["sh"][idx], rtol=2e-7, atol=1e-7)
    fin = np.isfinite(full["raw_opac"][idx])
    assert np.allclose(spl.raw_opacities.cpu().numpy()[fin], full["raw_opac"][idx][fin], rtol=3e-7, atol=3e-7)


def test_splat_data_subsample_reference_case(dev):
    """import.rs:672-715 with its numbers: 10 rows, max 3 -> rows 0, 4, 8; within budget / 0 -> untouched."""
    import brush_amd as ba
    tr, sh, op = _splats(10, 0)
    tr[:, 0] = np.arange(10)
    data = ply.splat_to_ply(tr, sh, op)
    for cap, want in ((10, list(range(10))), (0, list(range(10))), (3, [0, 4, 8])):
        spl, meta = ba.load_splat_from_ply(data, device=dev, max_splats=cap)
        assert spl.transforms[:, 0].cpu().tolist() == [float(v) for v in want] and meta.total_splats == len(want)
    with pytest.raises(ba.BrushHipError):
        ctx = ba.get_context(dev)
        t = torch.empty((10, 10), device=dev)
        ctx.check(ctx.lib.bh_splats_from_ply_strided(ctx._h, data, len(data), 5, 2, 4, t.data_ptr(), t.data_ptr(), t.data_ptr()))   # rows 5 7 9 11: past the end


def test_mutated_headers_that_parse_also_load(dev):
    """Whatever the host parser accepts, the device decode must take without faulting (its row / column offsets come from the
    header): mutated files that still parse are loaded, and the context stays usable."""
    import brush_amd as ba
    from test_oracle_ply import _mixed_ply
    rng = np.random.default_rng(1)
    bases = [ply.make_compressed_ply(600, 2, seed=3), ply.splat_to_ply(*_splats(50, 3)), _mixed_ply(80, colour="uchar")[0]]
    loaded = 0
    for it in range(900):
        b = bytearray(bases[it % 3])
        hdr_end = b.index(b"end_header") + 11
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, hdr_end))
            if rng.integers(0, 2):
                b[pos] = int(rng.integers(32, 127))
            else:
                b[pos:pos] = bytes(rng.integers(48, 58, 1, dtype=np.uint8))   # a stray digit: counts and indices change
        try:
            meta = ba.ply_parse_header(bytes(b))
        except ba.BrushHipError:
            continue
        if meta.total_splats > 2_000_000:
            continue
        try:
            spl, _ = ba.load_splat_from_ply(bytes(b), device=dev)
            loaded += 1
            assert spl.transforms.shape[0] == meta.total_splats
        except ba.BrushHipError:
            pass
    assert loaded > 10
    torch.cuda.synchronize()
    want = ply.load_compressed_ply(bases[0])
    spl, _ = ba.load_splat_from_ply(bases[0], device=dev)
    assert np.array_equal(spl.transforms.cpu().numpy()[:, :3], want["transforms"][:, :3])
