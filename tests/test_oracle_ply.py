"""PLY export / import: the oracle's numpy restatement against the properties the
reference's own tests assert (brush-serde/src/export.rs:206-349: field counts per SH degree, SH
coefficient ordering round trip, multi-splat round trip) and the product's host-side header parser
(bh_ply_parse_header — host code, runs without a GPU) against files the oracle writes."""
import numpy as np
import pytest

from oracle import ply


def _splats(n, deg, seed=0):
    rng = np.random.default_rng(seed + deg)
    c = (deg + 1) ** 2
    return (rng.normal(size=(n, 10)).astype(np.float32), rng.normal(size=(n, c, 3)).astype(np.float32), rng.normal(size=n).astype(np.float32))


@pytest.mark.parametrize("deg,rest_fields", [(0, 0), (1, 9), (2, 24), (3, 45), (4, 72)])
def test_ply_field_count_matches_sh_degree(deg, rest_fields):
    """export.rs:279-303"""
    tr, sh, op = _splats(3, deg)
    text = ply.splat_to_ply(tr, sh, op).decode("latin1")
    assert text.count("property float f_rest_") == rest_fields
    assert "f_dc_0" in text
    assert ("f_rest_0" in text) == (rest_fields > 0)
    assert "f_rest_%d\n" % rest_fields not in text
    assert len(ply.splat_to_ply(tr, sh, op)) == len(ply.header(3, deg)) + 3 * (14 + rest_fields) * 4


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_roundtrip_multiple_splats_and_sh_ordering(deg):
    """export.rs:305-349: export -> import keeps count, degree and every SH coefficient in place."""
    tr, sh, op = _splats(100, deg)
    d = ply.load_splat_from_ply(ply.splat_to_ply(tr, sh, op, render_mip=True, up_axis=(0.0, 0.0, 1.0)))
    assert d["meta"] == dict(up_axis=(0.0, 0.0, 1.0), render_mode="mip", total_splats=100, sh_degree=deg)
    assert np.array_equal(d["sh"], sh) and np.array_equal(d["raw_opac"], op)
    assert np.array_equal(d["transforms"][:, [0, 1, 2, 7, 8, 9]], tr[:, [0, 1, 2, 7, 8, 9]])
    q = d["transforms"][:, 3:7]
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)                    # normalised on export (export.rs:151-158)
    assert np.allclose(q, tr[:, 3:7] / np.linalg.norm(tr[:, 3:7], axis=1, keepdims=True), atol=1e-6)


def test_inria_rest_layout_is_channel_major():
    """f_rest_k = coefficient 1 + k % (C-1) of channel k // (C-1) (export.rs:91, import.rs:128-143)."""
    tr, sh, op = _splats(2, 2)
    r = ply.rows(tr, sh, op)
    per = 8
    for ch in range(3):
        for k in range(per):
            assert r[1, 14 + ch * per + k] == sh[1, 1 + k, ch]
    assert list(r[0, :3]) == list(tr[0, :3]) and list(r[0, 3:6]) == list(tr[0, 7:10]) and r[0, 6] == op[0]
    assert list(r[0, 11:14]) == list(sh[0, 0, :])


def test_floor_is_baked_on_export():
    """export.rs:183"""
    from oracle import bo
    tr, sh, op = _splats(50, 1)
    tr[:, 7:] = np.random.default_rng(1).uniform(-6, -2, (50, 3))
    f = np.full(50, 0.02, np.float32)
    d = ply.load_splat_from_ply(ply.splat_to_ply(tr, sh, op, min_scale=f))
    ft, fo = bo.fold_min_scale(tr, op, f)
    assert np.array_equal(d["transforms"][:, 7:], ft[:, 7:]) and np.array_equal(d["raw_opac"], fo)


def test_import_defaults_for_absent_properties():
    """import.rs:57-75: positions only -> rotation (1,0,0,0), log-scale -4, SH DC 0.5, raw opacity 0."""
    pts = np.arange(12, dtype="<f4").reshape(4, 3)
    data = b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nend_header\n" + pts.tobytes()
    d = ply.load_splat_from_ply(data)
    assert np.array_equal(d["transforms"][:, :3], pts)
    assert (d["transforms"][:, 3:7] == [1, 0, 0, 0]).all() and (d["transforms"][:, 7:] == -4.0).all()
    assert d["sh"].shape == (4, 1, 3) and (d["sh"] == 0.5).all() and (d["raw_opac"] == 0.0).all()


# ---- the product's host-side header parser (no GPU needed) -----------------------------------------
@pytest.mark.parametrize("deg", [0, 1, 3, 4])
def test_product_header_parser_reads_oracle_files(deg):
    import brush_amd as ba
    tr, sh, op = _splats(11, deg)
    for up, mip in ((None, False), ((0.25, -1.0, 0.0), True)):
        m = ba.ply_parse_header(ply.splat_to_ply(tr, sh, op, render_mip=mip, up_axis=up))
        assert m.total_splats == 11 and m.sh_degree == deg and m.render_mode == ("mip" if mip else "default")
        assert m.up_axis == ((0.0, -1.0, 0.0) if up is None else up)   # "y" means -Y (import.rs:203)


def test_product_header_parser_rejects_what_it_does_not_read():
    import brush_amd as ba
    ok = ply.splat_to_ply(*_splats(2, 0))
    with pytest.raises(ba.BrushHipError):
        ba.ply_parse_header(ok.replace(b"binary_little_endian", b"ascii"))
    with pytest.raises(ba.BrushHipError):
        ba.ply_parse_header(ok.replace(b"property float opacity", b"property uchar opacity"))
    with pytest.raises(ba.BrushHipError):
        ba.ply_parse_header(ok[:-8])                      # truncated body: "Unexpected EOF"
    with pytest.raises(ba.BrushHipError):
        ba.ply_parse_header(b"not a ply at all")
    with pytest.raises(ba.BrushHipError):                 # SuperSplat-compressed: chunk element first (import.rs:243-249)
        ba.ply_parse_header(b"ply\nformat binary_little_endian 1.0\nelement chunk 1\nproperty float min_x\nelement vertex 1\nproperty uint packed_position\nend_header\n" + b"\0" * 8)
