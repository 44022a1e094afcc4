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
        ba.ply_parse_header(ok.replace(b"property float opacity", b"property list uchar int opacity"))   # scalar properties only
    with pytest.raises(ba.BrushHipError):
        ba.ply_parse_header(ok[:-8])                      # truncated body: "Unexpected EOF"
    with pytest.raises(ba.BrushHipError):
        ba.ply_parse_header(b"not a ply at all")
    with pytest.raises(ba.BrushHipError):                 # SuperSplat-compressed (import.rs:243-249) without the chunk ranges
        ba.ply_parse_header(b"ply\nformat binary_little_endian 1.0\nelement chunk 1\nproperty float min_x\nelement vertex 1\nproperty uint packed_position\nend_header\n" + b"\0" * 8)


# ---- SuperSplat-compressed files (import.rs:407-600, quant.rs) ---------------------------------------------------------
def _pack_11_10_11(x, y, z):
    return (np.uint32(x) << 21) | (np.uint32(y) << 11) | np.uint32(z)


def test_compressed_bit_layouts_known_answers():
    """quant.rs:9-70 by hand: extreme and mid codes of each packed word."""
    v = ply._vec_11_10_11(np.array([_pack_11_10_11(2047, 0, 2047), _pack_11_10_11(0, 1023, 0), _pack_11_10_11(1, 1, 1)], np.uint32))
    assert np.array_equal(v[0], [1, 0, 1]) and np.array_equal(v[1], [0, 1, 0])
    assert v[2, 0] == np.float32(1) / np.float32(2047) and v[2, 1] == np.float32(1) / np.float32(1023)
    c = ply._vec_8_8_8_8(np.array([0xFF000080, 0x00FF00FF], np.uint32))
    assert np.array_equal(c[0], np.array([1, 0, 0, np.float32(128) / np.float32(255)], np.float32)) and np.array_equal(c[1], [0, 1, 0, 1])
    # the dropped component is rebuilt as sqrt(1 - a^2 - b^2 - c^2): a = b = c = code 512 ~ +0.00069 -> identity-like quaternion
    for largest in range(4):
        q = ply._quat_wxyz(np.array([(largest << 30) | (512 << 20) | (512 << 10) | 512], np.uint32))[0]
        assert abs(q[largest] - 1.0) < 1e-5 and np.all(np.abs(np.delete(q, largest)) < 1e-3)
        assert abs(np.linalg.norm(q) - 1.0) < 1e-6
    # extreme code: (1 - 0.5) / (0.5 sqrt 2) = 1 / sqrt 2
    q = ply._quat_wxyz(np.array([(0 << 30) | (1023 << 20) | (512 << 10) | 0], np.uint32))[0]
    assert abs(q[1] - 2 ** -0.5) < 1e-6 and abs(q[3] + 2 ** -0.5) < 1e-6


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_compressed_import_is_within_chunk_ranges(deg):
    """means / log-scales are lerps inside their chunk's [min, max]; the SH DC term inverts rgb_to_sh; f_rest bytes map to
    ((b / 254) - 0.5) * 8 in [channel][coeff] order whatever the property order in the header (serde looks fields up by name)."""
    n = 700
    data = ply.make_compressed_ply(n, deg, seed=deg)
    d = ply.load_compressed_ply(data)
    c = (deg + 1) ** 2
    assert d["transforms"].shape == (n, 10) and d["sh"].shape == (n, c, 3) and d["raw_opac"].shape == (n,)
    elems, body, _ = ply._parse_elements(data)
    nch = elems[0]["count"]
    assert nch == 3
    chunks = np.frombuffer(data, "<f4", count=nch * 18, offset=body).reshape(nch, 18)
    names = [p for _, p in elems[0]["props"]]
    lo = chunks[:, [names.index(k) for k in ("min_x", "min_y", "min_z")]][np.arange(n) // 256]
    hi = chunks[:, [names.index(k) for k in ("max_x", "max_y", "max_z")]][np.arange(n) // 256]
    assert np.all(d["transforms"][:, :3] >= lo - 1e-5) and np.all(d["transforms"][:, :3] <= hi + 1e-5)
    fin = np.isfinite(d["raw_opac"])
    assert (~fin).sum() == max(1, n // 40)                      # alpha bytes 0 / 255 -> -inf / +inf, as the reference's inverse_sigmoid gives
    a = np.frombuffer(data, "<u4", count=n * 4, offset=body + nch * 72).reshape(n, 4)[:, 3] & 0xFF
    assert np.allclose(1 / (1 + np.exp(-d["raw_opac"][fin].astype(np.float64))), a[fin] / 255.0, atol=1e-6)
    if deg:
        k = 3 * (c - 1)
        shb = np.frombuffer(data, np.uint8, count=n * k, offset=body + nch * 72 + n * 16).reshape(n, k)
        order = [int(p.split("_")[-1]) for _, p in elems[2]["props"]]
        for col, idx in enumerate(order):
            ch, i = divmod(idx, c - 1)
            assert np.array_equal(d["sh"][:, 1 + i, ch], ((shb[:, col].astype(np.float32) / np.float32(254)) - np.float32(0.5)) * np.float32(8))


def test_compressed_import_finds_chunk_and_vertex_fields_by_name():
    base = ply.load_compressed_ply(ply.make_compressed_ply(600, 1, seed=4))
    perm = ply.load_compressed_ply(ply.make_compressed_ply(600, 1, seed=4, chunk_order=list(reversed(ply.CHUNK_PROPS)),
                                                         vertex_order=list(reversed(ply.VERTEX_PROPS)), extra_chunk_props=("unused_a",)))
    for k in ("transforms", "sh", "raw_opac"):
        assert np.array_equal(base[k], perm[k], equal_nan=True)


def test_compressed_import_requires_every_chunk_field():
    """QuantMeta has no serde defaults (import.rs:416-436): a pre-colour-range file is an error in the reference too."""
    with pytest.raises(ValueError):
        ply.load_compressed_ply(ply.make_compressed_ply(300, 0, legacy_no_colour_range=True))


@pytest.mark.parametrize("deg", [0, 2, 3])
def test_header_parser_reads_compressed_files(deg):
    """bh_ply_parse_header (host code): element / property bookkeeping of a chunk + vertex [+ sh] file."""
    import brush_amd as ba
    data = ply.make_compressed_ply(1000, deg, seed=deg)
    meta = ba.ply_parse_header(data)
    assert meta.compressed and meta.total_splats == 1000 and meta.sh_degree == deg
    assert not ba.ply_parse_header(ply.splat_to_ply(*_splats(5, 1))).compressed
    with pytest.raises(ba.BrushHipError):
        ba.ply_parse_header(data[:-1])                          # "Unexpected EOF"
    with pytest.raises(ba.BrushHipError):
        ba.ply_parse_header(ply.make_compressed_ply(300, 0, legacy_no_colour_range=True))
    short = ply.make_compressed_ply(1000, 0).replace(b"element chunk 4", b"element chunk 3")
    with pytest.raises(ba.BrushHipError):
        ba.ply_parse_header(short)                              # fewer chunks than ceil(n / 256)


# ---- vertex rows that are not all float; the colour override (ply_gaussian.rs:36-99, import.rs:349-358) ------------------
def _mixed_ply(n, seed=0, colour="uchar", names=("red", "green", "blue"), with_sh=False, extra=True):
    """A point-cloud style file: double x, float y z, short / uchar extras, optional quantised colours."""
    rng = np.random.default_rng(seed)
    fields = [("x", "double", "<f8"), ("y", "float", "<f4"), ("z", "float", "<f4")]
    if extra:
        fields += [("nx", "short", "<i2"), ("scale_0", "float", "<f4"), ("scale_1", "char", "i1"), ("scale_2", "int", "<i4"), ("opacity", "uchar", "u1"),
                   ("rot_0", "ushort", "<u2"), ("rot_1", "uint", "<u4"), ("rot_2", "float", "<f4"), ("rot_3", "float", "<f4")]
    if colour:
        dt = {"uchar": "u1", "ushort": "<u2", "float": "<f4", "int": "<i4"}[colour]
        fields += [(nm, colour, dt) for nm in names]
    if with_sh:
        fields += [("f_dc_%d" % k, "float", "<f4") for k in range(3)]
    rec = np.zeros(n, np.dtype([(nm, dt) for nm, _, dt in fields]))
    for nm, _, dt in fields:
        if dt in ("<f4", "<f8"):
            rec[nm] = rng.normal(size=n)
        else:
            info = np.iinfo(np.dtype(dt))
            rec[nm] = rng.integers(max(info.min, -1000), min(info.max, 70000) + 1, n)
    head = ["ply", "format binary_little_endian 1.0", "element vertex %d" % n] + ["property %s %s" % (ty, nm) for nm, ty, _ in fields] + ["end_header"]
    return ("\n".join(head) + "\n").encode() + rec.tobytes(), rec


@pytest.mark.parametrize("colour,scale", [("uchar", 254.0), ("ushort", 65534.0), ("float", None)])
def test_mixed_rows_and_colour_override(colour, scale):
    data, rec = _mixed_ply(300, seed=3, colour=colour, names=("r", "green", "b"))
    d = ply.load_splat_from_ply(data)
    assert d["meta"]["total_splats"] == 300 and d["meta"]["sh_degree"] == 0 and d["sh"].shape == (300, 1, 3)
    assert np.array_equal(d["transforms"][:, 0], rec["x"].astype(np.float32)) and np.array_equal(d["transforms"][:, 8], rec["scale_1"].astype(np.float32))
    assert np.array_equal(d["transforms"][:, 3], rec["rot_0"].astype(np.float32)) and np.array_equal(d["raw_opac"], rec["opacity"].astype(np.float32))
    for ch, nm in enumerate(("r", "green", "b")):
        v = rec[nm].astype(np.float32) / np.float32(scale) if scale else rec[nm]
        assert np.array_equal(d["sh"][:, 0, ch], (v - np.float32(0.5)) / np.float32(0.2820948))


def test_colour_override_error_cases():
    with pytest.raises(ValueError):
        ply.load_splat_from_ply(_mixed_ply(10, colour="int")[0])                     # de_quant takes f32 / u8 / u16 only
    with pytest.raises(ValueError):
        ply.load_splat_from_ply(_mixed_ply(10, colour="uchar", with_sh=True)[0])      # 1 + (6 - 3) / 3 = 2 coefficients: not a square
    with pytest.raises(ValueError):
        ply.load_splat_from_ply(_mixed_ply(10, colour="uchar", names=("red", "r", "blue"))[0])


def test_header_parser_accepts_mixed_rows():
    import brush_amd as ba
    meta = ba.ply_parse_header(_mixed_ply(50, colour="uchar")[0])
    assert meta.total_splats == 50 and meta.sh_degree == 0 and not meta.compressed
    for bad in (_mixed_ply(10, colour="int")[0], _mixed_ply(10, colour="uchar", with_sh=True)[0], _mixed_ply(10, colour="uchar", names=("red", "r", "blue"))[0],
                _mixed_ply(10, colour="uchar")[0][:-1]):
        with pytest.raises(ba.BrushHipError):
            ba.ply_parse_header(bad)
    assert ba.ply_parse_header(_mixed_ply(10, colour=None, extra=False)[0]).total_splats == 10   # a bare x y z cloud


def test_header_parser_survives_mutated_headers():
    """Host-side robustness: random byte flips, deletions, insertions and truncations of three kinds of header either parse
    (and then the declared body fits the buffer: the parser checks it) or are rejected — never a crash."""
    import brush_amd as ba
    rng = np.random.default_rng(0)
    bases = [ply.make_compressed_ply(600, 2, seed=3), ply.splat_to_ply(*_splats(50, 3)), _mixed_ply(80, colour="uchar")[0]]
    parsed = rejected = 0
    for it in range(1500):
        b = bytearray(bases[it % 3])
        hdr_end = b.index(b"end_header") + 11
        for _ in range(int(rng.integers(1, 6))):
            mode, pos = int(rng.integers(0, 4)), int(rng.integers(0, max(hdr_end, 1)))
            if mode == 0:
                b[pos] = int(rng.integers(0, 256))
            elif mode == 1:
                del b[pos:pos + int(rng.integers(1, 12))]
            elif mode == 2:
                b[pos:pos] = bytes(rng.integers(32, 127, int(rng.integers(1, 10)), dtype=np.uint8))
            else:
                b = b[:int(rng.integers(0, len(b) + 1))]
            hdr_end = min(hdr_end, len(b))
            if hdr_end <= 1:
                break
        try:
            meta = ba.ply_parse_header(bytes(b))
            parsed += 1
            assert meta.total_splats >= 0 and 0 <= meta.sh_degree <= 4
        except ba.BrushHipError:
            rejected += 1
    assert parsed > 0 and rejected > 1000
