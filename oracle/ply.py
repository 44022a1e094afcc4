"""Oracle restatement (numpy) of Brush's PLY export / import for plain INRIA-layout files.
TEST INFRASTRUCTURE ONLY.

export: brush-serde/src/export.rs:86-204 (read_splat_data + splat_to_ply); the container (ASCII header,
little-endian `property float` rows) is serde_ply's (un-vendored dependency) binary_le output.
import: brush-serde/src/import.rs:172-330 (parse_ply) + :57-75 (SplatData::into_splats defaults) +
:128-143 (interleave_coeffs)."""
import numpy as np


def _f32_display(v):
    """Rust `{}` for f32: shortest representation that round-trips, never exponent notation."""
    v = np.float32(v)
    if np.isnan(v):
        return "NaN"
    if np.isinf(v):
        return "-inf" if v < 0 else "inf"
    return np.format_float_positional(v, unique=True, trim="-")


def header(n, sh_degree, render_mip=False, up_axis=None):
    lines = ["ply", "format binary_little_endian 1.0", "comment Exported from Brush"]
    if up_axis is not None:
        lines.append("comment Vertical axis: %s %s %s" % tuple(_f32_display(x) for x in up_axis))  # export.rs:189-191
    else:
        lines.append("comment Vertical axis: y")
    lines.append("comment SH degree: %d" % sh_degree)
    lines.append("comment SplatRenderMode: %s" % ("mip" if render_mip else "default"))
    lines.append("element vertex %d" % n)
    for p in ["x", "y", "z", "scale_0", "scale_1", "scale_2", "opacity", "rot_0", "rot_1", "rot_2", "rot_3", "f_dc_0", "f_dc_1", "f_dc_2"]:
        lines.append("property float " + p)
    for k in range(3 * ((sh_degree + 1) ** 2 - 1)):
        lines.append("property float f_rest_%d" % k)
    lines.append("end_header")
    return ("\n".join(lines) + "\n").encode("ascii")


def rows(transforms, sh, raw_opac):
    """export.rs:117-176: [N, 14 + 3(C-1)] f32."""
    t = np.ascontiguousarray(transforms, np.float32).reshape(-1, 10)
    n = t.shape[0]
    s = np.ascontiguousarray(sh, np.float32).reshape(n, -1, 3)
    o = np.ascontiguousarray(raw_opac, np.float32).reshape(n)
    q = t[:, 3:7]
    ss = ((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]) + q[:, 3] * q[:, 3]
    rn = np.maximum(np.sqrt(ss, dtype=np.float32), np.float32(1e-12))
    rest = s[:, 1:, :].transpose(0, 2, 1).reshape(n, -1)  # [n, channel, coeff]: the permute([0, 2, 1]) of export.rs:91
    return np.concatenate([t[:, 0:3], t[:, 7:10], o[:, None], (q / rn[:, None]).astype(np.float32), s[:, 0, :], rest], axis=1).astype(np.float32)


def splat_to_ply(transforms, sh, raw_opac, render_mip=False, up_axis=None, min_scale=None):
    if min_scale is not None:  # export.rs:183 bake_min_scale
        from oracle import bo
        transforms, raw_opac = bo.fold_min_scale(transforms, raw_opac, min_scale)
    n = np.asarray(transforms).reshape(-1, 10).shape[0]
    coeffs = np.asarray(sh).reshape(n, -1, 3).shape[1] if n else max(int(np.asarray(sh).size // 3), 1)
    deg = int(round(coeffs ** 0.5)) - 1
    r = rows(transforms, sh, raw_opac)
    return header(n, deg, render_mip, up_axis) + r.astype("<f4").tobytes()


_PLY_DTYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1",
               "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}
_RGB_NAMES = {"red": 0, "r": 0, "green": 1, "g": 1, "blue": 2, "b": 2}


def load_splat_from_ply(data):
    """-> dict(transforms, sh, raw_opac, meta) for a binary_little_endian file of scalar vertex rows.
    Plain fields are f32 in the reference's row struct: serde casts whatever scalar type the file holds (`as f32`).  The colour
    override fields red / green / blue (aliases r / g / b) go through de_quant (ply_gaussian.rs:36-58): f32 as is, u8 / 254,
    u16 / 65534, anything else an error; when all three are present the DC term is rgb_to_sh of them (import.rs:349-358)."""
    end = data.index(b"end_header")
    body = end + len(b"end_header")
    if data[body:body + 1] == b"\r":
        body += 1
    if data[body:body + 1] == b"\n":
        body += 1
    lines = [ln.strip() for ln in data[:end].decode("ascii", "replace").split("\n")]
    assert lines[0] == "ply"
    props, n, in_vertex = [], 0, False
    up, mode = None, None
    for ln in lines[1:]:
        if ln.startswith("comment"):
            c = ln[7:].strip().lower()
            if c.startswith("vertical axis: "):
                s = c[len("vertical axis: "):].strip()
                if s == "x":
                    up = (1.0, 0.0, 0.0)
                elif s == "y":
                    up = (0.0, -1.0, 0.0)
                elif s == "z":
                    up = (0.0, 0.0, -1.0)
                else:
                    parts = []
                    for tok in s.replace(",", " ").replace("[", " ").replace("]", " ").split():
                        try:
                            parts.append(float(tok))
                        except ValueError:
                            pass
                    if len(parts) == 3:
                        up = tuple(parts)
            elif c.startswith("splatrendermode: "):
                s = c[len("splatrendermode: "):].strip()
                mode = {"mip": "mip", "default": "default"}.get(s, mode)
        elif ln.startswith("element "):
            _, name, cnt = ln.split()
            in_vertex = name == "vertex"
            if in_vertex:
                n = int(cnt)
        elif ln.startswith("property ") and in_vertex:
            _, ty, name = ln.split()
            props.append((name, ty))
    rec = np.frombuffer(data, np.dtype([(nm, _PLY_DTYPES[ty]) for nm, ty in props]), count=n, offset=body)
    types = dict(props)
    props = [nm for nm, _ in props]
    col = {p: i for i, p in enumerate(props)}

    class _Rows:   # r[:, col[name]] -> the column cast to f32 (serde's integer / f64 -> f32 `as` cast)
        def __getitem__(self, key):
            return rec[props[key[1]]].astype(np.float32)
    r = _Rows()
    rgb = [None, None, None]
    for nm in props:
        if nm in _RGB_NAMES:
            if rgb[_RGB_NAMES[nm]] is not None:
                raise ValueError("duplicate field " + nm)
            ty = _PLY_DTYPES[types[nm]]
            if ty == "<f4":
                rgb[_RGB_NAMES[nm]] = rec[nm].astype(np.float32)
            elif ty == "u1":
                rgb[_RGB_NAMES[nm]] = rec[nm].astype(np.float32) / np.float32(254.0)
            elif ty == "<u2":
                rgb[_RGB_NAMES[nm]] = rec[nm].astype(np.float32) / np.float32(65534.0)
            else:
                raise ValueError("a quantized value or a float expected for " + nm)
    n_rgb = sum(v is not None for v in rgb)
    sh_count = sum(1 for p in props if p.startswith("f_dc_") or p.startswith("f_rest_")) + n_rgb   # import.rs:317-325
    if n_rgb not in (0, 3) or (n_rgb == 3 and sh_count != 3):
        raise ValueError("colour override next to SH properties: the reference's coefficient count is not a square")
    coeffs = max(sh_count // 3, 1)

    def get(name, default):
        return r[:, col[name]] if name in col else np.full(n, default, np.float32)
    tr = np.empty((n, 10), np.float32)
    for k, nm in enumerate("xyz"):
        tr[:, k] = r[:, col[nm]]
    has_rot, has_scale = "rot_0" in col, "scale_0" in col
    for k in range(4):
        tr[:, 3 + k] = get("rot_%d" % k, 0.0) if has_rot else (1.0 if k == 0 else 0.0)
    for k in range(3):
        tr[:, 7 + k] = get("scale_%d" % k, 0.0) if has_scale else -4.0
    op = get("opacity", 0.0).astype(np.float32).copy()
    sh = np.empty((n, coeffs, 3), np.float32)
    if sh_count == 0:
        sh[:] = 0.5
    else:
        for ch in range(3):
            sh[:, 0, ch] = get("f_dc_%d" % ch, 0.0) if n_rgb == 0 else ((rgb[ch] - np.float32(0.5)) / SH_C0).astype(np.float32)
        per = coeffs - 1
        for ch in range(3):          # interleave_coeffs: index = channel * per + coeff
            for k in range(per):
                sh[:, 1 + k, ch] = r[:, col["f_rest_%d" % (ch * per + k)]]
    return dict(transforms=tr, sh=sh, raw_opac=op, meta=dict(up_axis=up, render_mode=mode, total_splats=n, sh_degree=int(round(coeffs ** 0.5)) - 1))


# ---------------------------------------------------------------------------
# SuperSplat / PlayCanvas "compressed.ply": brush-serde/src/import.rs:407-600 (parse_compressed_ply),
# quant.rs:1-75 (bit layouts), ply_gaussian.rs:24-33,105-119 (QuantSplat / QuantSh).
# Elements: `chunk` (per 256 splats: min / max of position, log-scale, colour — floats, looked up by NAME),
# `vertex` (four packed u32: packed_position 11-10-11, packed_rotation 2-10-10-10, packed_scale 11-10-11,
# packed_color 8-8-8-8), optional `sh` (uchar f_rest_k, [channel][coeff] order).
# ---------------------------------------------------------------------------
CHUNK_PROPS = ["min_x", "min_y", "min_z", "max_x", "max_y", "max_z", "min_scale_x", "min_scale_y", "min_scale_z",
               "max_scale_x", "max_scale_y", "max_scale_z", "min_r", "min_g", "min_b", "max_r", "max_g", "max_b"]
VERTEX_PROPS = ["packed_position", "packed_rotation", "packed_scale", "packed_color"]
F = np.float32
SH_C0 = F(0.2820948)


def _unorm(p, bits):                                   # quant.rs:4-7
    return p.astype(F) / F((1 << bits) - 1)


def _vec_11_10_11(v):                                  # quant.rs:9-18
    v = v.astype(np.uint32)
    return np.stack([_unorm((v >> 21) & 0x7FF, 11), _unorm((v >> 11) & 0x3FF, 10), _unorm(v & 0x7FF, 11)], axis=1)


def _vec_8_8_8_8(v):                                   # quant.rs:20-36
    v = v.astype(np.uint32)
    return np.stack([_unorm((v >> 24) & 0xFF, 8), _unorm((v >> 16) & 0xFF, 8), _unorm((v >> 8) & 0xFF, 8), _unorm(v & 0xFF, 8)], axis=1)


def _quat_wxyz(v):                                     # quant.rs:38-70 -> scalar order (w, x, y, z), import.rs:511-517
    v = v.astype(np.uint32)
    largest = ((v >> 30) & 3).astype(np.int64)
    norm = F(0.5) * F(1.4142135623730951)
    with np.errstate(invalid="ignore"):
        abc = [((_unorm((v >> s) & 0x3FF, 10) - F(0.5)) / norm).astype(F) for s in (20, 10, 0)]
        m = np.sqrt(F(1.0) - ((abc[0] * abc[0] + abc[1] * abc[1]) + abc[2] * abc[2]), dtype=F)
    q = np.zeros((v.size, 4), F)
    for i in range(v.size):   # small (test-sized) inputs only
        ind = 0
        for k in range(4):
            if k == largest[i]:
                q[i, k] = m[i]
            else:
                q[i, k] = abc[ind][i]
                ind += 1
    return q   # quat[0..3] = (w, x, y, z)


def _parse_elements(data):
    end = data.index(b"end_header")
    body = end + len(b"end_header")
    if data[body:body + 1] == b"\r":
        body += 1
    if data[body:body + 1] == b"\n":
        body += 1
    elems, cur = [], None
    meta = dict(up_axis=None, render_mode=None)
    for ln in [x.strip() for x in data[:end].decode("ascii", "replace").split("\n")][1:]:
        if ln.startswith("element "):
            _, name, cnt = ln.split()
            cur = dict(name=name, count=int(cnt), props=[])
            elems.append(cur)
        elif ln.startswith("property ") and cur is not None:
            _, ty, name = ln.split()
            cur["props"].append((ty, name))
    return elems, body, meta


_SIZES = {"float": 4, "float32": 4, "uint": 4, "uint32": 4, "int": 4, "int32": 4, "uchar": 1, "uint8": 1}


def load_compressed_ply(data):
    """-> dict(transforms [N,10], sh [N,C,3], raw_opac [N]) for a SuperSplat-compressed file (subsample = 1)."""
    elems, off, _ = _parse_elements(data)
    assert elems[0]["name"] == "chunk" and elems[1]["name"] == "vertex"
    raw = {}
    for e in elems:
        stride = sum(_SIZES[t] for t, _ in e["props"])
        raw[e["name"]] = (np.frombuffer(data, np.uint8, count=e["count"] * stride, offset=off).reshape(e["count"], stride), e)
        off += e["count"] * stride

    def col(name, prop, dtype):
        arr, e = raw[name]
        o = 0
        for t, p in e["props"]:
            if p == prop:
                return arr[:, o:o + _SIZES[t]].copy().view(dtype).reshape(-1)
            o += _SIZES[t]
        return None
    n = raw["vertex"][1]["count"]
    ch = np.arange(n) // 256                                                       # import.rs:503
    cm = {p: col("chunk", p, "<f4") for p in CHUNK_PROPS}
    for p, v in cm.items():                                                        # QuantMeta's fields are not optional (import.rs:416-436)
        if v is None:
            raise ValueError("compressed PLY: chunk property '%s' is missing" % p)

    def lerp(rawv, lo, hi):                                                        # raw * (max - min) + min, import.rs:435-451
        lo = np.stack([cm[k][ch] for k in lo], axis=1).astype(F)
        hi = np.stack([cm[k][ch] for k in hi], axis=1).astype(F)
        return (rawv * (hi - lo) + lo).astype(F)
    pos = _vec_11_10_11(col("vertex", "packed_position", "<u4"))
    scl = _vec_11_10_11(col("vertex", "packed_scale", "<u4"))
    rgba = _vec_8_8_8_8(col("vertex", "packed_color", "<u4"))
    tr = np.empty((n, 10), F)
    tr[:, 0:3] = lerp(pos, ("min_x", "min_y", "min_z"), ("max_x", "max_y", "max_z"))
    tr[:, 3:7] = _quat_wxyz(col("vertex", "packed_rotation", "<u4"))
    tr[:, 7:10] = lerp(scl, ("min_scale_x", "min_scale_y", "min_scale_z"), ("max_scale_x", "max_scale_y", "max_scale_z"))
    with np.errstate(divide="ignore", invalid="ignore"):
        a = rgba[:, 3]
        op = np.log(a / (F(1.0) - a), dtype=F).astype(F)                           # inverse_sigmoid, gaussian_splats.rs:76-78
    dc = ((lerp(rgba[:, :3], ("min_r", "min_g", "min_b"), ("max_r", "max_g", "max_b")) - F(0.5)) / SH_C0).astype(F)   # rgb_to_sh, sh.rs:21-31
    if "sh" in raw:
        arr, e = raw["sh"]
        k = len(e["props"])
        per = k // 3
        names = [p for _, p in e["props"]]
        rest = np.zeros((n, k), F)
        for j, nm in enumerate(names):                                             # f_rest_<idx>: looked up by name
            idx = int(nm.split("_")[-1])
            rest[:, idx] = ((arr[:, j].astype(F) / F(254.0)) - F(0.5)) * F(8.0)    # de_quant_sh, ply_gaussian.rs:105-111
        sh = np.empty((n, per + 1, 3), F)
        sh[:, 0, :] = dc
        for c3 in range(3):                                                        # interleave_coeffs, import.rs:133-144
            for i in range(per):
                sh[:, 1 + i, c3] = rest[:, c3 * per + i]
    else:
        sh = dc.reshape(n, 1, 3)
    return dict(transforms=tr, sh=np.ascontiguousarray(sh), raw_opac=op)


def make_compressed_ply(n, sh_degree, seed=0, chunk_order=None, vertex_order=None, extra_chunk_props=(), legacy_no_colour_range=False):
    """TEST-ONLY synthesiser of a compressed file with random packed words (every bit pattern is a legal input)."""
    rng = np.random.default_rng(seed)
    nch = (n + 255) // 256
    cprops = list(chunk_order or CHUNK_PROPS)
    if legacy_no_colour_range:
        cprops = [p for p in cprops if not p.endswith(("_r", "_g", "_b"))]
    cprops += list(extra_chunk_props)
    vprops = list(vertex_order or VERTEX_PROPS)
    lo = rng.uniform(-3, 0, (nch, 9)).astype(F)
    hi = lo + rng.uniform(0.1, 4, (nch, 9)).astype(F)
    vals = {"min_x": lo[:, 0], "min_y": lo[:, 1], "min_z": lo[:, 2], "max_x": hi[:, 0], "max_y": hi[:, 1], "max_z": hi[:, 2],
            "min_scale_x": lo[:, 3] - 4, "min_scale_y": lo[:, 4] - 4, "min_scale_z": lo[:, 5] - 4,
            "max_scale_x": hi[:, 3] - 4, "max_scale_y": hi[:, 4] - 4, "max_scale_z": hi[:, 5] - 4,
            "min_r": lo[:, 6] * F(0.1), "min_g": lo[:, 7] * F(0.1), "min_b": lo[:, 8] * F(0.1),
            "max_r": F(1) + hi[:, 6] * F(0.1), "max_g": F(1) + hi[:, 7] * F(0.1), "max_b": F(1) + hi[:, 8] * F(0.1)}
    chunk_rows = np.stack([vals.get(p, np.zeros(nch, F)).astype(F) for p in cprops], axis=1)
    words = {p: rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for p in VERTEX_PROPS}
    # keep most quaternions real (a^2 + b^2 + c^2 <= 1) and the alpha byte away from 0 / 255 for most splats
    abc = rng.integers(160, 864, (n, 3)).astype(np.uint32)
    words["packed_rotation"] = (rng.integers(0, 4, n).astype(np.uint32) << 30) | (abc[:, 0] << 20) | (abc[:, 1] << 10) | abc[:, 2]
    words["packed_rotation"][: max(1, n // 50)] = rng.integers(0, 2 ** 32, max(1, n // 50), dtype=np.uint64).astype(np.uint32)
    alpha = rng.integers(1, 255, n).astype(np.uint32)
    alpha[: max(1, n // 40)] = rng.choice(np.array([0, 255], np.uint32), max(1, n // 40))
    words["packed_color"] = (words["packed_color"] & np.uint32(0xFFFFFF00)) | alpha
    vert_rows = np.stack([words[p] for p in vprops], axis=1)
    k = 3 * ((sh_degree + 1) ** 2 - 1)
    head = ["ply", "format binary_little_endian 1.0", "comment compressed test file", "element chunk %d" % nch]
    head += ["property float " + p for p in cprops]
    head += ["element vertex %d" % n] + ["property uint " + p for p in vprops]
    body = chunk_rows.astype("<f4").tobytes() + vert_rows.astype("<u4").tobytes()
    if k:
        order = rng.permutation(k) if seed % 2 else np.arange(k)
        head += ["element sh %d" % n] + ["property uchar f_rest_%d" % i for i in order]
        body += rng.integers(0, 256, (n, k), dtype=np.uint8).tobytes()
    head.append("end_header")
    return ("\n".join(head) + "\n").encode("ascii") + body
