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


def load_splat_from_ply(data):
    """-> dict(transforms, sh, raw_opac, meta) for a binary_little_endian file of float vertex rows."""
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
            assert ty in ("float", "float32")
            props.append(name)
    col = {p: i for i, p in enumerate(props)}
    r = np.frombuffer(data, "<f4", count=n * len(props), offset=body).reshape(n, len(props))
    sh_count = sum(1 for p in props if p.startswith("f_dc_") or p.startswith("f_rest_"))
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
            sh[:, 0, ch] = get("f_dc_%d" % ch, 0.0)
        per = coeffs - 1
        for ch in range(3):          # interleave_coeffs: index = channel * per + coeff
            for k in range(per):
                sh[:, 1 + k, ch] = r[:, col["f_rest_%d" % (ch * per + k)]]
    return dict(transforms=tr, sh=sh, raw_opac=op, meta=dict(up_axis=up, render_mode=mode, total_splats=n, sh_degree=int(round(coeffs ** 0.5)) - 1))
