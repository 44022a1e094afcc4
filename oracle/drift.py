"""How far is the shipped numerical specification from the reference's #[cube] sources taken literally?

TEST INFRASTRUCTURE (tests/test_oracle_literal_drift.py, scripts/literal_drift.py).  The specification build of the
oracle (libbrush_oracle.so: fixed exp / ln / atan2 polynomials, fma in calc_sigma and in the blend's colour sums,
exp_blend) and the literal builds (BO_LITERAL = 1, 2: libm functions, the expressions of brush-cube/src/lib.rs:561-578
and kernels/rasterize.rs:129-166 as written) render the same scene; this module reports the differences the
north-star's "within 1e-4 L-inf of the WGPU reference, tile assignment bit-exact" is about:
image L-inf, gradient relative L-inf, and the COUNT of differing (tile, splat) assignments / visible flags / shrunk
list ends.
"""
import numpy as np

from . import bo


def _pairs(r):
    """the (tile, splat id) assignments of a render as sorted u64 keys"""
    tid = r.get("tile_id_from_isect").astype(np.uint64)
    gid = r.get("global_from_compact_gid")[r.get("compact_gid_from_isect")].astype(np.uint64)
    return np.sort((tid << np.uint64(32)) | gid)


def render(scene, cam_params, variant, v_output=None, bg=(0.1, 0.2, 0.3), flags=bo.FLAG_BWD_INFO):
    r = bo.Render(variant).forward(bo.camera(**cam_params), scene["transforms"], scene["sh"], scene["raw_opac"], bg=bg, flags=flags)
    if v_output is not None:
        r.backward(v_output)
    return r


def measure(scene, cam_params, variant, v_output=None, bg=(0.1, 0.2, 0.3), flags=bo.FLAG_BWD_INFO, spec=None):
    """Render `scene` with the specification (or take `spec`, a finished render()) and with `variant`; returns the drift figures."""
    rs = spec if spec is not None else render(scene, cam_params, "spec", v_output, bg, flags)
    rl = render(scene, cam_params, variant, v_output, bg, flags)
    return compare(rs, rl, v_output is not None)


def compare(rs, rl, with_grads):
    out = {}
    out["num_visible"] = (int(rs.num_visible), int(rl.num_visible))
    out["num_intersections"] = (int(rs.num_intersections), int(rl.num_intersections))
    ia, ib = rs.image(), rl.image()
    dimg = np.abs(ia - ib)
    out["image_linf"] = float(dimg.max())
    out["image_values"] = int(dimg.size)
    # a blend decision (alpha >= 1/255, T' <= 1e-4: kernels/rasterize.rs:137-146) that falls the other way moves ONE pixel by up
    # to alpha*T*colour — a discontinuity of the reference's own rule, present between any two executions of it
    out["image_values_above_1e-5"] = int(np.count_nonzero(dimg > 1e-5))
    out["image_values_above_1e-4"] = int(np.count_nonzero(dimg > 1e-4))
    out["image_p99999"] = float(np.quantile(dimg, 0.99999))
    # the reference's own image tolerance (crates/brush-bench-test/src/reference.rs:50-51: atol 1e-5 + rtol 1e-2)
    out["image_outside_reference_tolerance"] = int(np.count_nonzero(dimg > 1e-5 + 1e-2 * np.abs(ib)))
    ps, pl = _pairs(rs), _pairs(rl)
    out["assignments_differing"] = int(np.setxor1d(ps, pl, assume_unique=True).size)
    out["visible_flags_differing"] = int(np.count_nonzero(rs.get("visible") != rl.get("visible")))
    # shrunk list ends: compare each tile's LAST useful splat (by splat id: list positions shift when an assignment differs)
    def last_useful(r):
        to = r.get("tile_offsets").reshape(-1, 2).astype(np.int64)
        g = r.get("global_from_compact_gid")[r.get("compact_gid_from_isect")].astype(np.int64)
        has = to[:, 1] > to[:, 0]
        return np.where(has, g[np.maximum(to[:, 1] - 1, 0)] if g.size else 0, -1)
    out["shrunk_ends_differing"] = int(np.count_nonzero(last_useful(rs) != last_useful(rl)))
    out["num_tiles"] = int(rs.num_tiles)
    if with_grads:
        for k in ("v_transforms", "v_coeffs", "v_raw_opac", "v_refine"):
            a, b = rs.get(k), rl.get(k)
            scale = max(float(np.abs(b).max()), 1e-30)
            out["grad_rel_linf_" + k] = float(np.abs(a - b).max() / scale)
    return out
