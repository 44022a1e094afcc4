"""ctypes loader for the CPU oracle (oracle/brush_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. Nothing under brush_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "libbrush_oracle.so")

FLAG_MIP = 1
FLAG_BWD_INFO = 2
FLAG_SMOOTH_CUTOFF = 4


class BoCamera(C.Structure):
    _fields_ = [
        ("vm", C.c_float * 12),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("lim_pos_x", C.c_float), ("lim_pos_y", C.c_float),
        ("lim_neg_x", C.c_float), ("lim_neg_y", C.c_float),
        ("cam_pos", C.c_float * 3),
        ("img_w", C.c_uint32), ("img_h", C.c_uint32),
        ("model", C.c_uint32), ("dist", C.c_float * 8), ("half_max_render_fov", C.c_float),
    ]


# kernels/camera_model/mod.rs:31-38
CAM_PINHOLE, CAM_KB4, CAM_RT8, CAM_TPF = 0, 1, 2, 3
_MODEL_IDS = {"pinhole": CAM_PINHOLE, "kb4": CAM_KB4, "rt8": CAM_RT8, "tpf": CAM_TPF}


def _model_args(model, dist):
    m = _MODEL_IDS[model] if isinstance(model, str) else int(model)
    d = np.zeros(8, np.float32)
    if dist is not None:
        dd = np.asarray(dist, np.float32).reshape(-1)
        d[: dd.size] = dd
    return m, d


def build(force=False):
    src = os.path.join(_DIR, "brush_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-s", "-B"])
    return _LIB_PATH


_libs = {}
# "spec": the numerical specification (what the HIP kernels are compared with).  "literal1" / "literal2": the #[cube] sources
# as written (BO_LITERAL in brush_oracle.cpp) — only for measuring the specification's drift against them.
VARIANTS = {"spec": "libbrush_oracle.so", "literal1": "libbrush_oracle_literal1.so", "literal2": "libbrush_oracle_literal2.so"}


def lib(variant="spec"):
    if variant not in _libs:
        path = os.path.join(_DIR, VARIANTS[variant])
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_DIR, "brush_oracle.cpp")):
            subprocess.check_call(["make", "-C", _DIR, "-s", VARIANTS[variant]])
        L = C.CDLL(path)
        fp, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.bo_expf.restype = C.c_float; L.bo_expf.argtypes = [C.c_float]
        L.bo_logf.restype = C.c_float; L.bo_logf.argtypes = [C.c_float]
        L.bo_calc_sigma.restype = C.c_float; L.bo_calc_sigma.argtypes = [C.c_float] * 7
        L.bo_powi.restype = C.c_float; L.bo_powi.argtypes = [C.c_float, C.c_int]
        L.bo_camera_setup.restype = None
        L.bo_camera_setup.argtypes = [fp, fp, C.c_double, C.c_double, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.POINTER(BoCamera)]
        L.bo_camera_setup_model.restype = None
        L.bo_camera_setup_model.argtypes = [fp, fp, C.c_double, C.c_double, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, fp, C.POINTER(BoCamera)]
        for nm in ("bo_focal_to_fov_model", "bo_fov_to_focal_model"):
            getattr(L, nm).restype = C.c_double
            getattr(L, nm).argtypes = [C.c_double, C.c_uint32, C.c_uint32, fp]
        L.bo_focal_to_fov.restype = C.c_double; L.bo_focal_to_fov.argtypes = [C.c_double, C.c_uint32]
        L.bo_fov_to_focal.restype = C.c_double; L.bo_fov_to_focal.argtypes = [C.c_double, C.c_uint32]
        L.bo_render_create.restype = C.c_void_p
        L.bo_render_free.argtypes = [C.c_void_p]
        L.bo_render_forward.restype = C.c_int
        L.bo_render_forward.argtypes = [C.c_void_p, C.POINTER(BoCamera), C.c_uint32, C.c_uint32, fp, fp, fp, fp, C.c_uint32]
        L.bo_render_backward.restype = C.c_int
        L.bo_render_backward.argtypes = [C.c_void_p, fp, fp, fp, fp]
        for nm in ("bo_num_visible", "bo_num_intersections", "bo_num_tiles"):
            getattr(L, nm).restype = C.c_uint32
            getattr(L, nm).argtypes = [C.c_void_p]
        L.bo_stage_seconds.restype = C.c_double; L.bo_stage_seconds.argtypes = [C.c_void_p, C.c_int]
        L.bo_radix_argsort.restype = None
        L.bo_radix_argsort.argtypes = [u32p, u32p, C.c_uint64, C.c_uint32, u32p, u32p]
        L.bo_prefix_sum.restype = None
        L.bo_prefix_sum.argtypes = [u32p, C.c_uint64, u32p]
        L.bo_image_loss_forward.restype = None
        L.bo_image_loss_forward.argtypes = [fp, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, fp, C.c_int, C.c_int, fp]
        L.bo_image_loss_backward.restype = None
        L.bo_image_loss_backward.argtypes = [fp, u32p, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, fp, C.c_int, C.c_int, fp]
        L.bo_adam_step.restype = None
        L.bo_adam_step.argtypes = [fp, fp, fp, fp, C.c_uint64, C.c_uint32, fp, C.c_float, C.c_uint32, C.c_int, C.c_float, C.c_float, C.c_float]
        L.bo_gather_stats.restype = None
        L.bo_gather_stats.argtypes = [fp, fp, fp, fp, fp, fp, C.c_uint64]
        L.bo_num_threads.restype = C.c_int
        L.bo_fold_min_scale.restype = None
        L.bo_fold_min_scale.argtypes = [fp, fp, fp, C.c_uint64, fp, fp]
        L.bo_fold_min_scale_backward.restype = None
        L.bo_fold_min_scale_backward.argtypes = [fp, fp, fp, C.c_uint64, fp, fp]
        L.bo_compute_min_scale.restype = None
        L.bo_compute_min_scale.argtypes = [fp, C.c_uint64, fp, C.c_uint32, C.c_float, fp]
        _libs[variant] = L
    return _libs[variant]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def camera(pos=(0.0, 0.0, 0.0), rot_xyzw=(0.0, 0.0, 0.0, 1.0), fov_x=1.0, fov_y=1.0, center_uv=(0.5, 0.5), img_w=64, img_h=64,
           model="pinhole", dist=None):
    """brush-render/src/camera.rs Camera -> kernel uniforms (oracle restatement).  `model` is
    "pinhole" | "kb4" (dist k1..k4) | "rt8" (k1 k2 k3 k4 k5 k6 p1 p2) | "tpf" (k1..k4 p1 p2 sx1 sy1)."""
    cam = BoCamera()
    p = f32(pos)
    r = f32(rot_xyzw)
    m, d = _model_args(model, dist)
    lib().bo_camera_setup_model(_fp(p), _fp(r), float(fov_x), float(fov_y), float(center_uv[0]), float(center_uv[1]), int(img_w), int(img_h),
                                m, _fp(d), C.byref(cam))
    return cam


def fov_to_focal(fov, pixels, model="pinhole", dist=None):
    m, d = _model_args(model, dist)
    return lib().bo_fov_to_focal_model(float(fov), int(pixels), m, _fp(d))


def focal_to_fov(focal, pixels, model="pinhole", dist=None):
    m, d = _model_args(model, dist)
    return lib().bo_focal_to_fov_model(float(focal), int(pixels), m, _fp(d))


_GETTERS = {
    "intersect_counts": np.uint32, "max_radius": np.float32, "depths_sorted": np.float32,
    "global_from_compact_gid": np.uint32, "cum_tiles_hit": np.uint32, "projected": np.float32,
    "tile_id_unsorted": np.uint32, "gid_unsorted": np.uint32,
    "tile_id_from_isect": np.uint32, "compact_gid_from_isect": np.uint32,
    "tile_offsets_pre": np.uint32, "tile_offsets": np.uint32, "out_img": np.float32,
    "out_packed": np.uint32, "visible": np.float32, "v_combined": np.float32,
    "v_transforms": np.float32, "v_coeffs": np.float32, "v_raw_opac": np.float32, "v_refine": np.float32,
}


class Render:
    """One forward (+ optional backward) of the oracle. Mirrors RenderOutput /
    RenderAuxInner (brush-render/src/render_aux.rs:17-68)."""

    def __init__(self, variant="spec"):
        self._lib = lib(variant)
        self._h = C.c_void_p(self._lib.bo_render_create())

    def __del__(self):
        try:
            self._lib.bo_render_free(self._h)
        except Exception:
            pass

    def forward(self, cam, transforms, sh, raw_opac, bg=(0.0, 0.0, 0.0), flags=FLAG_BWD_INFO):
        self.transforms = f32(transforms).reshape(-1, 10)
        n = self.transforms.shape[0]
        self.sh = f32(sh).reshape(n, -1, 3) if n else f32(sh).reshape(0, 1, 3)
        ncoef = self.sh.shape[1]
        self.sh_degree = int(round(ncoef ** 0.5)) - 1
        assert (self.sh_degree + 1) ** 2 == ncoef
        self.raw_opac = f32(raw_opac).reshape(n)
        self.cam = cam
        self.flags = flags
        bgv = f32(bg)
        rc = self._lib.bo_render_forward(self._h, C.byref(cam), n, self.sh_degree, _fp(self.transforms), _fp(self.sh), _fp(self.raw_opac), _fp(bgv), flags)
        if rc != 0:
            raise RuntimeError("bo_render_forward failed (rc=%d)" % rc)
        self.num_visible = self._lib.bo_num_visible(self._h)
        self.num_intersections = self._lib.bo_num_intersections(self._h)
        self.num_tiles = self._lib.bo_num_tiles(self._h)
        return self

    def backward(self, v_output):
        v = f32(v_output).reshape(self.cam.img_h, self.cam.img_w, 4)
        rc = self._lib.bo_render_backward(self._h, _fp(v), _fp(self.transforms), _fp(self.sh), _fp(self.raw_opac))
        if rc != 0:
            raise RuntimeError("bo_render_backward failed (rc=%d)" % rc)
        return self

    def get(self, name):
        cnt = C.c_uint64(0)
        fn = getattr(self._lib, "bo_get_" + name)
        ct = C.c_float if _GETTERS[name] == np.float32 else C.c_uint32
        fn.restype = C.POINTER(ct)
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        p = fn(self._h, C.byref(cnt))
        if cnt.value == 0:
            return np.zeros(0, dtype=_GETTERS[name])
        return np.ctypeslib.as_array(p, shape=(cnt.value,)).copy()

    def image(self):
        return self.get("out_img").reshape(self.cam.img_h, self.cam.img_w, 4)

    def stage_seconds(self):
        names = ["project_forward", "depth_sort", "scan", "project_visible", "map_isect", "tile_sort", "tile_offsets", "rasterize", "rasterize_bwd", "project_bwd"]
        return {k: self._lib.bo_stage_seconds(self._h, i) for i, k in enumerate(names)}


def radix_argsort(keys, vals, bits=32):
    k = np.ascontiguousarray(keys, dtype=np.uint32)
    v = np.ascontiguousarray(vals, dtype=np.uint32)
    ok, ov = np.empty_like(k), np.empty_like(v)
    lib().bo_radix_argsort(_u32p(k), _u32p(v), k.size, bits, _u32p(ok), _u32p(ov))
    return ok, ov


def prefix_sum(x):
    a = np.ascontiguousarray(x, dtype=np.uint32)
    o = np.empty_like(a)
    lib().bo_prefix_sum(_u32p(a), a.size, _u32p(o))
    return o


def image_loss_forward(pred_chw, gt_packed, l1_w, ssim_w, bg=None, mask=False):
    p = f32(pred_chw)
    c, h, w = p.shape
    g = np.ascontiguousarray(gt_packed, dtype=np.uint32).reshape(h, w)
    out = np.zeros_like(p)
    bgv = f32(bg if bg is not None else (0, 0, 0))
    lib().bo_image_loss_forward(_fp(p), _u32p(g), c, h, w, l1_w, ssim_w, _fp(bgv), int(bg is not None), int(mask), _fp(out))
    return out


def image_loss_backward(pred_chw, gt_packed, dl_dmap, l1_w, ssim_w, bg=None, mask=False):
    p = f32(pred_chw)
    c, h, w = p.shape
    g = np.ascontiguousarray(gt_packed, dtype=np.uint32).reshape(h, w)
    d = f32(dl_dmap).reshape(c, h, w)
    out = np.zeros_like(p)
    bgv = f32(bg if bg is not None else (0, 0, 0))
    lib().bo_image_loss_backward(_fp(p), _u32p(g), _fp(d), c, h, w, l1_w, ssim_w, _fp(bgv), int(bg is not None), int(mask), _fp(out))
    return out


def adam_step(param, grad, m1, m2, lr, t, col_scale=None, reduce_m2=False, beta1=0.9, beta2=0.999, eps=1e-15):
    """In place on param/m1/m2 (float32 C-contiguous arrays). param is [rows, row_len]."""
    assert param.dtype == np.float32 and param.flags.c_contiguous
    rows = param.shape[0]
    row_len = int(param.size // rows) if rows else 1
    g = f32(grad)
    cs = f32(col_scale) if col_scale is not None else None
    lib().bo_adam_step(_fp(param), _fp(g), _fp(m1), _fp(m2), rows, row_len, _fp(cs) if cs is not None else None, lr, t, int(reduce_m2), beta1, beta2, eps)


def fold_min_scale(transforms, raw_opac, min_scale):
    """gaussian_splats.rs:86-111 -> (folded transforms [N,10], folded raw opacity [N])."""
    t, o, f = f32(transforms).reshape(-1, 10), f32(raw_opac).reshape(-1), f32(min_scale).reshape(-1)
    ot, oo = np.empty_like(t), np.empty_like(o)
    lib().bo_fold_min_scale(_fp(t), _fp(o), _fp(f), t.shape[0], _fp(ot), _fp(oo))
    return ot, oo


def fold_min_scale_backward(transforms, raw_opac, min_scale, v_folded_transforms, v_folded_raw_opac):
    """Chain gradients w.r.t. the folded tensors back to the raw parameters -> (v_transforms, v_raw_opac)."""
    t, o, f = f32(transforms).reshape(-1, 10), f32(raw_opac).reshape(-1), f32(min_scale).reshape(-1)
    vt, vo = f32(v_folded_transforms).reshape(-1, 10).copy(), f32(v_folded_raw_opac).reshape(-1).copy()
    lib().bo_fold_min_scale_backward(_fp(t), _fp(o), _fp(f), t.shape[0], _fp(vt), _fp(vo))
    return vt, vo


def compute_min_scale(transforms, view_cams, factor=0.1):
    """brush-train/src/train.rs:102-125; view_cams [K,4] = centre xyz + focal px."""
    t = f32(transforms).reshape(-1, 10)
    vc = f32(view_cams).reshape(-1, 4)
    out = np.empty(t.shape[0], np.float32)
    lib().bo_compute_min_scale(_fp(t), t.shape[0], _fp(vc), vc.shape[0], float(factor), _fp(out))
    return out


def num_threads():
    return lib().bo_num_threads()
