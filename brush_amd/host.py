"""Host-side mirror of the Brush operator surface over libbrush_hip.so.

See brush_amd/__init__.py for the reference citations. Every function here is
plumbing around one C-ABI call; tensors are torch CUDA(HIP) tensors used purely
as device memory.
"""
import ctypes as C
import enum
import math
import os
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch

from . import _ffi
from ._ffi import BrushHipError
from .parallel import (DIRECT_ALLREDUCE_MIN_FLOATS, allreduce_direct, allreduce_exchange, allreduce_refine_maxima, allgather_strips, exchange_strip_halos, strip_spans_px,
                       strips_allow_halo_loss, tile_rows_for_rank)


# ---------------------------------------------------------------------------
# context
# ---------------------------------------------------------------------------
# The library reads no environment variable (bh_set_option is its one configuration entry point).  THIS harness — the test suite,
# bench.py, scripts/ab.sh — still lets a developer pick an A/B path from the shell: BH_OPTIONS="key=value,key=value", and the
# variable names earlier rounds' scripts use, translated here into options of every Context this process creates.
_LEGACY_ENV_OPTIONS = (   # (environment variable, option key, value or None = the variable's own value)
    ("BH_NO_LPT", "no_lpt", "1"), ("BH_GENERIC_DEPTH_SORT", "generic_depth_sort", "1"), ("BH_FORCE_PG", "force_exchange", "1"),
    ("BH_TRAIN_ZERO_GRADS", "zero_grads", "1"), ("BH_CUT_MIN_PAIRS", "cut_min_pairs", None), ("BH_CUT_SORT_ALL", "cut_sort_all", "1"),
    ("BH_NO_VIEW_HASH", "no_view_hash", "1"), ("BH_CUT_MARGIN_FIXED", "cut_margin_fixed", "1"), ("BH_CUT_CTRL", "cut_ctrl", None),
    ("BH_READBACK_COPY", "readback_copy", "1"), ("BH_EVENT_WAITS", "event_waits", "1"), ("BH_K16_ORDER", "k16_order", None),
    ("BH_CUT_MARGIN_PCT", "cut_margin_pct", None), ("BH_UPDATE_EARLY", "update_early", "1"), ("BH_UPDATE_NO_DORMANT", "no_dormant", "1"),
    ("BH_K5_EXACT_SPW", "k5_exact_spw", None), ("BH_TILE_SORT_LSD", "tile_sort", "lsd"), ("BH_LOSS_BANDS", "loss_bands", None),
    ("BH_UPDATE_ROWS", "update_rows", None), ("BH_SORT_KPT", "sort_kpt", None),
)


def options_from_environment(env=None):
    """[(key, value)] a developer asked for through the shell (see above); applied by Context.__init__."""
    env = os.environ if env is None else env
    out = []
    for var, key, val in _LEGACY_ENV_OPTIONS:
        if var in env:
            out.append((key, env[var] if val is None else val))
    for item in filter(None, (x.strip() for x in env.get("BH_OPTIONS", "").split(","))):
        k, _, v = item.partition("=")
        out.append((k.strip(), v.strip()))
    return out


class Context:
    """One bh_ctx: a HIP stream + scratch arena. Single-threaded by contract
    (brush-async/src/lib.rs:1-17); make one per thread / per GPU."""

    def __init__(self, device=None, use_torch_stream=True, lib=None, options=None):
        # lib: tests only — the fault-injection build (_ffi.load_test_hooks()); everything a Context does goes through self.lib
        self.lib = lib if lib is not None else _ffi.load()
        if not torch.cuda.is_available():
            raise BrushHipError("no HIP device visible: brush_amd has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
        # submit on torch's current stream (handle 0 = the default stream) so tensor ops
        # and brush_hip kernels are ordered without extra synchronisation
        stream = torch.cuda.current_stream(self.device).cuda_stream if use_torch_stream else 0
        # own stream (hipStreamNonBlocking): NOT ordered against torch's streams — the caller synchronises hand-overs
        # (what a viewer / trainer pair on separate threads wants, brush-async/src/lib.rs:4-17)
        self.uses_torch_stream = bool(use_torch_stream)
        self._h = self.lib.bh_create(self.device.index, C.c_void_p(stream), 0 if use_torch_stream else 1)
        if not self._h:
            raise BrushHipError("bh_create failed on %s" % self.device)
        self._h = C.c_void_p(self._h)
        for k, v in options_from_environment() + list((options or {}).items()):
            self.set_option(k, v)

    def set_option(self, key, value):
        """bh_set_option: select one of the library's alternative paths (same results; A/B measurements and their tests)."""
        self.check(self.lib.bh_set_option(self._h, str(key).encode(), str(value).encode()))

    def options(self):
        """{key: help} of every option this build knows (bh_option_name / bh_option_help)."""
        return {self.lib.bh_option_name(i).decode(): self.lib.bh_option_help(i).decode() for i in range(self.lib.bh_option_count())}

    def close(self):
        if getattr(self, "_h", None):
            self.lib.bh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise BrushHipError("brush_hip error %d: %s" % (rc, self.lib.bh_last_error(self._h).decode()))

    def sync(self):
        self.check(self.lib.bh_sync(self._h))

    # ---- RCCL inside the library (bh_comm_*): for hosts without torch.distributed ----
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId: call on rank 0, hand the 128 bytes to every rank (file, socket, TCPStore ...)."""
        buf = (C.c_char * 128)()
        rc = _ffi.load().bh_comm_unique_id(buf)
        if rc != 0:
            raise BrushHipError("bh_comm_unique_id failed (%d): RCCL not loadable" % rc)
        return bytes(buf)

    def comm_init(self, rank, world, unique_id: bytes):
        self.check(self.lib.bh_comm_init(self._h, int(rank), int(world), unique_id))

    def comm_destroy(self):
        self.check(self.lib.bh_comm_destroy(self._h))

    def comm_world(self):
        return int(self.lib.bh_comm_world(self._h))

    def comm_rank(self):
        return int(self.lib.bh_comm_rank(self._h))

    def comm_selftest(self):
        """bh_comm_selftest: every RCCL entry point the library binds, on rank-dependent patterns, checked on the host (collective)."""
        self.check(self.lib.bh_comm_selftest(self._h))

    def allreduce_sum(self, t: torch.Tensor):
        """In place, asynchronous on the ctx stream; t: contiguous float32 device tensor."""
        self.check(self.lib.bh_allreduce_sum_f32(self._h, _ptr(t), t.numel()))

    def allreduce_max(self, t: torch.Tensor):
        self.check(self.lib.bh_allreduce_max_f32(self._h, _ptr(t), t.numel()))

    def allgather_bytes(self, send: torch.Tensor, recv: torch.Tensor):
        """bh_allgather_bytes: every rank's `send` (contiguous, any dtype) lands in `recv` (world x the same bytes) in rank order."""
        nbytes = send.numel() * send.element_size()
        if recv.numel() * recv.element_size() != nbytes * self.comm_world():
            raise ValueError("recv must hold world x the bytes of send")
        self.check(self.lib.bh_allgather_bytes(self._h, _ptr(send), _ptr(recv), nbytes))

    def exchange_strip_halos(self, img_hwc4: torch.Tensor, row_begin_px: int, row_end_px: int):
        """bh_exchange_strip_halos: one frame split into strips of tile rows — fetch the 21 pixel rows above and below this rank's strip
        [row_begin_px, row_end_px) of the [H,W,4] f32 image from the neighbouring ranks, in place on the ctx stream."""
        h, w, c = img_hwc4.shape
        if c != 4:
            raise ValueError("img must be [H,W,4]")
        self.check(self.lib.bh_exchange_strip_halos(self._h, _ptr(img_hwc4), h, w, int(row_begin_px), int(row_end_px)))

    @staticmethod
    def strip_halo_plan(img_h, row_begin_px, row_end_px, rank, world):
        """The host arithmetic behind exchange_strip_halos (bh_strip_halo_plan): [(send, peer, first row, rows)], at most four."""
        ops = (_ffi.BhHaloOp * 4)()
        k = _ffi.load().bh_strip_halo_plan(int(img_h), int(row_begin_px), int(row_end_px), int(rank), int(world), ops)
        if k < 0:
            raise BrushHipError("strip_halo_plan: bad strip (%d)" % k)
        return [(bool(ops[i].send), int(ops[i].peer), int(ops[i].row_begin_px), int(ops[i].rows)) for i in range(k)]

    def set_list_cut_threshold(self, min_pairs: int):
        """bh_set_list_cut_threshold: frames with fewer intersections than this keep complete lists (default 1.5 M)."""
        self.check(self.lib.bh_set_list_cut_threshold(self._h, int(min_pairs)))

    def view_table_count(self):
        """bh_view_table_count: per-view tables the ctx holds (diagnostics)."""
        return int(self.lib.bh_view_table_count(self._h))

    def forget_views(self):
        """bh_forget_views: drop the per-view tile tables (after loading another scene)."""
        self.check(self.lib.bh_forget_views(self._h))

    def profile(self, on=True):
        """True/1: HIP events around every stage; 2: only around the dominant kernel; False/0: off."""
        self.check(self.lib.bh_profile_enable(self._h, int(on)))

    def profile_fetch(self):
        """{stage: (total_ms, calls)} accumulated since the last fetch."""
        cap = 32
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        calls = (C.c_uint32 * cap)()
        n = self.lib.bh_profile_fetch(self._h, names, ms, calls, cap)
        return {names[i].decode(): (ms[i], calls[i]) for i in range(n)}


_CONTEXTS = {}


def get_context(device=None) -> Context:
    if not torch.cuda.is_available():
        raise BrushHipError("no HIP device visible: brush_amd has no CPU path")
    idx = torch.cuda.current_device() if device is None else (torch.device(device).index or 0)
    if idx not in _CONTEXTS:
        _CONTEXTS[idx] = Context(torch.device("cuda", idx))
    return _CONTEXTS[idx]


class _DevArray:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _view(ptr, shape, dtype, device):
    """Zero-copy torch view of ctx-owned device memory (valid until the next forward)."""
    n = 1
    for s in shape:
        n *= s
    if n == 0 or not ptr:
        return torch.empty(shape, dtype=dtype, device=device)
    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(t.data_ptr() if t is not None else 0)


def _f32c(t, device):
    t = torch.as_tensor(t, dtype=torch.float32, device=device)
    return t.contiguous()  # render.rs:57-59 into_contiguous


# ---------------------------------------------------------------------------
# Camera (brush-render/src/camera.rs)
# ---------------------------------------------------------------------------
def _model_and_dist(camera_model, dist):
    c = Camera(camera_model=camera_model, dist=tuple(dist or ()))
    return c.model_id(), c._dist8()


def fov_to_focal(fov, pixels, camera_model="pinhole", dist=None):
    """camera.rs:85-101 (f64): focal such that pixels/2 = focal * lens_law(fov/2)."""
    m, d = _model_and_dist(camera_model, dist)
    return _ffi.load().bh_fov_to_focal(float(fov), int(pixels), m, d)


def focal_to_fov(focal, pixels, camera_model="pinhole", dist=None):
    """camera.rs:104-118 (f64)."""
    m, d = _model_and_dist(camera_model, dist)
    return _ffi.load().bh_focal_to_fov(float(focal), int(pixels), m, d)


@dataclass
class Camera:
    position: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    rotation: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 1.0)  # glam order x, y, z, w
    fov_x: float = 1.0
    fov_y: float = 1.0
    center_uv: Tuple[float, float] = (0.5, 0.5)
    # CameraModel (kernels/camera_model/mod.rs:31-38): "pinhole" | "kb4" | "rt8" | "tpf" (or the BH_CAMERA_* id)
    # with its distortion parameters in the reference's struct order (see include/brush_hip.h)
    camera_model: object = "pinhole"
    dist: Tuple[float, ...] = ()

    def model_id(self):
        ids = {"pinhole": _ffi.CAMERA_PINHOLE, "kb4": _ffi.CAMERA_KANNALA_BRANDT_4, "rt8": _ffi.CAMERA_RADIAL_TANGENTIAL_8,
               "tpf": _ffi.CAMERA_THIN_PRISM_FISHEYE}
        return ids[self.camera_model] if isinstance(self.camera_model, str) else int(self.camera_model)

    def _dist8(self):
        d = (C.c_float * 8)()
        for i, v in enumerate(self.dist):
            d[i] = float(v)
        return d

    def is_valid(self):
        vals = list(self.position) + list(self.rotation) + [self.fov_x, self.fov_y] + list(self.center_uv)
        return all(math.isfinite(v) for v in vals)

    def uniforms(self, img_size, tile_rows=None) -> "_ffi.BhCamera":
        """Kernel uniforms for an (img_w, img_h) render: pinhole params, 3x4 view
        matrix, Jacobian clamp limits (camera.rs:63-101,200-254).  `tile_rows` = (begin, end)
        restricts the render to a strip of 16-px tile rows (one frame partitioned over GPUs)."""
        w, h = int(img_size[0]), int(img_size[1])
        cam = _ffi.BhCamera()
        pos = (C.c_float * 3)(*self.position)
        rot = (C.c_float * 4)(*self.rotation)
        rc = _ffi.load().bh_camera_setup_model(pos, rot, float(self.fov_x), float(self.fov_y), float(self.center_uv[0]),
                                               float(self.center_uv[1]), w, h, self.model_id(), self._dist8(), C.byref(cam))
        if rc != 0:
            raise BrushHipError("bh_camera_setup_model failed (%d): image size must be non-zero, camera model known" % rc)
        if tile_rows is not None:
            cam.tile_row_begin, cam.tile_row_end = int(tile_rows[0]), int(tile_rows[1])
        return cam


# ---------------------------------------------------------------------------
# Splats (brush-render/src/gaussian_splats.rs:62-74)
# ---------------------------------------------------------------------------
class RasterPass(enum.Enum):
    Forward = 0
    Backward = 1
    BackwardSmoothCutoff = 2

    def bwd_info(self):
        return self is not RasterPass.Forward

    def smooth_cutoff(self):
        return self is RasterPass.BackwardSmoothCutoff


class Splats:
    """transforms [N,10] = means(3) quat wxyz(4) log-scales(3); sh_coeffs [N,C,3];
    raw_opacities [N] (logits)."""

    def __init__(self, transforms, sh_coeffs, raw_opacities, render_mip=False, device=None, min_scale=None):
        device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # Splats::min_scale (gaussian_splats.rs:69-73): optional frozen per-splat world-space scale floor [N]
        self.min_scale = None if min_scale is None else _f32c(min_scale, device).reshape(-1)
        self.transforms = _f32c(transforms, device).reshape(-1, 10)
        n = self.transforms.shape[0]
        self.sh_coeffs = _f32c(sh_coeffs, device).reshape(n, -1, 3) if n else _f32c(sh_coeffs, device).reshape(0, 1, 3)
        self.raw_opacities = _f32c(raw_opacities, device).reshape(n)
        self.render_mip = bool(render_mip)
        c = self.sh_coeffs.shape[1]
        deg = int(round(math.sqrt(c))) - 1
        if (deg + 1) ** 2 != c or deg > 4:
            raise ValueError("sh_coeffs must have (d+1)^2 coefficients, d <= 4 (got %d)" % c)  # sh.rs sh_degree_from_coeffs
        self._sh_degree = deg

    @staticmethod
    def from_tensor_data(means, rotations_wxyz, log_scales, sh_coeffs, raw_opacities, render_mip=False, device=None):
        device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        tr = torch.cat([_f32c(means, device).reshape(-1, 3), _f32c(rotations_wxyz, device).reshape(-1, 4),
                        _f32c(log_scales, device).reshape(-1, 3)], dim=1)
        return Splats(tr, sh_coeffs, raw_opacities, render_mip, device)

    def num_splats(self):
        return self.transforms.shape[0]

    def sh_degree(self):
        return self._sh_degree

    @property
    def device(self):
        return self.transforms.device

    def clone(self):
        return Splats(self.transforms.clone(), self.sh_coeffs.clone(), self.raw_opacities.clone(), self.render_mip, self.device,
                      None if self.min_scale is None else self.min_scale.clone())

    # ---- Mip-Splatting 3D filter (gaussian_splats.rs:188-256) ----
    def with_min_scale(self, f):
        """Attach a per-splat world-space scale floor [N] (Splats::with_min_scale)."""
        f = _f32c(f, self.device).reshape(-1)
        if f.numel() != self.num_splats():
            raise ValueError("min_scale must have one entry per splat")
        self.min_scale = f
        return self

    def folded(self, ctx=None):
        """(transforms, raw_opacities) the renderer sees: fold_min_scale(params) when a floor is set
        (gaussian_splats.rs:379-386), the raw parameters otherwise."""
        if self.min_scale is None:
            return self.transforms, self.raw_opacities
        ctx = ctx or get_context(self.device)
        ft, fo = torch.empty_like(self.transforms), torch.empty_like(self.raw_opacities)
        ctx.check(ctx.lib.bh_fold_min_scale(ctx._h, _ptr(self.transforms), _ptr(self.raw_opacities), _ptr(self.min_scale), self.num_splats(),
                                            _ptr(ft), _ptr(fo)))
        return ft, fo

    def opacities(self, ctx=None):
        """Post-activation opacity incl. the floor's energy compensation (Splats::opacities)."""
        return torch.sigmoid(self.folded(ctx)[1])

    def scales(self, ctx=None):
        """World-space scales sqrt(s^2 + f^2) (Splats::scales)."""
        return torch.exp(self.folded(ctx)[0][:, 7:10])

    def bake_min_scale(self, ctx=None):
        """Permanently fold the floor into the raw parameters and clear it (Splats::bake_min_scale): in place."""
        if self.min_scale is not None:
            ctx = ctx or get_context(self.device)
            ctx.check(ctx.lib.bh_fold_min_scale(ctx._h, _ptr(self.transforms), _ptr(self.raw_opacities), _ptr(self.min_scale), self.num_splats(),
                                                _ptr(self.transforms), _ptr(self.raw_opacities)))
            self.min_scale = None
        return self


@dataclass
class RenderAux:
    """RenderAux (brush-render/src/render_aux.rs:17-68) + the tensors saved for backward."""
    num_visible: int
    num_intersections: int
    img_size: Tuple[int, int]
    visible: Optional[torch.Tensor]
    max_radius: torch.Tensor
    tile_offsets: torch.Tensor
    projected_splats: torch.Tensor
    compact_gid_from_isect: torch.Tensor
    tile_id_from_isect: torch.Tensor
    global_from_compact_gid: torch.Tensor
    cum_tiles_hit: torch.Tensor
    intersect_counts: torch.Tensor
    depths_sorted: torch.Tensor
    # depth-sliced lists (render_splats(..., sliced=True), the train step's default): the far slice's [T,2] segment table, or
    # None when the lists are the exact ones; list_budget = pairs the near pass listed; num_listed_splats = entries of the compact
    # arrays (per-tile cuts sort and number only the splats that own a listed pair: a sub-sequence of the full depth order)
    tile_offsets_far: Optional[torch.Tensor] = None
    list_budget: int = 0
    num_listed_splats: int = 0

    def validate(self, num_splats):
        # render_aux.rs:30-45
        tiles = self.tile_offsets.shape[0]
        assert self.num_visible <= num_splats
        assert self.num_intersections <= max(self.num_visible, 1) * max(tiles, 1)


def set_list_slicing(near_share: float, ctx: Optional["Context"] = None, device=None):
    """Near slice's share of the pair list for sliced forwards on `ctx` (bh_set_list_slicing): (0, 1] fixed, <= 0 automatic."""
    ctx = ctx or get_context(device)
    ctx.check(ctx.lib.bh_set_list_slicing(ctx._h, float(near_share)))


def set_view_id(view_id: int, ctx: Optional["Context"] = None, device=None):
    """The view the following forwards on `ctx` render (bh_set_view_id; sticky, 0 = unknown): selects the per-tile depth-cut
    table that sliced forwards read and refresh."""
    ctx = ctx or get_context(device)
    ctx.check(ctx.lib.bh_set_view_id(ctx._h, int(view_id)))


def _forward(ctx, splats, camera, img_size, background, pass_, sliced=False):
    if img_size[0] <= 0 or img_size[1] <= 0:
        raise BrushHipError("Can't render images with 0 size.")  # render.rs:50-53
    cam = camera if isinstance(camera, _ffi.BhCamera) else camera.uniforms(img_size)
    flags = (_ffi.FLAG_MIP if splats.render_mip else 0)
    if pass_.bwd_info():
        flags |= _ffi.FLAG_BWD_INFO
    if pass_.smooth_cutoff():
        flags |= _ffi.FLAG_SMOOTH_CUTOFF
    if sliced:
        flags |= _ffi.FLAG_SLICED_LISTS
    out = _ffi.BhRenderOut()
    bg = (C.c_float * 3)(*[float(b) for b in background])
    n = splats.num_splats()
    r_t, r_o = splats.folded(ctx)  # gaussian_splats.rs:379-386: the 3D-filter floor is part of the splat
    ctx.check(ctx.lib.bh_render_forward(ctx._h, C.byref(cam), n, splats.sh_degree(), _ptr(r_t), _ptr(splats.sh_coeffs),
                                        _ptr(r_o), bg, flags, C.byref(out)))
    return cam, out, (r_t, r_o)


def last_list_counts(ctx: Optional["Context"] = None, device=None):
    """(near_pairs, far_pairs) the last forward on `ctx` actually listed (bh_last_list_counts; blocking): the pair lists of a
    depth-sliced forward hold that many defined entries, not num_intersections."""
    ctx = ctx or get_context(device)
    a, b = C.c_uint32(), C.c_uint32()
    ctx.check(ctx.lib.bh_last_list_counts(ctx._h, C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


def _aux_from(out, n, w, h, device, copy, ctx=None):
    nv, ni, T = out.num_visible, out.num_intersections, out.num_tiles
    nl = out.num_listed_splats   # entries of the compact (depth-ordered) arrays: == nv unless per-tile cuts listed a subset of the splats
    i32, f32 = torch.int32, torch.float32
    listed = ni
    if out.tile_offsets_far and copy and ctx is not None:
        # depth-sliced lists: only the near slice's pairs and, behind them, the far slice's are defined (copies are trimmed to
        # them; copy=False views keep the arena's length and cost no readback)
        near, far = last_list_counts(ctx)
        listed = min(ni, near + far)

    def mk(ptr, shape, dt):
        v = _view(ptr, shape, dt, device)
        return v.clone() if copy else v
    return RenderAux(
        num_visible=nv, num_intersections=ni, img_size=(w, h),
        visible=mk(out.visible, (n,), f32) if out.visible else None,
        max_radius=mk(out.max_radius, (n,), f32),
        tile_offsets=mk(out.tile_offsets, (T, 2), i32),
        projected_splats=mk(out.projected, (nl, 9), f32),
        compact_gid_from_isect=mk(out.compact_gid_from_isect, (listed,), i32),
        tile_id_from_isect=mk(out.tile_id_from_isect, (listed,), i32),
        global_from_compact_gid=mk(out.global_from_compact_gid, (nl,), i32),
        cum_tiles_hit=mk(out.cum_tiles_hit, (nl,), i32),
        intersect_counts=mk(out.intersect_counts, (n,), i32),
        depths_sorted=mk(out.depths_sorted, (nl,), f32),
        tile_offsets_far=mk(out.tile_offsets_far, (T, 2), i32) if out.tile_offsets_far else None,
        list_budget=int(out.list_budget),
        num_listed_splats=int(nl),
    )


def render_splats(splats: Splats, camera, img_size, background=(0.0, 0.0, 0.0), pass_: RasterPass = RasterPass.Forward,
                  ctx: Optional[Context] = None, copy=True, tile_rows=None, sliced=False):
    """Forward render. RasterPass.Forward returns a packed rgba8 image [H,W] (int32
    bit pattern, r in bits 0-7); the Backward variants return f32 [H,W,4].
    Returns (image, RenderAux). With copy=False the tensors alias ctx scratch
    memory and are only valid until the next render on `ctx`.
    sliced=True: BH_FLAG_SLICED_LISTS (same image / visible / counts; the list outputs are truncated, see the header)."""
    ctx = ctx or get_context(splats.device)
    w, h = int(img_size[0]), int(img_size[1])
    if tile_rows is not None and not isinstance(camera, _ffi.BhCamera):
        camera = camera.uniforms((w, h), tile_rows)
    _, out, _ = _forward(ctx, splats, camera, (w, h), background, pass_, sliced)
    if pass_.bwd_info():
        img = _view(out.out_img, (h, w, 4), torch.float32, splats.device)
    else:
        img = _view(out.out_img_packed, (h, w), torch.int32, splats.device)
    if copy:
        img = img.clone()
    return img, _aux_from(out, splats.num_splats(), w, h, splats.device, copy, ctx)


def render_splats_bwd(splats: Splats, camera, img_size, background, v_output, pass_: RasterPass = RasterPass.Backward,
                      ctx: Optional[Context] = None, tile_rows=None, sliced=False):
    """Differentiable render: forward (Backward pass flags) + backward for a given
    dL/d(out_img) `v_output` [H,W,4] (a tensor, or a callable img -> v_output).
    Returns dict(img, aux, v_transforms, v_sh_coeffs, v_raw_opacities, v_refine_weight, v_combined)."""
    assert pass_.bwd_info(), "render_splats_bwd requires a Backward variant"  # bwd/burn_glue.rs:281-284
    ctx = ctx or get_context(splats.device)
    w, h = int(img_size[0]), int(img_size[1])
    dev = splats.device
    if tile_rows is not None and not isinstance(camera, _ffi.BhCamera):
        camera = camera.uniforms((w, h), tile_rows)
    _, out, (r_t, r_o) = _forward(ctx, splats, camera, (w, h), background, pass_, sliced)
    img = _view(out.out_img, (h, w, 4), torch.float32, dev).clone()
    aux = _aux_from(out, splats.num_splats(), w, h, dev, True, ctx)
    if callable(v_output):
        v_output = v_output(img)
    v_output = _f32c(v_output, dev).reshape(h, w, 4)
    n, c = splats.num_splats(), splats.sh_coeffs.shape[1]
    v_t = torch.empty((n, 10), dtype=torch.float32, device=dev)
    v_sh = torch.empty((n, c, 3), dtype=torch.float32, device=dev)
    v_op = torch.empty((n,), dtype=torch.float32, device=dev)
    v_rf = torch.empty((n,), dtype=torch.float32, device=dev)
    # the saved state goes in explicitly (SplatBwdOps::rasterize_bwd / project_bwd, bwd/burn_glue.rs:62-92)
    ctx.check(ctx.lib.bh_render_backward_saved(ctx._h, C.byref(out), _ptr(v_output), _ptr(r_t), _ptr(splats.sh_coeffs), _ptr(r_o),
                                               _ptr(v_t), _ptr(v_sh), _ptr(v_op), _ptr(v_rf)))
    if splats.min_scale is not None:  # chain through the fold (the autodiff of bwd/burn_glue.rs:260-270)
        ctx.check(ctx.lib.bh_fold_min_scale_backward(ctx._h, _ptr(splats.transforms), _ptr(splats.raw_opacities), _ptr(splats.min_scale), n,
                                                     _ptr(v_t), _ptr(v_op)))
    vc = _view(ctx.lib.bh_last_v_combined(ctx._h), (max(out.num_listed_splats, 1), 10), torch.float32, dev).clone()
    return dict(img=img, aux=aux, v_transforms=v_t, v_sh_coeffs=v_sh, v_raw_opacities=v_op, v_refine_weight=v_rf, v_combined=vc)


class RenderNode:
    """One differentiable render = the autodiff node `render_splats` registers in the reference (bwd/burn_glue.rs:223-311: the
    forward's outputs + the state RenderBackwards saves, :336-371).  `backward(v_output)` may be called after OTHER renders have
    run on the same ctx if the node was created with retain=True (bh_render_retain); without it a later forward makes the node
    stale and backward raises BrushHipError (BH_ERR_STATE) instead of returning another frame's gradients."""

    def __init__(self, ctx, splats, out, folded, img_size, retained):
        self.ctx, self.splats, self.out, self._folded, self.img_size, self.retained = ctx, splats, out, folded, img_size, retained
        w, h = img_size
        self.img = _view(out.out_img, (h, w, 4), torch.float32, splats.device)   # aliases ctx memory: valid while the node is (retained nodes: until release)

    def backward(self, v_output):
        ctx, splats, dev = self.ctx, self.splats, self.splats.device
        w, h = self.img_size
        n, c = splats.num_splats(), splats.sh_coeffs.shape[1]
        v_output = _f32c(v_output, dev).reshape(h, w, 4)
        r_t, r_o = self._folded
        v_t = torch.empty((n, 10), dtype=torch.float32, device=dev)
        v_sh = torch.empty((n, c, 3), dtype=torch.float32, device=dev)
        v_op = torch.empty((n,), dtype=torch.float32, device=dev)
        v_rf = torch.empty((n,), dtype=torch.float32, device=dev)
        ctx.check(ctx.lib.bh_render_backward_saved(ctx._h, C.byref(self.out), _ptr(v_output), _ptr(r_t), _ptr(splats.sh_coeffs), _ptr(r_o),
                                                   _ptr(v_t), _ptr(v_sh), _ptr(v_op), _ptr(v_rf)))
        if splats.min_scale is not None:
            ctx.check(ctx.lib.bh_fold_min_scale_backward(ctx._h, _ptr(splats.transforms), _ptr(splats.raw_opacities), _ptr(splats.min_scale), n,
                                                         _ptr(v_t), _ptr(v_op)))
        return dict(v_transforms=v_t, v_sh_coeffs=v_sh, v_raw_opacities=v_op, v_refine_weight=v_rf)

    def release(self):
        if self.retained:
            self.ctx.check(self.ctx.lib.bh_render_release(self.ctx._h, C.byref(self.out)))
            self.retained = False


def render_splats_diff(splats: Splats, camera, img_size, background=(0.0, 0.0, 0.0), pass_: RasterPass = RasterPass.Backward,
                       ctx: Optional[Context] = None, retain=False, sliced=False) -> RenderNode:
    """Forward of a differentiable render; gradients later through RenderNode.backward (bwd/burn_glue.rs:223-311)."""
    assert pass_.bwd_info()
    ctx = ctx or get_context(splats.device)
    w, h = int(img_size[0]), int(img_size[1])
    _, out, folded = _forward(ctx, splats, camera, (w, h), background, pass_, sliced)
    if retain:
        ctx.check(ctx.lib.bh_render_retain(ctx._h, C.byref(out)))
    return RenderNode(ctx, splats, out, folded, (w, h), bool(retain))


# ---------------------------------------------------------------------------
# PLY at the edges (brush-serde)
# ---------------------------------------------------------------------------
def splat_to_ply(splats: Splats, up_axis=None, ctx: Optional[Context] = None) -> bytes:
    """splat_to_ply (brush-serde/src/export.rs:179-204): the INRIA-layout binary PLY, rows packed on the
    device (3D-filter floor baked, quaternions normalised, SH permuted to [channel][coeff])."""
    ctx = ctx or get_context(splats.device)
    n = splats.num_splats()
    up = (C.c_float * 3)(*[float(v) for v in up_axis]) if up_axis is not None else None
    need = C.c_uint64(0)
    args = (_ptr(splats.transforms), _ptr(splats.sh_coeffs), _ptr(splats.raw_opacities),
            _ptr(splats.min_scale) if splats.min_scale is not None else None, n, splats.sh_degree(), int(splats.render_mip), up)
    ctx.check(ctx.lib.bh_splat_to_ply(ctx._h, *args, None, 0, C.byref(need)))
    buf = (C.c_char * need.value)()
    ctx.check(ctx.lib.bh_splat_to_ply(ctx._h, *args, buf, need.value, C.byref(need)))
    return bytes(buf)


@dataclass
class ParseMetadata:
    """brush-serde/src/import.rs:19-24"""
    up_axis: Optional[Tuple[float, float, float]]
    render_mode: Optional[str]
    total_splats: int
    sh_degree: int
    compressed: bool = False   # a SuperSplat-compressed file (import.rs:244-250, 407-600)


def ply_parse_header(data: bytes) -> ParseMetadata:
    """Header of a splat PLY (host only; no GPU needed)."""
    info = _ffi.BhPlyInfo()
    rc = _ffi.load().bh_ply_parse_header(data, len(data), C.byref(info))
    if rc != 0:
        raise BrushHipError("unsupported PLY (%d): only binary_little_endian float vertex rows (or SuperSplat-compressed chunks) are read" % rc
                            if rc == -5 else "malformed PLY (%d)" % rc)
    return ParseMetadata(tuple(info.up_axis) if info.has_up_axis else None, {0: "default", 1: "mip"}.get(info.render_mode), int(info.num_splats),
                         int(info.sh_degree), bool(info.compressed))


def load_splat_from_ply(data: bytes, device=None, render_mip=None, ctx: Optional[Context] = None, subsample_points: Optional[int] = None,
                        max_splats: Optional[int] = None):
    """load_splat_from_ply + SplatData::into_splats (import.rs:166-181, 77-97) -> (Splats, ParseMetadata).
    render_mip None = take the file's SplatRenderMode comment (default mode when absent).
    subsample_points = s keeps every s-th row (rows s-1, 2s-1, ...: import.rs:346-349); max_splats then applies
    SplatData::subsample (import.rs:49-74: rows 0, step, 2 step, ... with step = ceil(n / max_splats); 0 / None = no cap), as
    the training stream does (brush-process/src/train_stream.rs:111).  meta.total_splats is the number of rows kept."""
    meta = ply_parse_header(data)
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    ctx = ctx or get_context(device)
    s = int(subsample_points or 1)
    if s < 1:
        raise BrushHipError("subsample_points must be >= 1")
    first, step, n = s - 1, s, meta.total_splats // s
    if max_splats and n > max_splats:
        step2 = -(-n // int(max_splats))
        n, step = -(-n // step2), step * step2
    c = (meta.sh_degree + 1) ** 2
    tr = torch.empty((n, 10), dtype=torch.float32, device=device)
    sh = torch.empty((n, c, 3), dtype=torch.float32, device=device)
    op = torch.empty((n,), dtype=torch.float32, device=device)
    ctx.check(ctx.lib.bh_splats_from_ply_strided(ctx._h, data, len(data), first, step, n, _ptr(tr), _ptr(sh), _ptr(op)))
    mip = (meta.render_mode == "mip") if render_mip is None else bool(render_mip)
    meta.total_splats = n
    return Splats(tr, sh, op, mip, device), meta


# ---------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------
def _as_u32(t, device):
    t = torch.as_tensor(t, device=device)
    if t.dtype not in (torch.int32, torch.uint32):
        t = t.to(torch.int64).to(torch.int32) if t.dtype != torch.int64 else (t & 0xFFFFFFFF).to(torch.int32)
    return t.contiguous()


def radix_argsort(keys, values=None, sorting_bits=32, ctx: Optional[Context] = None):
    """Stable argsort of u32 keys (int32 bit patterns) on their low `sorting_bits`
    bits; returns (sorted_keys, sorted_values). brush-sort/src/lib.rs:16-33 asserts."""
    dev = keys.device if isinstance(keys, torch.Tensor) and keys.is_cuda else torch.device("cuda", torch.cuda.current_device())
    ctx = ctx or get_context(dev)
    k = _as_u32(keys, dev)
    if k.dim() != 1:
        raise BrushHipError("radix_argsort: keys must be 1-D")
    v = None
    if values is not None:
        v = _as_u32(values, dev)
        if v.shape != k.shape:
            raise BrushHipError("radix_argsort: input keys and values must have the same number of elements")
    if sorting_bits > 32:
        raise BrushHipError("radix_argsort: can only sort up to 32 bits")
    ok, ov = torch.empty_like(k), torch.empty_like(k)
    ctx.check(ctx.lib.bh_radix_argsort(ctx._h, _ptr(k), _ptr(v) if v is not None else None, k.numel(), int(sorting_bits), _ptr(ok), _ptr(ov)))
    return ok, ov


def tile_sort_offsets(tile_ids, compact_gids, num_tiles: int, ctx: Optional[Context] = None):
    """The forward's tile sort + offsets table as one operator (render.rs:228-243 + get_tile_offset.rs:11-58): (tile id, compact
    splat id) pairs in depth order -> (tile_ids_sorted, compact_gids_sorted, tile_offsets [num_tiles, 2]); stable."""
    dev = tile_ids.device if isinstance(tile_ids, torch.Tensor) and tile_ids.is_cuda else torch.device("cuda", torch.cuda.current_device())
    ctx = ctx or get_context(dev)
    k, v = _as_u32(tile_ids, dev), _as_u32(compact_gids, dev)
    if k.dim() != 1 or v.shape != k.shape:
        raise BrushHipError("tile_sort_offsets: tile ids and splat ids must be 1-D and of the same length")
    ok, ov = torch.empty_like(k), torch.empty_like(k)
    offs = torch.empty((int(num_tiles), 2), dtype=torch.int32, device=dev)
    ctx.check(ctx.lib.bh_tile_sort_offsets(ctx._h, _ptr(k), _ptr(v), k.numel(), int(num_tiles), _ptr(ok), _ptr(ov), _ptr(offs)))
    return ok, ov, offs


def prefix_sum(x, ctx: Optional[Context] = None):
    """Inclusive u32 prefix sum (brush-prefix-sum/src/lib.rs:11)."""
    dev = x.device if isinstance(x, torch.Tensor) and x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    ctx = ctx or get_context(dev)
    a = _as_u32(x, dev)
    o = torch.empty_like(a)
    ctx.check(ctx.lib.bh_prefix_sum(ctx._h, _ptr(a), a.numel(), _ptr(o)))
    return o


def _loss_cfg(l1_weight, ssim_weight, composite_bg, mask):
    cfg = _ffi.BhLossConfig()
    cfg.l1_weight, cfg.ssim_weight = float(l1_weight), float(ssim_weight)
    bg = composite_bg if composite_bg is not None else (0.0, 0.0, 0.0)
    cfg.bg[0], cfg.bg[1], cfg.bg[2] = [float(b) for b in bg]
    cfg.composite_bg = 1 if composite_bg is not None else 0
    cfg.mask = 1 if mask else 0
    return cfg


def image_loss(pred_hwc, gt_packed, l1_weight=0.8, ssim_weight=-0.2, composite_bg=None, mask=False, ctx: Optional[Context] = None):
    """Per-pixel l1_w*|pred-gt| + ssim_w*SSIM loss map [H,W,C] (C=3, or 4 with the
    alpha-match plane); gt_packed [H,W] rgba8 as int32. brush-loss/src/lib.rs:1075-1104."""
    dev = pred_hwc.device
    ctx = ctx or get_context(dev)
    h, w, c = pred_hwc.shape
    pred_chw = pred_hwc.permute(2, 0, 1).contiguous().float()  # lib.rs:1076
    gt = _as_u32(gt_packed, dev).reshape(h, w)
    out = torch.empty_like(pred_chw)
    cfg = _loss_cfg(l1_weight, ssim_weight, composite_bg, mask)
    ctx.check(ctx.lib.bh_image_loss_forward(ctx._h, _ptr(pred_chw), _ptr(gt), c, h, w, C.byref(cfg), _ptr(out)))
    return out.permute(1, 2, 0).contiguous()


def image_loss_backward(pred_hwc, gt_packed, dl_dmap_hwc, l1_weight=0.8, ssim_weight=-0.2, composite_bg=None, mask=False,
                        ctx: Optional[Context] = None):
    dev = pred_hwc.device
    ctx = ctx or get_context(dev)
    h, w, c = pred_hwc.shape
    pred_chw = pred_hwc.permute(2, 0, 1).contiguous().float()
    dl = dl_dmap_hwc.permute(2, 0, 1).contiguous().float()
    gt = _as_u32(gt_packed, dev).reshape(h, w)
    out = torch.empty_like(pred_chw)
    cfg = _loss_cfg(l1_weight, ssim_weight, composite_bg, mask)
    ctx.check(ctx.lib.bh_image_loss_backward(ctx._h, _ptr(pred_chw), _ptr(gt), _ptr(dl), c, h, w, C.byref(cfg), _ptr(out)))
    return out.permute(1, 2, 0).contiguous()


def image_loss_value_and_grad(img_hwc4, gt_packed, l1_weight=0.8, ssim_weight=-0.2, composite_bg=None, mask=False,
                              alpha_weight=0.0, ctx: Optional[Context] = None):
    """Fused train-step loss: returns (loss [1] device tensor, dloss/dimg [H,W,4]).  Equivalent to
    mean(image_loss(...)) (+ alpha term) and its gradient (train.rs:227-260), in two kernels."""
    dev = img_hwc4.device
    ctx = ctx or get_context(dev)
    h, w, c = img_hwc4.shape
    if c != 4:
        raise ValueError("img must be [H,W,4]")
    img = img_hwc4.contiguous().float()
    gt = _as_u32(gt_packed, dev).reshape(h, w)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    v_out = torch.empty_like(img)
    cfg = _loss_cfg(l1_weight, ssim_weight, composite_bg, mask)
    ctx.check(ctx.lib.bh_image_loss_value_and_grad(ctx._h, _ptr(img), _ptr(gt), h, w, C.byref(cfg), float(alpha_weight), _ptr(loss), _ptr(v_out)))
    return loss, v_out


def gather_stats(refine_weight_norm, vis_weight, max_screen_size, refine_weight, visible, screen_radius, ctx: Optional[Context] = None):
    """RefineRecord::gather_stats (brush-train/src/stats.rs:40-50), in place on the three running [N] tensors: maxima of the refine
    weight and the screen radius, sum of the visibility flags."""
    ctx = ctx or get_context(refine_weight.device)
    n = refine_weight.numel()
    ctx.check(ctx.lib.bh_gather_stats(ctx._h, _ptr(refine_weight_norm), _ptr(vis_weight), _ptr(max_screen_size), _ptr(refine_weight), _ptr(visible),
                                      _ptr(screen_radius), n))


def adam_step(param, grad, m1, m2, lr, t, col_scale=None, reduce_m2=False, beta1=0.9, beta2=0.999, eps=1e-15, ctx: Optional[Context] = None):
    """In-place AdamScaled step on a [rows, ...] parameter (adam_scaled.rs:75-147)."""
    ctx = ctx or get_context(param.device)
    rows = param.shape[0]
    row_len = param.numel() // rows if rows else 1
    ctx.check(ctx.lib.bh_adam_step(ctx._h, _ptr(param), _ptr(grad), _ptr(m1), _ptr(m2), rows, row_len,
                                   _ptr(col_scale) if col_scale is not None else None, float(lr), int(t), int(bool(reduce_m2)),
                                   float(beta1), float(beta2), float(eps)))


# ---------------------------------------------------------------------------
# training (brush-train)
# ---------------------------------------------------------------------------
@dataclass
class TrainConfig:
    """Subset of brush-train/src/config.rs:7-132 that defines step(); same defaults."""
    total_train_iters: int = 30000
    lr_mean: float = 2e-5
    lr_mean_end: float = 2e-7
    lr_coeffs_dc: float = 2e-3
    lr_coeffs_sh_scale: float = 10.0
    lr_opac: float = 0.012
    lr_scale: float = 5e-3
    lr_rotation: float = 2e-3
    ssim_weight: float = 0.2
    match_alpha_weight: float = 0.1
    mean_noise_weight: float = 50.0
    background_color: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    background_noise_strength: float = 0.1
    render_mip: bool = False
    # not in the reference: False = the step's forward builds depth-sliced per-tile lists (BH_FLAG_SLICED_LISTS, same results);
    # True = the reference's full lists
    exact_lists: bool = False
    # refine options (config.rs:47-86)
    max_splats: int = 10_000_000
    refine_every: int = 200
    growth_grad_threshold: float = 0.0025
    growth_select_fraction: float = 0.25
    growth_stop_iter: int = 15000
    split_at_screen_size: float = 0.5
    opac_decay: float = 0.004


@dataclass
class RefineStats:
    """brush-train/src/msg.rs RefineStats (+ the number of splits drawn to refill the pruned budget)."""
    num_added: int
    num_split_oversized: int
    num_split_high_grad: int
    num_pruned: int
    num_pruned_non_finite: int
    total_splats: int
    num_resampled: int = 0


BOUND_PERCENTILE = 0.8  # train.rs:30


def splat_bounds(splats, percentile=BOUND_PERCENTILE, ctx=None):
    """get_splat_bounds (train.rs:124-133, splat_init.rs:130-160): (center[3], extent[3]) of the
    per-axis percentile box of the means."""
    ctx = ctx or get_context(splats.device)
    c, e = (C.c_float * 3)(), (C.c_float * 3)()
    ctx.check(ctx.lib.bh_splat_bounds(ctx._h, _ptr(splats.transforms), splats.num_splats(), float(percentile), c, e))
    return tuple(c), tuple(e)


def bounds_median_size(extent):
    """BoundingBox::median_size (bounding_box.rs:23-29)."""
    return sorted(float(x) for x in extent)[1] * 2.0


@dataclass
class SceneBatch:
    """brush-dataset/src/scene.rs:139-147: packed rgba8 GT [H,W] + camera."""
    img_packed: torch.Tensor
    camera: Camera
    has_alpha: bool = False
    alpha_is_mask: bool = False
    view_id: int = 0   # which view of the dataset this is (index + 1; 0 = unknown): keys the per-tile depth cuts (BhTrainBatch.view_id)

    def img_size(self):
        return tuple(self.img_packed.shape)  # (h, w)


@dataclass
class TrainStepStats:
    num_visible: int
    num_intersections: int
    lr_mean: float
    loss: float
    exchange_rows: int = 0  # mask-keyed exchange: gradient rows sent this step (0 = the dense block)


class BatchUploader:
    """Ring of pinned staging slots + a copy stream that turns decoded host images into packed rgba8
    device batches while the previous batch trains (bh_uploader_*: view_to_packed_data of
    brush-dataset/src/scene.rs:97-136 on the device, the hand-off of scene_loader.rs:59-174)."""

    def __init__(self, max_pixels, slots=3, ctx: Optional[Context] = None):
        self.ctx = ctx or get_context()
        self.lib = self.ctx.lib
        self._h = C.c_void_p(self.lib.bh_uploader_create(self.ctx._h, int(max_pixels), int(slots)))
        if not self._h:
            raise BrushHipError("bh_uploader_create failed (max_pixels > 0, 2 <= slots <= 16, enough pinned memory)")
        self.max_pixels, self.slots = int(max_pixels), int(slots)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.bh_uploader_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise BrushHipError("uploader error %d: %s" % (rc, self.lib.bh_uploader_last_error(self._h).decode()))
        return rc

    def map(self, nbytes):
        """-> (slot, writable uint8 numpy view of the slot's pinned buffer): decode straight into it, then commit()."""
        import numpy as np
        p = C.c_void_p()
        slot = self._check(self.lib.bh_uploader_begin(self._h, int(nbytes), C.byref(p)))
        buf = (C.c_uint8 * int(nbytes)).from_address(p.value)
        return slot, np.frombuffer(buf, dtype=np.uint8)

    def commit(self, slot, w, h, channels, premultiply):
        self._check(self.lib.bh_uploader_commit(self._h, int(slot), int(w), int(h), int(channels), int(bool(premultiply))))

    def submit(self, img_u8, premultiply=True):
        """img_u8: C-contiguous uint8 [H,W,3] or [H,W,4] host array.  `premultiply` applies to RGBA views
        (AlphaMode::Transparent); returns the slot index."""
        import numpy as np
        a = np.ascontiguousarray(img_u8, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] not in (3, 4):
            raise ValueError("image must be [H,W,3] or [H,W,4] uint8")
        h, w, c = a.shape
        return self._check(self.lib.bh_uploader_submit(self._h, a.ctypes.data_as(C.c_void_p), w, h, c, int(bool(premultiply) and c == 4)))

    def acquire(self, slot):
        """-> (packed int32 [H,W] device tensor aliasing the slot, has_alpha).  Work queued on the ctx stream
        afterwards is ordered behind the upload; call release(slot) once the step that reads it is queued."""
        p, w, h, ha = C.c_void_p(), C.c_uint32(), C.c_uint32(), C.c_int()
        self._check(self.lib.bh_uploader_acquire(self._h, int(slot), C.byref(p), C.byref(w), C.byref(h), C.byref(ha)))
        return _view(p.value, (int(h.value), int(w.value)), torch.int32, self.ctx.device), bool(ha.value)

    def release(self, slot):
        self._check(self.lib.bh_uploader_release(self._h, int(slot)))


class SceneLoader:
    """SceneLoader (brush-dataset/src/scene_loader.rs:59-174): an endless shuffled stream of SceneBatch over
    a list of views, prefetched by a loader thread through a BatchUploader so the H2D copy + packing of the
    next views overlap the current train step.

    views: sequence of (image, Camera[, alpha_is_mask]) with image a uint8 [H,W,3|4] array or a callable
    returning one (the decode).  Shuffling: every epoch is a seeded Fisher-Yates permutation (SplitMix64) of
    this rank's views — the reference's order comes from rand::StdRng inside racing loader tasks and is not
    reproducible, so only the "every view once per epoch" property is kept.  `rank`/`world` shard the view
    list for data-parallel training (view i belongs to rank i % world)."""

    def __init__(self, views, seed=0, uploader: Optional[BatchUploader] = None, slots=3, rank=0, world=1, ctx: Optional[Context] = None):
        import queue
        import threading
        self.views = [v for i, v in enumerate(views) if i % world == rank]
        self.view_ids = [i + 1 for i in range(len(views)) if i % world == rank]   # dataset index + 1 (0 = "unknown view")
        if not self.views:
            raise ValueError("Need at least one view in dataset")  # scene_loader.rs:130
        self._own_uploader = uploader is None
        if uploader is None:
            mp = 0
            for v in self.views:
                img = v[0]() if callable(v[0]) else v[0]
                mp = max(mp, img.shape[0] * img.shape[1])
            uploader = BatchUploader(mp, slots, ctx)
        self.up = uploader
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._q = queue.Queue(maxsize=max(1, self.up.slots - 1))  # slots in flight = queued + the one being trained on
        self._stop = threading.Event()
        self._held = None
        self._thread = threading.Thread(target=self._run, name="brush-hip-loader", daemon=True)
        self._thread.start()

    @staticmethod
    def _splitmix(state):
        state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return state, z ^ (z >> 31)

    def epoch_order(self, epoch):
        """The view order of `epoch` (deterministic in seed, epoch and the shard)."""
        n = len(self.views)
        order = list(range(n))
        st = (self._seed ^ (0xD1B54A32D192ED03 * (epoch + 1))) & 0xFFFFFFFFFFFFFFFF
        for i in range(n - 1, 0, -1):
            st, r = self._splitmix(st)
            j = r % (i + 1)
            order[i], order[j] = order[j], order[i]
        return order

    def _run(self):
        import queue
        epoch = 0
        try:
            while not self._stop.is_set():
                for idx in self.epoch_order(epoch):
                    view = self.views[idx]
                    img = view[0]() if callable(view[0]) else view[0]
                    mask = bool(view[2]) if len(view) > 2 else False
                    # wait for a free place in the queue BEFORE mapping a slot, so a mapped slot is never parked
                    while not self._stop.is_set() and self._q.full():
                        self._stop.wait(0.0005)
                    if self._stop.is_set():
                        return
                    slot = self.up.submit(img, premultiply=not mask)
                    self._q.put((slot, idx, view[1], mask))
                epoch += 1
        except Exception as e:  # surface loader failures to the consumer ("Scene loader failed to load an image")
            try:
                self._q.put(e, timeout=1.0)
            except queue.Full:
                pass

    def next_batch(self):
        """-> SceneBatch (its img_packed aliases an uploader slot that stays valid until the NEXT next_batch call).
        Call after queuing the train step of the previous batch: that is what releases its slot."""
        if self._held is not None:
            self.up.release(self._held)
            self._held = None
        item = self._q.get()
        if isinstance(item, Exception):
            raise BrushHipError("Scene loader failed to load an image: %r" % (item,))
        slot, idx, cam, mask = item
        packed, has_alpha = self.up.acquire(slot)
        self._held = slot
        b = SceneBatch(packed, cam, has_alpha=has_alpha, alpha_is_mask=mask, view_id=self.view_ids[idx])
        b.view_index = idx
        return b

    def close(self):
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()
        except Exception:
            pass
        self._thread.join(timeout=5.0)
        if self._own_uploader:
            self.up.close()


class SplatTrainer:
    """SplatTrainer::{new, step} (brush-train/src/train.rs:140-429). Owns the three
    Adam states and the RefineRecord; `step` runs forward, L1+SSIM loss, backward,
    statistics, Adam and the optional mean noise inside one C-ABI call.

    Data parallel (not in the reference): pass `process_group` and the
    per-rank gradients + visible flags are summed with ONE torch.distributed all_reduce
    (RCCL over xGMI) between backward and Adam, the gradients scaled by 1/world inside the
    update; every rank applies the identical update.  The RefineRecord's running maxima stay
    rank-local until `sync_refine_stats()` (called by `refine`) MAX-reduces them."""

    def __init__(self, config: TrainConfig, median_scene_scale: float = 1.0, process_group=None, ctx: Optional[Context] = None,
                 partition: str = "cameras", native_comm: bool = False, sparse_exchange: bool = True, seed: Optional[int] = None,
                 allreduce: str = "ring"):
        """seed: an int turns on the two stochastic terms of the reference's step — the visibility-gated noise on the
        means (train.rs:389-416) and the background jitter (train.rs:896-908) — drawn by the library's counter-based
        generator as pure functions of (seed, step[, splat]); data-parallel ranks must pass the same seed.  None (the
        default of this mirror, which the parity tests rely on): the terms appear only when injected through
        step(background=..., noise_samples=...).

        partition (only with a process_group): "cameras" = data parallel, every rank its own view,
        mean gradient; "tiles" = every rank renders a strip of tile rows of the SAME view, strips are
        all-gathered before the loss and the partial gradients summed."""
        if partition not in ("cameras", "tiles"):
            raise ValueError("partition must be 'cameras' or 'tiles'")
        if allreduce not in ("ring", "direct"):
            raise ValueError("allreduce must be 'ring' (all_reduce / ncclAllReduce) or 'direct' (reduce-scatter + all-gather over point-to-point messages)")
        # how long messages of the gradient exchange are summed: the collective library's all-reduce, or the direct algorithm for a
        # fully connected node (comm.hip comm_allreduce_direct; here, for the hook path, parallel.allreduce_direct).  With
        # native_comm the caller selects it on the context: ctx.set_option("grad_allreduce", "direct")
        self.allreduce = allreduce
        # native_comm: the ctx carries an RCCL communicator (Context.comm_init) and bh_train_step all-reduces the
        # exchange buffer itself — no torch.distributed, no callback (data parallel over cameras only)
        # (partition "tiles": the library also moves the strips' 21-px halos itself — bh_exchange_strip_halos, strip-wise loss only)
        if native_comm and process_group is not None:
            raise ValueError("native_comm excludes process_group")
        self.native_comm = bool(native_comm)
        self.seed = None if seed is None else (int(seed) & 0xFFFFFFFFFFFFFFFF)
        # exchange only the gradient rows of splats some rank (view or strip) saw (BhTrainBatch.exchange_mode 1,
        # brush_amd/csrc/exchange.hip); False = one dense all-reduce of the whole exchange buffer
        self.sparse_exchange = bool(sparse_exchange)
        self.partition = partition
        self._img_hook = None
        self.bounds = None  # (center, extent); None = unit box scaled by median_scene_scale (set by refine / set_bounds)
        self.config = config
        self.median_scene_scale = float(median_scene_scale)
        self.step_count = 0
        self.state = None
        self.ctx = ctx
        self.pg = process_group
        self._hook = None
        self.generator = None
        self.view_cams = []  # [(centre xyz, focal px)] of the train views: enables the Mip-Splatting 3D filter
        # partition == "tiles": strips are re-cut every `rebalance_every` steps so that every rank gets the same number of
        # blended intersections (the backward's work), measured on the previous frame ("balance by
        # intersection count, not rows"); 0 keeps equal-height strips
        self.rebalance_every = 8
        self._row_weights = None
        # partition == "tiles": evaluate the loss strip-wise (each rank on its own rows, 21-px halos from the neighbours)
        # instead of all-gathering the frame and computing the whole loss everywhere; TrainStepStats.loss is then this
        # rank's share of the frame's loss (reduce_loss() sums the shares)
        self.strip_loss = True
        self._strip_loss_now = False
        self.batch_patch = None   # optional callable(BhTrainBatch): edits the C struct right before bh_train_step

    MIN_SCALE_FACTOR = 0.1       # train.rs:44
    MIN_SCALE_FREEZE_FRAC = 0.9  # train.rs:37

    def set_view_cams(self, view_cams):
        """SplatTrainer::set_view_cams (train.rs:170-174): per train view (world centre (x,y,z), focal in px at
        native resolution).  Empty disables the 3D filter."""
        self.view_cams = [(tuple(float(v) for v in c), float(f)) for c, f in view_cams]

    def compute_min_scale(self, splats, ctx=None):
        """compute_min_scale (train.rs:102-125) -> [N] tensor, or None when there are no view cameras."""
        if not self.view_cams or self.MIN_SCALE_FACTOR <= 0.0:
            return None
        ctx = ctx or self.ctx or get_context(splats.device)
        k = len(self.view_cams)
        vc = (C.c_float * (4 * k))()
        for i, (c, f) in enumerate(self.view_cams):
            vc[4 * i], vc[4 * i + 1], vc[4 * i + 2], vc[4 * i + 3] = c[0], c[1], c[2], f
        out = torch.empty(splats.num_splats(), dtype=torch.float32, device=splats.device)
        ctx.check(ctx.lib.bh_compute_min_scale(ctx._h, _ptr(splats.transforms), splats.num_splats(), vc, k, float(self.MIN_SCALE_FACTOR), _ptr(out)))
        return out

    def _init_state(self, splats: Splats):
        dev = splats.device
        n = splats.num_splats()
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)  # noqa: E731
        self.state = dict(m1_t=z(n, 10), m2_t=z(n, 10), m1_sh=z(*splats.sh_coeffs.shape), m2_sh=z(n), m1_o=z(n), m2_o=z(n),
                          refine_weight_norm=z(n), vis_weight=z(n), max_screen_size=z(n))
        if self.ctx is not None and not self.ctx.uses_torch_stream:
            torch.cuda.current_stream(dev).synchronize()   # the zero-fills ran on torch's stream, the step runs on the ctx's own

    def sample_background(self, rng=None, step=None):
        """train.rs:896-908: base + U(-s, s)^3, clamped to [0,1].  With the trainer's seed: the library's generator at
        (seed, step); else from `rng` (a random.Random) if given; else the base colour."""
        s = self.config.background_noise_strength
        base = self.config.background_color
        if self.seed is not None and rng is None:
            out = (C.c_float * 3)()
            _ffi.load().bh_sample_background(self.seed, int(self.step_count + 1 if step is None else step), (C.c_float * 3)(*[float(b) for b in base]), float(s), out)
            return tuple(out)
        if s <= 0.0 or rng is None:
            return tuple(base)
        return tuple(min(1.0, max(0.0, b + (rng.random() * 2.0 - 1.0) * s)) for b in base)

    def normal_samples(self, n, step, device, ctx=None):
        """The [n,3] N(0,1) samples the seeded step number `step` draws (bh_normal_samples)."""
        ctx = ctx or self.ctx or get_context(device)
        out = torch.empty((int(n), 3), dtype=torch.float32, device=device)
        ctx.check(ctx.lib.bh_normal_samples(ctx._h, int(self.seed or 0), int(step), int(n), _ptr(out)))
        return out

    def _make_hook(self, dev):
        import torch.distributed as dist
        pg = self.pg
        world = dist.get_world_size(pg)

        views = {}  # device pointer -> flat float32 view of the longest range summed there so far (only [:count] is touched)

        def hook(_user, exch_ptr, sum_count):
            try:
                cnt = int(sum_count)
                t = views.get(exch_ptr)
                if t is None or t.numel() < cnt:
                    t = views[exch_ptr] = _view(exch_ptr, (cnt,), torch.float32, dev)
                if self.allreduce == "direct" and cnt >= DIRECT_ALLREDUCE_MIN_FLOATS:
                    allreduce_direct(t, cnt, pg)
                else:
                    allreduce_exchange(t, cnt, pg)
                return 0
            except Exception:  # never unwind across the C boundary
                return 1
        self._world = world
        return _ffi.GRAD_HOOK(hook)

    def _make_image_hook(self, dev):
        pg = self.pg

        def hook(_user, img_ptr, h, w, r0, r1):
            try:
                img = _view(img_ptr, (int(h), int(w), 4), torch.float32, dev)
                import torch.distributed as dist
                spans = strip_spans_px(int(h), dist.get_world_size(pg), self._row_weights)
                if self._strip_loss_now:
                    exchange_strip_halos(img, spans, dist.get_rank(pg), pg)
                else:
                    allgather_strips(img, int(r0), int(r1), pg, spans=spans)
                return 0
            except Exception:  # never unwind across the C boundary
                return 1
        return _ffi.IMAGE_HOOK(hook)

    def step(self, batch: SceneBatch, splats: Splats, background=None, noise_samples=None) -> Tuple[Splats, TrainStepStats]:
        """One optimisation step, in place on `splats`. `background` / `noise_samples`
        [N,3] inject the two stochastic terms; None = base colour / no noise."""
        ctx = self.ctx or get_context(splats.device)
        dev = splats.device
        if self.state is None:
            self._init_state(splats)
        cfg = _ffi.BhTrainConfig()
        c = self.config
        cfg.lr_mean, cfg.lr_mean_end, cfg.total_train_iters = c.lr_mean, c.lr_mean_end, c.total_train_iters
        cfg.lr_coeffs_dc, cfg.lr_coeffs_sh_scale, cfg.lr_opac = c.lr_coeffs_dc, c.lr_coeffs_sh_scale, c.lr_opac
        cfg.lr_scale, cfg.lr_rotation, cfg.ssim_weight = c.lr_scale, c.lr_rotation, c.ssim_weight
        cfg.match_alpha_weight, cfg.mean_noise_weight = c.match_alpha_weight, c.mean_noise_weight
        cfg.background[0], cfg.background[1], cfg.background[2] = c.background_color
        cfg.median_scene_scale = self.median_scene_scale
        cfg.render_mip = 1 if (c.render_mip or splats.render_mip) else 0
        cfg.exact_lists = 1 if getattr(c, "exact_lists", False) else 0
        cfg.growth_stop_iter = int(c.growth_stop_iter)   # from that step on nobody reads the refine weight (train.rs:589-614): the step stops computing it
        s = self.state
        st = _ffi.BhTrainState()
        st.n, st.sh_degree = splats.num_splats(), splats.sh_degree()
        st.transforms, st.sh_coeffs, st.raw_opacities = splats.transforms.data_ptr(), splats.sh_coeffs.data_ptr(), splats.raw_opacities.data_ptr()
        st.m1_transforms, st.m2_transforms = s["m1_t"].data_ptr(), s["m2_t"].data_ptr()
        st.m1_sh, st.m2_sh = s["m1_sh"].data_ptr(), s["m2_sh"].data_ptr()
        st.m1_opac, st.m2_opac = s["m1_o"].data_ptr(), s["m2_o"].data_ptr()
        st.refine_weight_norm, st.vis_weight, st.max_screen_size = s["refine_weight_norm"].data_ptr(), s["vis_weight"].data_ptr(), s["max_screen_size"].data_ptr()
        st.step_count = self.step_count
        st.min_scale = splats.min_scale.data_ptr() if splats.min_scale is not None else None
        h, w = batch.img_size()
        b = _ffi.BhTrainBatch()
        b.camera = batch.camera if isinstance(batch.camera, _ffi.BhCamera) else batch.camera.uniforms((w, h))
        native_tiles = self.native_comm and self.partition == "tiles" and ctx.comm_world() > 1
        tiles = (self.pg is not None and self.partition == "tiles") or native_tiles
        if tiles:
            if native_tiles:
                t_rank, t_world = ctx.comm_rank(), ctx.comm_world()
            else:
                import torch.distributed as dist
                t_rank, t_world = dist.get_rank(self.pg), dist.get_world_size(self.pg)
            rows = tile_rows_for_rank((h + 15) // 16, t_rank, t_world, self._row_weights)
            cam = _ffi.BhCamera()
            C.memmove(C.byref(cam), C.byref(b.camera), C.sizeof(cam))
            cam.tile_row_begin, cam.tile_row_end = rows
            b.camera = cam
            halo_ok = strips_allow_halo_loss(strip_spans_px(h, t_world, self._row_weights))
            if native_tiles:
                # no hook: the library exchanges the halos over its own communicator (strip-wise loss is the only native mode)
                if not halo_ok:
                    raise BrushHipError("native tile partition needs every strip to be at least 21 pixel rows tall")
                self._strip_loss_now = True
            else:
                if self._img_hook is None:
                    self._img_hook = self._make_image_hook(dev)
                b.image_hook = C.cast(self._img_hook, C.c_void_p)
                self._strip_loss_now = bool(self.strip_loss) and halo_ok
            b.strip_loss = int(self._strip_loss_now)
        gt = _as_u32(batch.img_packed, dev)
        b.gt_packed = gt.data_ptr()
        b.has_alpha, b.alpha_is_mask = int(batch.has_alpha), int(batch.alpha_is_mask)
        bg = background if background is not None else (self.sample_background() if self.seed is not None else c.background_color)
        b.background[0], b.background[1], b.background[2] = [float(v) for v in bg]
        ns = None
        if noise_samples is not None:
            ns = _f32c(noise_samples, dev).reshape(-1, 3)
            b.noise_samples = ns.data_ptr()
        elif self.seed is not None:
            b.device_noise, b.noise_seed = 1, self.seed
        stats = _ffi.BhTrainStats()
        b.exchange_mode = 1 if (self.sparse_exchange and (self.pg is not None or self.native_comm)) else 0
        b.view_id = int(getattr(batch, "view_id", 0)) & 0xFFFFFFFF
        hook, scale = None, 1.0
        if self.native_comm:
            scale = 1.0 if tiles else 1.0 / ctx.comm_world()
        if self.pg is not None:
            if self._hook is None:
                self._hook = self._make_hook(dev)
            hook, scale = self._hook, (1.0 if tiles else 1.0 / self._world)
        if self.batch_patch is not None:   # last word on the BhTrainBatch (callers that partition a frame themselves; tests)
            self.batch_patch(b)
        ctx.check(ctx.lib.bh_train_step(ctx._h, C.byref(cfg), C.byref(st), C.byref(b), C.cast(hook, C.c_void_p) if hook else None, None,
                                        float(scale), C.byref(stats)))
        self.step_count = st.step_count
        self._last_stats = stats
        self._keep = (gt, ns)
        if tiles and self.rebalance_every > 0:
            if self.step_count % self.rebalance_every == 0:
                self._measure_row_weights(ctx, (h + 15) // 16, (w + 15) // 16, dev)
            elif self.step_count == 1:
                self._warm_rebalance((h + 15) // 16, (w + 15) // 16, dev)
        return splats, stats

    def _measure_row_weights(self, ctx, tile_bh, tile_bw, dev):
        """Per tile row: intersections this frame actually blended (the lists' shrunk ends), summed over the ranks'
        strips -> the weights of the next cut.  One small readback + one [tile_bh] all-reduce every rebalance_every steps."""
        out = _ffi.BhRenderOut()
        ctx.check(ctx.lib.bh_last_render_out(ctx._h, C.byref(out)))
        per_row = self._rows_blended(_view(out.tile_offsets, (tile_bh * tile_bw, 2), torch.int32, dev), tile_bh, tile_bw)
        if out.tile_offsets_far:   # depth-sliced lists: a tile's blended splats = its near segment + its far segment
            per_row = per_row + self._rows_blended(_view(out.tile_offsets_far, (tile_bh * tile_bw, 2), torch.int32, dev), tile_bh, tile_bw)
        if self.native_comm:
            per_row = per_row.contiguous()
            if not ctx.uses_torch_stream:
                torch.cuda.current_stream(dev).synchronize()
            ctx.allreduce_sum(per_row)
            ctx.sync()
        else:
            import torch.distributed as dist
            dist.all_reduce(per_row, op=dist.ReduceOp.SUM, group=self.pg)
        self._row_weights = [float(x) + 1.0 for x in per_row.tolist()]  # +1: empty rows still cost a launch slot

    @staticmethod
    def _rows_blended(tile_offsets, tile_bh, tile_bw):
        to = tile_offsets.to(torch.int64)
        return (to[:, 1] - to[:, 0]).clamp(min=0).view(tile_bh, tile_bw).sum(1).to(torch.float32)

    def _warm_rebalance(self, tile_bh, tile_bw, dev):
        """The first use of each torch kernel above loads its code object (~200 ms in total on ROCm): pay that in the
        first step, not in the middle of training when the first re-cut happens."""
        per_row = self._rows_blended(torch.zeros((tile_bh * tile_bw, 2), dtype=torch.int32, device=dev), tile_bh, tile_bw)
        if not self.native_comm:
            import torch.distributed as dist
            dist.all_reduce(per_row, op=dist.ReduceOp.SUM, group=self.pg)
        per_row.tolist()

    def _train_state(self, splats, s):
        st = _ffi.BhTrainState()
        st.n, st.sh_degree = splats.num_splats(), splats.sh_degree()
        st.transforms, st.sh_coeffs, st.raw_opacities = splats.transforms.data_ptr(), splats.sh_coeffs.data_ptr(), splats.raw_opacities.data_ptr()
        st.m1_transforms, st.m2_transforms = s["m1_t"].data_ptr(), s["m2_t"].data_ptr()
        st.m1_sh, st.m2_sh = s["m1_sh"].data_ptr(), s["m2_sh"].data_ptr()
        st.m1_opac, st.m2_opac = s["m1_o"].data_ptr(), s["m2_o"].data_ptr()
        st.refine_weight_norm, st.vis_weight, st.max_screen_size = s["refine_weight_norm"].data_ptr(), s["vis_weight"].data_ptr(), s["max_screen_size"].data_ptr()
        st.step_count = self.step_count
        return st

    def sync_refine_stats(self):
        """Multi-GPU: MAX-reduce the RefineRecord's running maxima (refine_weight_norm,
        max_screen_size) over the ranks; afterwards every replica's RefineRecord is identical.
        Needed once before refine, not per step (brush_amd/parallel.py)."""
        if self.pg is not None and self.state is not None:
            allreduce_refine_maxima(self.state["refine_weight_norm"], self.state["max_screen_size"], self.pg)
        elif self.native_comm and self.state is not None:
            ctx = self.ctx or get_context(self.state["refine_weight_norm"].device)
            ctx.allreduce_max(self.state["refine_weight_norm"])
            ctx.allreduce_max(self.state["max_screen_size"])

    def set_bounds(self, center, extent):
        self.bounds = (tuple(float(x) for x in center), tuple(float(x) for x in extent))
        self.median_scene_scale = bounds_median_size(self.bounds[1])

    def refine(self, iter: int, splats: Splats, seed: Optional[int] = None):
        """SplatTrainer::refine (train.rs:431-663): prune dead / oversized / out-of-bounds / non-finite
        splats, refill the pruned budget by opacity x visibility sampling, split splats that are too big
        on screen or have a high positional gradient, reset their Adam moments, decay opacities and
        recompute the scene bounds.  Returns (new Splats, RefineStats); the trainer's optimizer state
        and RefineRecord are replaced.  `seed` (default: derived from iter) drives every random choice, so
        data-parallel ranks stay identical."""
        ctx = self.ctx or get_context(splats.device)
        dev = splats.device
        if self.state is None:
            raise BrushHipError("Can only refine if refine stats are initialized")  # train.rs:445
        splats.bake_min_scale(ctx)  # train.rs:433-437: refine manipulates the canonical (un-floored) params
        self.sync_refine_stats()
        if self.bounds is None:
            self.set_bounds(*splat_bounds(splats, ctx=ctx))
        c = self.config
        cfg = _ffi.BhRefineConfig()
        cfg.iter, cfg.total_train_iters = int(iter), max(int(c.total_train_iters), 1)
        cfg.growth_stop_iter = min(int(c.growth_stop_iter), int(c.total_train_iters))  # train.rs:150
        cfg.max_splats = int(c.max_splats)
        cfg.growth_grad_threshold, cfg.growth_select_fraction = c.growth_grad_threshold, c.growth_select_fraction
        cfg.split_at_screen_size, cfg.opac_decay = c.split_at_screen_size, c.opac_decay
        for k in range(3):
            cfg.bounds_center[k], cfg.bounds_extent[k] = self.bounds[0][k], self.bounds[1][k]
        cfg.seed = int(seed) if seed is not None else (0x5EED0000 + int(iter))
        st_in = self._train_state(splats, self.state)
        rs = _ffi.BhRefineStats()
        ctx.check(ctx.lib.bh_refine_plan(ctx._h, C.byref(cfg), C.byref(st_in), C.byref(rs)))
        n2 = rs.total_splats
        z = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        coeffs = splats.sh_coeffs.shape[1]
        new = Splats(z(n2, 10), z(n2, coeffs, 3), z(n2), splats.render_mip, dev)
        ns = dict(m1_t=z(n2, 10), m2_t=z(n2, 10), m1_sh=z(n2, coeffs, 3), m2_sh=z(n2), m1_o=z(n2), m2_o=z(n2),
                  refine_weight_norm=z(n2), vis_weight=z(n2), max_screen_size=z(n2))
        st_out = self._train_state(new, ns)
        self.last_refine_plan = {k: _view(ctx.lib.bh_refine_plan_flags(ctx._h, i), (st_in.n,), torch.int32, dev).clone()
                                 for i, k in enumerate(("keep", "new_row", "split", "child_slot"))}
        ctx.check(ctx.lib.bh_refine_apply(ctx._h, C.byref(cfg), C.byref(st_in), C.byref(st_out)))
        self.state = ns
        self.set_bounds(*splat_bounds(new, ctx=ctx))  # train.rs:634
        # train.rs:636-648: recompute the 3D-filter floor for the new positions / count unless frozen
        if float(iter) / float(max(int(c.total_train_iters), 1)) < self.MIN_SCALE_FREEZE_FRAC:
            f = self.compute_min_scale(new, ctx)
            if f is not None:
                new.with_min_scale(f)
        stats = RefineStats(rs.num_added, rs.num_split_oversized, rs.num_split_high_grad, rs.num_pruned, rs.num_pruned_non_finite,
                            rs.total_splats, rs.num_resampled)
        return new, stats

    def reduce_loss(self, stats: "TrainStepStats") -> float:
        """Tile-partitioned frame with the strip-wise loss: every rank holds its strip's share of the frame's loss;
        this sums the shares (a collective: call it on every rank).  Otherwise returns stats.loss."""
        if self.pg is not None and self.partition == "tiles" and self._strip_loss_now:
            import torch.distributed as dist
            t = torch.tensor([stats.loss], dtype=torch.float64, device=self.state["m2_o"].device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
            return float(t.item())
        if self.native_comm and self.partition == "tiles" and self._strip_loss_now:
            ctx = self.ctx or get_context(self.state["m2_o"].device)
            t = torch.tensor([stats.loss], dtype=torch.float32, device=self.state["m2_o"].device)
            if not ctx.uses_torch_stream:
                torch.cuda.current_stream(t.device).synchronize()
            ctx.allreduce_sum(t)
            ctx.sync()
            return float(t.item())
        return stats.loss

    def stats(self, ctx=None) -> TrainStepStats:
        """Resolve the stats of the last step (synchronises)."""
        (ctx or self.ctx or get_context()).sync()
        s = self._last_stats
        return TrainStepStats(s.num_visible, s.num_intersections, s.lr_mean, s.loss, s.exchange_rows)
