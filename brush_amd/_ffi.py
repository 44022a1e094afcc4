"""ctypes binding of libbrush_hip.so (include/brush_hip.h).

There is NO fallback path: if the HIP library is missing this module raises, and
every entry point raises on a non-zero status with bh_last_error()'s message.
"""
import ctypes as C
import os

_DIR = os.path.dirname(os.path.abspath(__file__))
# BRUSH_HIP_LIB: developer override used to A/B kernel variants (scripts/ab.sh build); still a HIP build.
LIB_PATH = os.environ.get("BRUSH_HIP_LIB") or os.path.join(_DIR, "libbrush_hip.so")

FLAG_MIP = 1
FLAG_BWD_INFO = 2
FLAG_SMOOTH_CUTOFF = 4
FLAG_SLICED_LISTS = 8

fp = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)


class BhCamera(C.Structure):
    _fields_ = [
        ("vm", C.c_float * 12),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("lim_pos_x", C.c_float), ("lim_pos_y", C.c_float),
        ("lim_neg_x", C.c_float), ("lim_neg_y", C.c_float),
        ("cam_pos", C.c_float * 3),
        ("img_w", C.c_uint32), ("img_h", C.c_uint32),
        ("tile_row_begin", C.c_uint32), ("tile_row_end", C.c_uint32),
        ("model", C.c_uint32), ("dist", C.c_float * 8), ("half_max_render_fov", C.c_float),
    ]


CAMERA_PINHOLE, CAMERA_KANNALA_BRANDT_4, CAMERA_RADIAL_TANGENTIAL_8, CAMERA_THIN_PRISM_FISHEYE = 0, 1, 2, 3


class BhRenderOut(C.Structure):
    _fields_ = [
        ("num_visible", C.c_uint32), ("num_intersections", C.c_uint32),
        ("num_tiles", C.c_uint32), ("tile_bw", C.c_uint32), ("tile_bh", C.c_uint32),
        ("flags", C.c_uint32),
        ("out_img", C.c_void_p), ("out_img_packed", C.c_void_p), ("visible", C.c_void_p),
        ("max_radius", C.c_void_p), ("tile_offsets", C.c_void_p), ("projected", C.c_void_p),
        ("compact_gid_from_isect", C.c_void_p), ("tile_id_from_isect", C.c_void_p),
        ("global_from_compact_gid", C.c_void_p), ("cum_tiles_hit", C.c_void_p),
        ("intersect_counts", C.c_void_p), ("depths_sorted", C.c_void_p),
        ("tile_offsets_far", C.c_void_p), ("list_budget", C.c_uint32), ("num_listed_splats", C.c_uint32),
        ("generation", C.c_uint64),
    ]


class BhHaloOp(C.Structure):
    _fields_ = [("send", C.c_int32), ("peer", C.c_int32), ("row_begin_px", C.c_uint32), ("rows", C.c_uint32)]


class BhLossConfig(C.Structure):
    _fields_ = [("l1_weight", C.c_float), ("ssim_weight", C.c_float), ("bg", C.c_float * 3),
                ("composite_bg", C.c_int32), ("mask", C.c_int32)]


class BhTrainConfig(C.Structure):
    _fields_ = [
        ("lr_mean", C.c_double), ("lr_mean_end", C.c_double), ("total_train_iters", C.c_uint32),
        ("lr_coeffs_dc", C.c_double), ("lr_coeffs_sh_scale", C.c_float), ("lr_opac", C.c_double),
        ("lr_scale", C.c_double), ("lr_rotation", C.c_double), ("ssim_weight", C.c_float),
        ("match_alpha_weight", C.c_float), ("mean_noise_weight", C.c_float), ("background", C.c_float * 3),
        ("median_scene_scale", C.c_float), ("render_mip", C.c_int32), ("exact_lists", C.c_int32), ("growth_stop_iter", C.c_uint32),
    ]


class BhTrainState(C.Structure):
    _fields_ = [
        ("n", C.c_uint32), ("sh_degree", C.c_uint32),
        ("transforms", C.c_void_p), ("sh_coeffs", C.c_void_p), ("raw_opacities", C.c_void_p),
        ("m1_transforms", C.c_void_p), ("m2_transforms", C.c_void_p),
        ("m1_sh", C.c_void_p), ("m2_sh", C.c_void_p),
        ("m1_opac", C.c_void_p), ("m2_opac", C.c_void_p),
        ("refine_weight_norm", C.c_void_p), ("vis_weight", C.c_void_p), ("max_screen_size", C.c_void_p),
        ("step_count", C.c_uint32),
        ("min_scale", C.c_void_p),
    ]


class BhRefineConfig(C.Structure):
    _fields_ = [
        ("iter", C.c_uint32), ("total_train_iters", C.c_uint32), ("growth_stop_iter", C.c_uint32), ("max_splats", C.c_uint32),
        ("growth_grad_threshold", C.c_float), ("growth_select_fraction", C.c_float), ("split_at_screen_size", C.c_float),
        ("opac_decay", C.c_float), ("bounds_center", C.c_float * 3), ("bounds_extent", C.c_float * 3), ("seed", C.c_uint64),
    ]


class BhRefineStats(C.Structure):
    _fields_ = [("num_added", C.c_uint32), ("num_split_oversized", C.c_uint32), ("num_split_high_grad", C.c_uint32),
                ("num_pruned", C.c_uint32), ("num_pruned_non_finite", C.c_uint32), ("total_splats", C.c_uint32),
                ("num_resampled", C.c_uint32)]


class BhTrainBatch(C.Structure):
    _fields_ = [
        ("camera", BhCamera), ("gt_packed", C.c_void_p), ("has_alpha", C.c_int32), ("alpha_is_mask", C.c_int32),
        ("background", C.c_float * 3), ("noise_samples", C.c_void_p), ("device_noise", C.c_int32), ("noise_seed", C.c_uint64),
        ("image_hook", C.c_void_p), ("image_hook_user", C.c_void_p), ("exchange_mode", C.c_int32), ("strip_loss", C.c_int32),
        ("view_id", C.c_uint32),
    ]


class BhTrainStats(C.Structure):
    _fields_ = [("num_visible", C.c_uint32), ("num_intersections", C.c_uint32), ("lr_mean", C.c_double), ("loss", C.c_float),
                ("exchange_rows", C.c_uint32)]


class BhPlyInfo(C.Structure):
    _fields_ = [("num_splats", C.c_uint64), ("sh_degree", C.c_uint32), ("row_floats", C.c_uint32), ("body_offset", C.c_uint64),
                ("render_mode", C.c_int32), ("has_up_axis", C.c_int32), ("up_axis", C.c_float * 3), ("compressed", C.c_int32)]


GRAD_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)
IMAGE_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)

# every symbol include/brush_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "bh_create": (C.c_void_p, [C.c_int, C.c_void_p, C.c_int]),
    "bh_destroy": (None, [C.c_void_p]),
    "bh_last_error": (C.c_char_p, [C.c_void_p]),
    "bh_sync": (C.c_int, [C.c_void_p]),
    "bh_version": (C.c_char_p, []),
    "bh_abi_version": (C.c_uint32, []),
    "bh_struct_size": (C.c_uint32, [C.c_uint32]),
    "bh_last_list_counts": (C.c_int, [C.c_void_p, u32p, u32p]),
    "bh_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "bh_option_count": (C.c_int, []),
    "bh_option_name": (C.c_char_p, [C.c_int]),
    "bh_option_help": (C.c_char_p, [C.c_int]),
    "bh_camera_setup": (C.c_int, [fp, fp, C.c_double, C.c_double, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.POINTER(BhCamera)]),
    "bh_camera_setup_model": (C.c_int, [fp, fp, C.c_double, C.c_double, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, fp, C.POINTER(BhCamera)]),
    "bh_fov_to_focal": (C.c_double, [C.c_double, C.c_uint32, C.c_uint32, fp]),
    "bh_focal_to_fov": (C.c_double, [C.c_double, C.c_uint32, C.c_uint32, fp]),
    "bh_render_forward": (C.c_int, [C.c_void_p, C.POINTER(BhCamera), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, fp, C.c_uint32, C.POINTER(BhRenderOut)]),
    "bh_set_list_slicing": (C.c_int, [C.c_void_p, C.c_float]),
    "bh_set_view_id": (C.c_int, [C.c_void_p, C.c_uint32]),
    "bh_set_list_cut_threshold": (C.c_int, [C.c_void_p, C.c_uint32]),
    "bh_last_list_share": (C.c_float, [C.c_void_p]),
    "bh_far_slices_queued": (C.c_uint32, [C.c_void_p]),
    "bh_view_table_count": (C.c_uint32, [C.c_void_p]),
    "bh_render_backward": (C.c_int, [C.c_void_p] * 9),
    "bh_render_backward_saved": (C.c_int, [C.c_void_p, C.POINTER(BhRenderOut)] + [C.c_void_p] * 8),
    "bh_render_retain": (C.c_int, [C.c_void_p, C.POINTER(BhRenderOut)]),
    "bh_render_release": (C.c_int, [C.c_void_p, C.POINTER(BhRenderOut)]),
    "bh_last_v_combined": (C.c_void_p, [C.c_void_p]),
    "bh_last_render_out": (C.c_int, [C.c_void_p, C.POINTER(BhRenderOut)]),
    "bh_radix_argsort": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bh_tile_sort_offsets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bh_prefix_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "bh_image_loss_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(BhLossConfig), C.c_void_p]),
    "bh_image_loss_value_and_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(BhLossConfig), C.c_float, C.c_void_p, C.c_void_p]),
    "bh_image_loss_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(BhLossConfig), C.c_void_p]),
    "bh_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_float, C.c_uint32, C.c_int, C.c_float, C.c_float, C.c_float]),
    "bh_gather_stats": (C.c_int, [C.c_void_p] * 7 + [C.c_uint64]),
    "bh_refine_plan": (C.c_int, [C.c_void_p, C.POINTER(BhRefineConfig), C.POINTER(BhTrainState), C.POINTER(BhRefineStats)]),
    "bh_refine_plan_flags": (C.c_void_p, [C.c_void_p, C.c_int]),
    "bh_refine_apply": (C.c_int, [C.c_void_p, C.POINTER(BhRefineConfig), C.POINTER(BhTrainState), C.POINTER(BhTrainState)]),
    "bh_splat_bounds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "bh_fold_min_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bh_fold_min_scale_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bh_compute_min_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, fp, C.c_uint32, C.c_float, C.c_void_p]),
    "bh_splat_to_ply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, fp, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "bh_ply_parse_header": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(BhPlyInfo)]),
    "bh_splats_from_ply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bh_splats_from_ply_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bh_uploader_create": (C.c_void_p, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "bh_uploader_destroy": (None, [C.c_void_p]),
    "bh_uploader_last_error": (C.c_char_p, [C.c_void_p]),
    "bh_uploader_begin": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "bh_uploader_commit": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    "bh_uploader_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    "bh_uploader_acquire": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "bh_uploader_release": (C.c_int, [C.c_void_p, C.c_int]),
    "bh_comm_unique_id": (C.c_int, [C.c_void_p]),
    "bh_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bh_comm_destroy": (C.c_int, [C.c_void_p]),
    "bh_comm_world": (C.c_int, [C.c_void_p]),
    "bh_allreduce_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "bh_allreduce_max_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "bh_allgather_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "bh_forget_views": (C.c_int, [C.c_void_p]),
    "bh_comm_rank": (C.c_int, [C.c_void_p]),
    "bh_comm_selftest": (C.c_int, [C.c_void_p]),
    "bh_strip_halo_plan": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p]),
    "bh_exchange_strip_halos": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "bh_train_step": (C.c_int, [C.c_void_p, C.POINTER(BhTrainConfig), C.POINTER(BhTrainState), C.POINTER(BhTrainBatch), C.c_void_p, C.c_void_p, C.c_float, C.POINTER(BhTrainStats)]),
    "bh_sample_background": (None, [C.c_uint64, C.c_uint32, fp, C.c_float, fp]),
    "bh_normal_samples": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]),
    "bh_philox4x32_10": (None, [u32p, u32p, u32p]),
    "bh_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "bh_profile_fetch": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), fp, u32p, C.c_int]),
}

ABI_VERSION = 7   # the BH_ABI_VERSION of include/brush_hip.h these mirrors were written against
# bh_struct_size index -> mirror (the BH_STRUCT_* order of the header)
STRUCT_MIRRORS = (BhCamera, BhRenderOut, BhLossConfig, BhTrainConfig, BhTrainState, BhTrainBatch, BhTrainStats, BhRefineConfig, BhRefineStats, BhPlyInfo)

_lib = None
_lib_th = None
# the test suite's fault-injection build (csrc/Makefile, -DBH_TEST_HOOKS): never loaded by product code
TEST_HOOKS_LIB_PATH = os.path.join(_DIR, "libbrush_hip_testhooks.so")
TEST_HOOK_SYMBOLS = {"bh_debug_fill_train_scratch": (C.c_int, [C.c_void_p, C.c_uint32]),
                     "bh_debug_split_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)])}


class BrushHipError(RuntimeError):
    pass


def _bind(path, symbols):
    if not os.path.exists(path):
        raise BrushHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    # ABI guard: the library fills these structs with ITS layout; a mirror of another revision would be overrun
    if lib.bh_abi_version() != ABI_VERSION:
        raise BrushHipError("%s speaks ABI %d, this binding ABI %d" % (path, lib.bh_abi_version(), ABI_VERSION))
    for i, mirror in enumerate(STRUCT_MIRRORS):
        if lib.bh_struct_size(i) != C.sizeof(mirror):
            raise BrushHipError("%s: sizeof(%s) is %d in the library, %d in this binding" % (path, mirror.__name__, lib.bh_struct_size(i), C.sizeof(mirror)))
    return lib


def load():
    """Load libbrush_hip.so and bind every declared symbol. Raises if the library is absent."""
    global _lib
    if _lib is None:
        _lib = _bind(LIB_PATH, SYMBOLS)
    return _lib


def load_test_hooks():
    """TESTS ONLY: the fault-injection build of the same sources (a second, independent copy of the library in the process);
    pass it to Context(lib=...)."""
    global _lib_th
    if _lib_th is None:
        _lib_th = _bind(TEST_HOOKS_LIB_PATH, dict(SYMBOLS, **TEST_HOOK_SYMBOLS))
    return _lib_th
