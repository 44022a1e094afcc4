"""brush_amd — host-side mirror (Python) of the Brush operator surface for the
splat-rasterizer hot path, above the C ABI of libbrush_hip.so.

Names and argument meaning follow the reference (paths under
/root/reference/crates/):
    Camera            brush-render/src/camera.rs:12-19 (camera_model: kernels/camera_model/mod.rs:31-38)
    fov_to_focal / focal_to_fov   brush-render/src/camera.rs:85-118
    Splats            brush-render/src/gaussian_splats.rs:62-74
    RasterPass        brush-render/src/gaussian_splats.rs:28-48
    render_splats     brush-render/src/gaussian_splats.rs:365-446 (forward / eval)
    render_splats_bwd brush-render/src/bwd/burn_glue.rs:223-311 (+ RenderBackwards::backward :121-182)
    render_splats_diff / RenderNode   the same as an autodiff node: forward now, backward later from its SAVED state (burn_glue.rs:336-371)
    radix_argsort     brush-sort/src/lib.rs:16
    tile_sort_offsets render.rs:228-243 + get_tile_offset.rs:11-58 (the forward's tile sort and offsets table, one operator)
    prefix_sum        brush-prefix-sum/src/lib.rs:11
    image_loss        brush-loss/src/lib.rs:1075-1104
    splat_to_ply / load_splat_from_ply   brush-serde/src/export.rs:179-204, import.rs:166-330 (plain PLY)
    BatchUploader / SceneLoader          brush-dataset/src/scene.rs:97-136, scene_loader.rs:59-174
    SplatTrainer      brush-train/src/train.rs:140-429 (step) and :431-893 (refine)

torch is used only for device memory, streams and torch.distributed; every
computation runs in the hand-written HIP kernels. No CPU fallback exists.
"""
from .host import (  # noqa: F401
    Camera, Context, RasterPass, RenderAux, SplatTrainer, Splats, TrainConfig, SceneBatch,
    get_context, image_loss, image_loss_backward, image_loss_value_and_grad, prefix_sum, radix_argsort, tile_sort_offsets, render_splats,
    render_splats_bwd, adam_step, gather_stats, RefineStats, splat_bounds, bounds_median_size, fov_to_focal, focal_to_fov,
    splat_to_ply, load_splat_from_ply, ply_parse_header, ParseMetadata, BatchUploader, SceneLoader, set_list_slicing, last_list_counts, set_view_id,
    render_splats_diff, RenderNode,
)
from ._ffi import BrushHipError  # noqa: F401
