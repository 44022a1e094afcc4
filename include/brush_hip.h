/* brush_hip.h — C ABI of libbrush_hip.so: the MI355X (gfx950) implementation of
 * Brush's differentiable splat rasterizer + training step.
 *
 * This is the drop-in boundary.  Brush (Rust) has no C ABI at this level; its
 * de-facto operator API for the hot path is a set of Rust traits/functions on
 * burn tensor primitives.  Each entry point below replaces the body of one of
 * them (paths relative to /root/reference/crates/):
 *
 *   bh_render_forward   <- <MainBackendBase as SplatOps>::render
 *                          brush-render/src/lib.rs:55-77, render.rs:37-314
 *   bh_render_backward  <- SplatBwdOps::{rasterize_bwd, project_bwd}
 *                          brush-render/src/bwd/burn_glue.rs:62-92, bwd/render_bwd.rs:21-171
 *   bh_radix_argsort    <- brush_sort::radix_argsort          brush-sort/src/lib.rs:16-125
 *   bh_prefix_sum       <- brush_prefix_sum::prefix_sum       brush-prefix-sum/src/lib.rs:11-93
 *   bh_image_loss_*     <- LossOps::{image_loss_forward, image_loss_backward}
 *                          brush-loss/src/lib.rs:718-733 (kernels :181, :371)
 *   bh_adam_step        <- AdamScaled::step                   brush-train/src/adam_scaled.rs:75-147
 *   bh_gather_stats     <- RefineRecord::gather_stats         brush-train/src/stats.rs:40-50
 *   bh_train_step       <- SplatTrainer::step                 brush-train/src/train.rs:176-429
 *   bh_fold_min_scale[_backward] <- fold_min_scale (+ its autodiff)  brush-render/src/gaussian_splats.rs:86-111
 *   bh_compute_min_scale <- compute_min_scale                brush-train/src/train.rs:102-125
 *   bh_uploader_*       <- view_to_packed_data + the SceneLoader hand-off   brush-dataset/src/scene.rs:97-136, scene_loader.rs:59-174
 *   bh_splat_to_ply     <- splat_to_ply                       brush-serde/src/export.rs:86-204
 *   bh_ply_parse_header / bh_splats_from_ply[_strided] <- load_splat_from_ply (plain + SuperSplat-compressed PLY, subsample)  brush-serde/src/import.rs:49-74,166-600
 *   bh_camera_setup[_model] <- Camera::{build_pinhole_params, world_to_local}, fov_to_focal,
 *                          calculate_jacobian_clamp_limits    brush-render/src/camera.rs:63-254
 *                          (+ kernels/camera_model/{pinhole,kannala_brandt_4,radial_tangential_8,thin_prism_fisheye}.rs: pinhole, Kannala-Brandt 4, radial-tangential 8,
 *                          thin-prism fisheye projection / Jacobian / VJP inside the project kernels)
 *
 * Conventions (following the reference's only C ABI, apps/brush-c/src/lib.rs:109-163):
 *   - every function returns 0 on success, <0 on error; nothing throws or
 *     aborts across the boundary; bh_last_error(ctx) gives a host string;
 *   - all pointers are HIP device pointers unless the parameter says "host";
 *   - tensors use the reference's layouts: transforms [N,10] = mean(3),
 *     quat (w,x,y,z) un-normalised (4), log-scale (3); sh_coeffs [N,C,3];
 *     raw_opacities [N] (logits); projected [Nv,9] = xy, conic(3), alpha, rgb;
 *   - a bh_ctx owns one HIP stream and a scratch arena and is single-threaded;
 *     distinct contexts may be used from distinct threads (trainer vs viewer,
 *     one per GPU) — the reference's Actor/stream-per-thread contract
 *     (brush-async/src/lib.rs:1-17);
 *   - calls are asynchronous on the ctx stream except bh_render_forward (it
 *     reads the visible/intersection counts back, like render.rs:146-168),
 *     bh_train_step (calls it) and bh_sync.
 */
#ifndef BRUSH_HIP_H
#define BRUSH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bh_ctx bh_ctx;

/* Error codes */
enum {
    BH_OK = 0,
    BH_ERR_INVALID_ARG = -1,
    BH_ERR_HIP = -2,
    BH_ERR_OOM = -3,
    BH_ERR_STATE = -4,
    BH_ERR_UNSUPPORTED = -5
};

/* Render flags (RasterPass + SplatRenderMode, gaussian_splats.rs:17-48) */
enum {
    BH_FLAG_MIP = 1,           /* SplatRenderMode::Mip */
    BH_FLAG_BWD_INFO = 2,      /* RasterPass::Backward: f32 RGBA out + visible[] + tile-end shrink */
    BH_FLAG_SMOOTH_CUTOFF = 4, /* RasterPass::BackwardSmoothCutoff (test-only C^1 alpha cutoff) */
    /* Not in the reference: build the per-tile lists in two depth slices instead of listing and sorting every (tile, splat)
     * pair (map_gaussians.rs:15-80 + render.rs:228-230).  A tile stops blending once its pixels are saturated
     * (rasterize.rs:116-189), so on scenes that saturate most of the sorted list is never read: the near slice of the depth
     * order is listed, sorted and blended first, and the rest is listed only into tiles that still have live pixels.
     * IDENTICAL to the exact path: out_img / out_img_packed bit for bit, visible[], max_radius, num_visible,
     * num_intersections (K1's count of the exact list), projected rows of every listed splat, and — through
     * bh_render_backward — all gradients and the refine weights (same replay; float-atomic order aside).
     * TRUNCATED under this flag: compact_gid_from_isect / tile_id_from_isect hold the near slice's pairs sorted by tile, then
     * the far slice's; tile_offsets indexes the near part, tile_offsets_far the far part (a tile's blended splats are
     * tile_offsets[t] followed by tile_offsets_far[t]); projected rows of splats that were never listed are undefined.
     * bh_train_step uses it by default (its reference counterpart returns only num_visible and the loss, train.rs:418-427). */
    BH_FLAG_SLICED_LISTS = 8
};

/* Camera models = CameraModel (brush-render/src/kernels/camera_model/mod.rs:31-38).  The reference
 * bakes the distortion parameters into the kernels at JIT time (#[comptime]); here they are data
 * in BhCamera.dist, in the field order of the reference's parameter structs:
 *   BH_CAMERA_KANNALA_BRANDT_4     k1 k2 k3 k4                       (kannala_brandt_4.rs:10-16)
 *   BH_CAMERA_RADIAL_TANGENTIAL_8  k1 k2 k3 k4 k5 k6 p1 p2           (radial_tangential_8.rs:12-22)
 *   BH_CAMERA_THIN_PRISM_FISHEYE   k1 k2 k3 k4 (kb4) p1 p2 sx1 sy1   (thin_prism_fisheye.rs:24-31) */
enum {
    BH_CAMERA_PINHOLE = 0,
    BH_CAMERA_KANNALA_BRANDT_4 = 1,
    BH_CAMERA_RADIAL_TANGENTIAL_8 = 2,
    BH_CAMERA_THIN_PRISM_FISHEYE = 3
};

/* Host-side view uniforms = ProjectUniforms (kernels/types.rs:53-81) without the
 * per-launch counters. */
typedef struct BhCamera {
    float vm[12]; /* world-to-camera 3x4, column-major: col0(x,y,z) col1 col2 translation */
    float fx, fy, cx, cy;
    float lim_pos_x, lim_pos_y, lim_neg_x, lim_neg_y; /* Jacobian clamp limits */
    float cam_pos[3];
    uint32_t img_w, img_h;
    /* Tile-row window [begin, end) of the 16x16 tile grid this call renders (0,0 = the
     * whole image).  Used to partition ONE frame over GPUs by strips of tile rows
     * (SURVEY.md 8e): splats are binned only into the window's tiles and only the window's
     * pixels of out_img are written; per-tile splat lists are identical to a full render. */
    uint32_t tile_row_begin, tile_row_end;
    uint32_t model;            /* BH_CAMERA_* */
    float dist[8];             /* distortion parameters of `model` (unused entries 0) */
    float half_max_render_fov; /* fisheye/distorted models cull on the view angle (render.rs:70-71, project_forward.rs:53-61) */
} BhCamera;

/* Forward outputs = RenderOutput + RenderAuxInner (render_aux.rs:17-68) and the
 * state saved for backward (bwd/burn_glue.rs:336-371).  Device pointers are owned
 * by the ctx and stay valid until the next bh_render_forward on the same ctx or
 * bh_destroy. */
typedef struct BhRenderOut {
    uint32_t num_visible;       /* host scalars */
    uint32_t num_intersections;
    uint32_t num_tiles, tile_bw, tile_bh;
    uint32_t flags;
    float* out_img;                    /* [H,W,4] f32 (BH_FLAG_BWD_INFO) else NULL */
    uint32_t* out_img_packed;          /* [H,W] rgba8 (forward-only) else NULL */
    float* visible;                    /* [N] 1.0 where the splat touched a pixel (BWD_INFO) */
    float* max_radius;                 /* [N] screen radius as a fraction of the image */
    uint32_t* tile_offsets;            /* [T,2] start,end into the isect list */
    float* projected;                  /* [Nv,9] */
    uint32_t* compact_gid_from_isect;  /* [I] sorted by (tile, depth) */
    uint32_t* tile_id_from_isect;      /* [I] sorted */
    uint32_t* global_from_compact_gid; /* [Nv] splat ids front-to-back */
    uint32_t* cum_tiles_hit;           /* [Nv] inclusive scan of per-splat tile counts */
    uint32_t* intersect_counts;        /* [N] tiles hit per splat (0 when culled) */
    float* depths_sorted;              /* [Nv] */
    /* BH_FLAG_SLICED_LISTS and the frame was actually sliced: [T,2] start,end of each tile's far-slice segment (absolute
     * indices into compact_gid_from_isect; 0,0 for tiles the near slice finished); NULL otherwise */
    uint32_t* tile_offsets_far;
    /* pairs the near pass listed (== num_intersections: the lists are the exact ones) */
    uint32_t list_budget;
    /* entries of the compact (depth-ordered) arrays — global_from_compact_gid, depths_sorted, cum_tiles_hit, projected and the
     * compact ids inside compact_gid_from_isect.  == num_visible, except under BH_FLAG_SLICED_LISTS with per-tile cuts: then only
     * the splats that own a listed pair are sorted and numbered (a sub-sequence of the full depth order). */
    uint32_t num_listed_splats;
    /* Which forward of its context this is (1, 2, ...): bh_render_backward_saved / bh_render_retain / bh_render_release identify
     * the forward by it, and refuse (BH_ERR_STATE) a struct whose buffers a later forward has since taken over. */
    uint64_t generation;
} BhRenderOut;

/* ---- ABI guard -------------------------------------------------------------- */
/* The structs of this header are passed by pointer and filled / read with the layout the LIBRARY was built with; a binding
 * built against another revision would be overrun (BhRenderOut, BhTrainBatch and BhTrainConfig have grown).  A binding
 * checks once, at load time: bh_abi_version() == the BH_ABI_VERSION it was written against, and bh_struct_size(i) == the
 * size of its own mirror of struct i (brush_amd/_ffi.py and include/brush_hip.hpp do; INTEGRATION.md shows the Rust side). */
#define BH_ABI_VERSION 7u
enum {
    BH_STRUCT_CAMERA = 0, BH_STRUCT_RENDER_OUT, BH_STRUCT_LOSS_CONFIG, BH_STRUCT_TRAIN_CONFIG, BH_STRUCT_TRAIN_STATE,
    BH_STRUCT_TRAIN_BATCH, BH_STRUCT_TRAIN_STATS, BH_STRUCT_REFINE_CONFIG, BH_STRUCT_REFINE_STATS, BH_STRUCT_PLY_INFO,
    BH_STRUCT_COUNT
};
uint32_t bh_abi_version(void);
uint32_t bh_struct_size(uint32_t which); /* sizeof(struct BH_STRUCT_*) in the library; 0 for an unknown index */

/* ---- context ------------------------------------------------------------- */
/* own_stream != 0: the ctx creates (and owns) a non-blocking stream; `stream` is
 * ignored.  own_stream == 0: submit on the caller's `stream` (a hipStream_t; NULL
 * is the device's default stream), e.g. the host framework's current stream. */
bh_ctx* bh_create(int device, void* stream, int own_stream);
void bh_destroy(bh_ctx* ctx);
const char* bh_last_error(bh_ctx* ctx); /* host string, valid until the next call */
int bh_sync(bh_ctx* ctx);
const char* bh_version(void);

/* ---- options ----------------------------------------------------------------- */
/* The library reads NO environment variable.  Everything earlier revisions took from BH_* variables at bh_create is one key of
 * this setter (value as text; BH_ERR_INVALID_ARG for an unknown key or a value out of range).  Options choose between paths that
 * produce the SAME results — A/B measurements and the tests of the alternative paths — never between results:
 *   cut_min_pairs u32 | cut_margin_pct 0..10000 | cut_margin_fixed 0|1 | cut_ctrl up:down:floor:gap_exp | cut_sort_all 0|1 |
 *   auto_exact_share 0..1 | no_view_hash 0|1 | k16_order 0|1|2 | band_mode 0|1 | k16_waves 0..8 | k16_split 0..1000 | k16_split_min 1..1023 | k16_split_of_max 0..100 | k5_exact_spw 16|32|64 | bwd_jobs 0|1 | no_lpt 0|1 | lpt_classes log|linear | generic_depth_sort 0|1 | dsort_splitters 0|1 |
 *   tile_sort auto|bucket|lsd | spec_k5 0|1 | event_waits 0|1 | readback_copy 0|1 | force_exchange 0|1 | zero_grads 0|1 | loss_bands 0|1 |
 *   update_rows 0|64|128|256 | update_early 0|1 | no_dormant 0|1 | sort_kpt 0|4|8|16 | grad_allreduce ring|direct
 * bh_option_count / bh_option_name / bh_option_help enumerate them with one line of documentation each (host strings). */
int bh_set_option(bh_ctx* ctx, const char* key /*host*/, const char* value /*host*/);
int bh_option_count(void);
const char* bh_option_name(int index);
const char* bh_option_help(int index);

/* ---- camera (host only) --------------------------------------------------- */
/* pos[3], rot_xyzw[4] (glam order), fov in radians (f64 like camera.rs), centre in uv. */
int bh_camera_setup(const float* pos, const float* rot_xyzw, double fov_x, double fov_y, float center_u,
                    float center_v, uint32_t img_w, uint32_t img_h, BhCamera* out /*host*/);
/* Same with a camera model: focal from fov through the model's radial law (camera.rs:85-101), the
 * Jacobian clamp limits of the model (camera.rs:200-254; RT8 inverts its radial law, the fisheye
 * models are not clamped), the view-angle cull bound.  dist: host [8] in the order documented at
 * BH_CAMERA_* (NULL = all zero). */
int bh_camera_setup_model(const float* pos, const float* rot_xyzw, double fov_x, double fov_y, float center_u,
                          float center_v, uint32_t img_w, uint32_t img_h, uint32_t model, const float* dist /*host[8]*/,
                          BhCamera* out /*host*/);
/* fov_to_focal / focal_to_fov (camera.rs:85-118), f64 like the reference; NaN on an unknown model. */
double bh_fov_to_focal(double fov, uint32_t pixels, uint32_t model, const float* dist /*host[8]*/);
double bh_focal_to_fov(double focal, uint32_t pixels, uint32_t model, const float* dist /*host[8]*/);

/* ---- render ---------------------------------------------------------------- */
int bh_render_forward(bh_ctx* ctx, const BhCamera* cam /*host*/, uint32_t n, uint32_t sh_degree,
                      const float* transforms, const float* sh_coeffs, const float* raw_opacities,
                      const float* background /*host [3]*/, uint32_t flags, BhRenderOut* out /*host*/);

/* BH_FLAG_SLICED_LISTS: how the near lists are cut.
 *   near_share <= 0 (the default): PER TILE, from the last frame of the same view on this ctx (bh_set_view_id).  Every blend
 *     launch records, per tile, the depth behind which the tile needed no splat + a margin (1.5x the tile's depth rank more of the
 *     depth order for a view that alternates with one other; deeper the longer the view stays away and after forecasts that
 *     failed, tighter while they hold; "everything" for a tile that did not saturate); the view's next frame lists a (splat,
 *     tile) pair only if the splat lies at or in front of the tile's cut — a fifth to a half of the pairs, whatever the frame
 *     looks like (a blank background or thin regions keep their own short lists whole).  If a tile is still live behind a cut list
 *     the forecast has failed: only the splats in front of the cuts were depth-ordered, so there is nothing to continue from and
 *     the FRAME IS RENDERED AGAIN with complete lists (a second K1 .. blend, ~0.45 ms at 1 M splats / 1080p; the train step then
 *     also evaluates its loss again), which re-seeds the table — about one frame in 100-200 with the adaptive margin.  A view's
 *     first frame, and frames after repeated misses, are rendered with complete lists from the start.
 *   near_share in (0, 1]: ONE cut for the whole frame — the first near_share of the exact list's slots (splats in depth order);
 *     1 = never slice.  Tests and A/B measurements.
 * Results do not depend on the choice, only the time does.  Under a per-tile cut cum_tiles_hit is the scan of the near counts.
 * Note for bh_render_forward: a sliced frame makes the call wait for the near pass's blend (a 4-byte word decides whether the
 * frame is complete); bh_train_step hides that wait behind its loss kernels. */
int bh_set_list_slicing(bh_ctx* ctx, float near_share);
/* The view the following forwards on this ctx render (sticky; 0 = not named, the default): selects the per-tile depth-cut table
 * BH_FLAG_SLICED_LISTS forwards read and refresh (forwards with complete lists refresh it too and take their blend's tile order
 * from it: tiles start in descending order of the work they had at the same view's last frame).  bh_train_step sets it from
 * BhTrainBatch.view_id for its own forward.
 * A frame without an id is keyed by its CAMERA (a hash of the BhCamera's view matrix, intrinsics, size, model and tile window): the
 * views of a dataset are fixed cameras, so a caller that passes the reference's SceneBatch unchanged (no view index,
 * brush-dataset/src/scene.rs:138-147) gets the same tables as one that numbers its views.  One table is 8 bytes per tile (cut +
 * last work); the most recently used 4096 views / 256 MB of tables are kept. */
int bh_set_view_id(bh_ctx* ctx, uint32_t view_id);
/* Drop every per-view table of this ctx (another scene or dataset was loaded: what the tables forecast no longer exists).  Only
 * time depends on it — stale tables cost a few re-rendered frames until they have re-learnt — never results.  Blocking. */
int bh_forget_views(bh_ctx* ctx);
/* Per-tile cuts pay when there are lists to shorten: a view whose last frame had fewer than min_pairs intersections keeps
 * complete lists (default 1 500 000: below that the near count in the projection kernel and an occasional second attempt cost more than
 * listing and sorting everything; 0 = always cut; the test suite, whose scenes are small, sets 0).  Same as option "cut_min_pairs". */
int bh_set_list_cut_threshold(bh_ctx* ctx, uint32_t min_pairs);
/* share the last BH_FLAG_SLICED_LISTS forward on this ctx used (1 = it ran as one slice) */
float bh_last_list_share(bh_ctx* ctx);
/* number of BH_FLAG_SLICED_LISTS forwards on this ctx whose near pass did not finish the frame: second attempts with complete lists
 * (per-tile cuts) or far slices (a fixed near_share).  Diagnostics. */
uint32_t bh_far_slices_queued(bh_ctx* ctx);
/* per-view tables this ctx holds right now (8 bytes per tile each).  Training frames and frames that name their view always get one;
 * a forward-only frame keyed by its camera (a viewer's free camera, an eval render) gets one only from its camera's SECOND frame on,
 * and at most 32 such tables exist at a time: a moving camera allocates nothing.  Diagnostics. */
uint32_t bh_view_table_count(bh_ctx* ctx);

/* How many pairs the last forward on this ctx actually LISTED: compact_gid_from_isect / tile_id_from_isect hold near_pairs
 * entries sorted by tile, then far_pairs entries sorted by tile; everything behind near_pairs + far_pairs is undefined.
 * Exact lists (no BH_FLAG_SLICED_LISTS, or a frame that chose one slice): near_pairs = num_intersections, far_pairs = 0.
 * Blocking (the counts live on the device: one 16-byte readback); BH_ERR_STATE without a forward. */
int bh_last_list_counts(bh_ctx* ctx, uint32_t* near_pairs /*host*/, uint32_t* far_pairs /*host*/);

/* Backward of the LAST BH_FLAG_BWD_INFO forward on this ctx (shorthand for bh_render_backward_saved with that forward's
 * BhRenderOut; a caller that may have rendered something else in between uses the _saved form).  v_output [H,W,4].
 * All four outputs are dense and fully overwritten (zero where the splat got no
 * gradient).  v_refine_weight replaces the reference's "gradient of a dummy [1]
 * tensor" side channel (bwd/burn_glue.rs:165-180). */
int bh_render_backward(bh_ctx* ctx, const float* v_output, const float* transforms, const float* sh_coeffs,
                       const float* raw_opacities, float* v_transforms /*[N,10]*/, float* v_sh_coeffs /*[N,C,3]*/,
                       float* v_raw_opacities /*[N]*/, float* v_refine_weight /*[N]*/);
/* The same with the forward's saved state passed explicitly — the shape of SplatBwdOps::{rasterize_bwd, project_bwd}, which
 * receive the tensors RenderBackwards saved (bwd/burn_glue.rs:62-92, 336-371, consumed at :121-182).  `saved` is the BhRenderOut a
 * BH_FLAG_BWD_INFO forward on THIS ctx returned.  Valid: the ctx's most recent forward, or any forward that was retained
 * (bh_render_retain) and not yet released.  Anything else — a forward whose buffers a later forward has overwritten, a struct of
 * another ctx — fails with BH_ERR_STATE instead of computing the gradients of the wrong frame.  transforms / sh_coeffs /
 * raw_opacities must be the tensors that forward rendered (the reference clones them into the saved state). */
int bh_render_backward_saved(bh_ctx* ctx, const BhRenderOut* saved /*host*/, const float* v_output, const float* transforms,
                             const float* sh_coeffs, const float* raw_opacities, float* v_transforms /*[N,10]*/,
                             float* v_sh_coeffs /*[N,C,3]*/, float* v_raw_opacities /*[N]*/, float* v_refine_weight /*[N]*/);
/* Keep a forward replayable while later forwards run on the same ctx (two render nodes in one autodiff graph; an eval render
 * between a training forward and its backward): detaches the buffers `out` points into from the ctx's arena — they stay valid, and
 * bh_render_backward_saved(out) keeps working, until bh_render_release(out).  Only the ctx's most recent forward can be retained.
 * Costs no copy; the next forward takes fresh blocks (released ones are recycled, so a steady retain / release cycle allocates
 * nothing).  After bh_train_step, `visible` / `max_radius` of its forward point into the step's own buffers and are not kept. */
int bh_render_retain(bh_ctx* ctx, const BhRenderOut* out /*host*/);
int bh_render_release(bh_ctx* ctx, const BhRenderOut* out /*host*/);
/* The BhRenderOut of the last forward on this ctx (also the one inside bh_train_step); BH_ERR_STATE if there is none. */
int bh_last_render_out(bh_ctx* ctx, BhRenderOut* out /*host*/);
/* [Nv,10] rasterize-backward accumulator of the last bh_render_backward (RasterizeGrads). */
const float* bh_last_v_combined(bh_ctx* ctx);

/* ---- primitives ------------------------------------------------------------ */
/* Stable LSD argsort on the low `bits` bits; vals may be NULL (= 0..n-1). */
int bh_radix_argsort(bh_ctx* ctx, const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t bits,
                     uint32_t* out_keys, uint32_t* out_vals);
/* Inclusive prefix sum (wrapping u32). in == out allowed. */
int bh_prefix_sum(bh_ctx* ctx, const uint32_t* in, uint32_t n, uint32_t* out);
/* The forward's tile sort and its offsets table as ONE operator — what render.rs:228-243 (radix_argsort of the tile ids, sorting
 * bits = bits of num_tiles) followed by get_tile_offset.rs:11-58 produce: `n` (tile id, compact splat id) pairs in depth order ->
 * the same pairs grouped by tile, depth order kept (stable), and tile_offsets[2 t .. 2 t + 1] = [begin, end) of tile t's run
 * (0, 0 for a tile without pairs; the table is written entirely).  A tile id is < num_tiles or the reference's sentinel
 * 0xFFFFFFFF (map_gaussians.rs:73-79): sentinel rows sort behind every tile and get no row.  All pointers device; inputs and outputs must not overlap.  bh_render_forward
 * uses the same code: for 9..16 id bits and up to 16 M pairs that is five launches (a stable pass on the high digit, then the
 * rest in parts of 4096 pairs — a count launch and a place launch — whichever tiles the pairs belong to: a list whose pairs sit in a few
 * consecutive tiles sorts as fast as an even one), otherwise the two LSD passes of bh_radix_argsort and an offsets kernel. */
int bh_tile_sort_offsets(bh_ctx* ctx, const uint32_t* tile_ids, const uint32_t* compact_gids, uint32_t n, uint32_t num_tiles,
                         uint32_t* tile_ids_sorted, uint32_t* compact_gids_sorted, uint32_t* tile_offsets /*[2 * num_tiles]*/);

/* ---- image loss ------------------------------------------------------------ */
typedef struct BhLossConfig {
    float l1_weight, ssim_weight;
    float bg[3];
    int32_t composite_bg; /* gt_eff = gt + (1 - gt.a) * bg */
    int32_t mask;         /* loss *= gt.a */
} BhLossConfig;
/* pred [C,H,W] (C = 3, or 4 for the alpha-match plane), gt_packed [H,W] rgba8, loss_map [C,H,W]. */
int bh_image_loss_forward(bh_ctx* ctx, const float* pred_chw, const uint32_t* gt_packed, uint32_t channels,
                          uint32_t h, uint32_t w, const BhLossConfig* cfg /*host*/, float* loss_map);
int bh_image_loss_backward(bh_ctx* ctx, const float* pred_chw, const uint32_t* gt_packed, const float* dl_dmap,
                           uint32_t channels, uint32_t h, uint32_t w, const BhLossConfig* cfg /*host*/,
                           float* dl_dpred);

/* Fused value-and-gradient of the train-step loss (what SplatTrainer::step composes from
 * image_loss + mean + autodiff, brush-train/src/train.rs:227-260, brush-loss/src/lib.rs:1041-1104):
 *   loss = mean_{H,W,3}(l1_w*|p-g| + ssim_w*SSIM) + alpha_weight * mean_{H,W}|p.a - g.a|   (alpha term iff alpha_weight > 0)
 * img_hwc4 [H,W,4] is the rasterizer output; loss_out is ONE device float; v_output [H,W,4] =
 * dloss/dimg is fully overwritten (alpha channel 0 without the alpha term). */
int bh_image_loss_value_and_grad(bh_ctx* ctx, const float* img_hwc4, const uint32_t* gt_packed, uint32_t h, uint32_t w,
                                 const BhLossConfig* cfg /*host*/, float alpha_weight, float* loss_out, float* v_output);

/* ---- optimizer / stats ----------------------------------------------------- */
/* One AdamScaled step on a [rows,row_len] parameter.  t = state.time after this
 * step (1 on the first call: moments are initialised, not decayed).  col_scale
 * [row_len] device or NULL.  reduce_m2 != 0: second moment is one scalar per row
 * (m2 has `rows` entries).  beta1=.9 beta2=.999 eps=1e-15 in the reference. */
int bh_adam_step(bh_ctx* ctx, float* param, const float* grad, float* m1, float* m2, uint64_t rows,
                 uint32_t row_len, const float* col_scale, float lr, uint32_t t, int reduce_m2, float beta1,
                 float beta2, float eps);
int bh_gather_stats(bh_ctx* ctx, float* refine_weight_norm, float* vis_weight, float* max_screen_size,
                    const float* refine_weight, const float* visible, const float* screen_radius, uint64_t n);

/* ---- Mip-Splatting 3D filter (world-space scale floor) ------------------------ */
/* fold_min_scale (brush-render/src/gaussian_splats.rs:86-111): scales -> sqrt(s^2 + f^2), opacity
 * energy-compensated by sqrt(det1/det2) (clamped to [1e-6, 1-1e-6], returned as a logit).  The
 * reference applies it in front of every render of a Splats with a floor (gaussian_splats.rs:379-386,
 * bwd/burn_glue.rs:260-270) and to bake the floor into the parameters (Splats::bake_min_scale,
 * gaussian_splats.rs:245-256: pass out == in).  transforms [N,10], raw_opac [N], min_scale [N]. */
int bh_fold_min_scale(bh_ctx* ctx, const float* transforms, const float* raw_opacities, const float* min_scale, uint32_t n,
                      float* out_transforms, float* out_raw_opacities);
/* Its VJP w.r.t. the learned log-scales / raw opacity (min_scale is a constant) — what burn's autodiff
 * derives for the fold.  In place: v_transforms [N,10] and v_raw_opacities [N] hold the gradients w.r.t.
 * the FOLDED tensors on entry and w.r.t. the raw parameters on return (columns 0..6 pass through). */
int bh_fold_min_scale_backward(bh_ctx* ctx, const float* transforms, const float* raw_opacities, const float* min_scale, uint32_t n,
                               float* v_transforms, float* v_raw_opacities);
/* compute_min_scale (brush-train/src/train.rs:102-125): out[i] = sqrt(factor) * min_v |mean_i - centre_v| / max(focal_v, 1e-6).
 * view_cams: host [num_views,4] = camera centre xyz + focal length in pixels at native resolution. */
int bh_compute_min_scale(bh_ctx* ctx, const float* transforms, uint32_t n, const float* view_cams /*host*/, uint32_t num_views,
                         float factor, float* out /*[N]*/);

/* ---- training step ---------------------------------------------------------- */
/* TrainConfig subset that defines step() (brush-train/src/config.rs:7-132). */
typedef struct BhTrainConfig {
    double lr_mean, lr_mean_end;  /* 2e-5 -> 2e-7 over total_train_iters */
    uint32_t total_train_iters;   /* 30000 */
    double lr_coeffs_dc;          /* 2e-3 */
    float lr_coeffs_sh_scale;     /* 10: bands >= 1 use lr/10 */
    double lr_opac;               /* 0.012 */
    double lr_scale;              /* 5e-3 */
    double lr_rotation;           /* 2e-3 */
    float ssim_weight;            /* 0.2 */
    float match_alpha_weight;     /* 0.1 */
    float mean_noise_weight;      /* 50; 0 disables the noise term */
    float background[3];          /* base background colour */
    float median_scene_scale;     /* bounds.median_size() */
    int32_t render_mip;
    int32_t exact_lists;          /* 0 (default): the step's forward runs with BH_FLAG_SLICED_LISTS; 1: the reference's full lists */
    /* TrainConfig::growth_stop_iter (config.rs:72, 15000).  The refine weight a step accumulates into RefineRecord::refine_weight_norm
     * is read by ONE consumer, refine()'s growth selection, and only while iter < growth_stop_iter (train.rs:589-614).  From step
     * number growth_stop_iter on, bh_train_step therefore runs its blend backward WITHOUT the refine weight's per-pixel norm (a
     * third of that kernel's gradient block): refine_weight_norm then stays as refine() zeroed it; every gradient, vis_weight and
     * max_screen_size are computed as before.  0 = always compute it (what the reference's step does). */
    uint32_t growth_stop_iter;
} BhTrainConfig;

/* Parameters + optimizer state of one model replica, all device memory owned
 * by the caller (Splats + SplatOptim + RefineRecord). */
typedef struct BhTrainState {
    uint32_t n, sh_degree;
    float* transforms;    /* [N,10] */
    float* sh_coeffs;     /* [N,C,3] */
    float* raw_opacities; /* [N] */
    float* m1_transforms; float* m2_transforms; /* [N,10] each */
    float* m1_sh; float* m2_sh;                 /* [N,C,3], [N] (reduced).  A ZERO entry of m2_sh may carry a negative sign (-0.0f): the
                                                   library's mark "every Adam moment of this splat is zero" (bh_train_step then skips the
                                                   splat while it receives no gradient and is not reached by the view).  -0.0 compares equal
                                                   to 0.0 and behaves like it in the recurrence; zero-filling or overwriting the tensor simply
                                                   removes the marks.  Copy the moment tensors together (checkpoints, refine does). */
    float* m1_opac; float* m2_opac;             /* [N] each */
    float* refine_weight_norm; float* vis_weight; float* max_screen_size; /* [N] each */
    uint32_t step_count; /* number of steps already taken (host; incremented by the call) */
    /* Splats::min_scale (gaussian_splats.rs:69-73): optional frozen per-splat world-space scale floor [N]
     * (Mip-Splatting 3D filter), NULL = none.  bh_train_step renders fold_min_scale(params) and chains the
     * gradients back to the raw parameters; refine bakes it (train.rs:437) and the caller recomputes it
     * with bh_compute_min_scale afterwards (train.rs:636-648). */
    const float* min_scale;
} BhTrainState;

/* Image hook: called (if non-NULL) after the forward render and before the loss with the
 * ctx-owned out_img [H,W,4]; a tile-partitioned caller all-gathers the strips of the other
 * ranks into it (pixel rows [row_begin_px, row_end_px) are this rank's) — or, with
 * BhTrainBatch.strip_loss, just the 21 rows on either side of its strip.  Return 0. */
typedef int (*bh_image_hook)(void* user, float* out_img, uint32_t h, uint32_t w, uint32_t row_begin_px, uint32_t row_end_px);

typedef struct BhTrainBatch {
    BhCamera camera;
    const uint32_t* gt_packed; /* [H,W] rgba8 device */
    int32_t has_alpha;
    int32_t alpha_is_mask;
    /* Stochastic terms are injected so a step is reproducible (the reference draws
     * them from burn's GPU PRNG / rand::rng(), train.rs:395-399,896-908): */
    float background[3];       /* background actually used this step */
    const float* noise_samples; /* [N,3] N(0,1) device: injected samples (parity tests), or NULL */
    /* device_noise != 0 and noise_samples == NULL: the samples are drawn inside the step by a counter-based generator
     * (Philox-4x32-10 + Box-Muller, brush_amd/csrc/device_rng.h) as a pure function of (noise_seed, step number, splat
     * index) — what `Tensor::random(.., Normal(0,1))` is in the reference (train.rs:395-399).  Data-parallel ranks passing
     * the same seed on identical replicas draw identical noise.  Both unset = no noise term. */
    int32_t device_noise;
    uint64_t noise_seed;
    bh_image_hook image_hook;   /* NULL unless the frame is tile-partitioned over ranks */
    void* image_hook_user;
    /* Multi-GPU only.  0: one SUM over visible | gradients (dense).  1: mask-keyed — the hook (or the
     * library's communicator) is called for the visible flags first, then for a compact block holding only the gradient
     * rows of the splats some rank saw (their union is known from the summed flags and identical on every rank); falls
     * back to the dense block when that union exceeds half of the scene.  The result equals mode 0 up to the summation
     * order inside the collective; per view only the splats that reached a pixel carry a gradient, so the message is
     * typically several times smaller.  Costs one more 4-byte readback per step.  Applies to the tile-partitioned frame too
     * (image_hook set): a strip's rows are non-zero only for the splats that reached one of its pixels, the compact rows
     * then carry one more column, the refine weight (the strips' partial sums add up), and the dense fall-back is the whole
     * span behind the visible section. */
    int32_t exchange_mode;
    /* Tile-partitioned frame only (image_hook set).  0: the hook all-gathers the whole image and every rank evaluates the
     * loss on all of it.  1: strip-wise loss — the hook only has to deliver the image rows within 21 px (one tile row + the
     * 5-px SSIM window) above and below this rank's strip; the loss kernels run on the strip (pass A one tile row wider),
     * BhTrainStats.loss is the strip's share of the frame's mean loss (sum the ranks' values). */
    int32_t strip_loss;
    /* Which view of the dataset this batch is (any stable non-zero number, e.g. its index + 1; 0 = unknown).  Only the TIME of a
     * step depends on it: the forward keeps, per view id, how deep every tile had to go the last time that view was rendered
     * and lists only that much of every tile the next time (bh_set_view_id).  Optional: with 0 the table is keyed by the camera
     * itself, which is the same thing for a dataset of fixed views. */
    uint32_t view_id;
} BhTrainBatch;

typedef struct BhTrainStats {
    uint32_t num_visible, num_intersections;
    double lr_mean;
    float loss; /* delivered by the next host wait on this ctx — bh_sync, bh_refine_plan, bh_splat_bounds (or the next
                   bh_train_step, which replaces the pending delivery): the struct must stay alive until one of them has
                   returned.  0 until then, and only the most recent step's stats are completed */
    uint32_t exchange_rows; /* exchange_mode 1: gradient rows in the compact block this step (0 = dense block was sent) */
} BhTrainStats;

/* Exchange hook (multi-GPU callers; not in the reference, which is single-GPU): called (if
 * non-NULL) after the backward and before the statistics / Adam update with the step's ONE
 * exchange buffer on the ctx stream:
 *     visible[N] | v_transforms[10N] | v_sh[3CN] | v_raw_opac[N] | refine_weight[N]
 * (each section padded with zeros to a multiple of 4 floats, so all of them are 16-byte aligned).
 * The caller must SUM the first `sum_count` floats over its ranks, in place:
 *   - data parallel over cameras: sum_count = everything before refine_weight — the per-view visible flags and the
 *     gradients (scaled by `grad_scale` = 1/K inside the update).  refine_weight stays local: the
 *     RefineRecord keeps running MAXima (refine_weight_norm, max_screen_size), which a caller reduces
 *     over ranks with MAX once, before refine — not every step.  vis_weight counts views.
 *   - one frame partitioned by tile rows (image_hook set): sum_count = the whole buffer — the refine
 *     weight is a per-pixel sum, so the strips' partial sums add; `visible` is clamped to 1 afterwards
 *     (exchange_mode 1: the visible section, then the compact rows incl. the refine weight, as above).
 * One buffer = one collective per step (BhTrainBatch.exchange_mode 0; mode 1 calls the hook twice: for the leading visible
 * section, then for a compact scratch block — the contract is always "sum `sum_count` floats at `exchange`").  Return 0. */
typedef int (*bh_grad_hook)(void* user, float* exchange, uint64_t sum_count);

/* ---- collectives inside the library (optional) --------------------------------- */
/* For a host without a collective layer of its own (the Rust pipeline): one RCCL communicator per
 * ctx, one process per GPU.  RCCL is bound at run time (dlopen) — nothing here is needed, or loaded, on one GPU.
 * rank 0 calls bh_comm_unique_id and hands the 128 bytes to every rank by its own means (env, file, socket);
 * every rank calls bh_comm_init.  With a communicator of world > 1 attached and hook == NULL, bh_train_step
 * all-reduces its exchange buffer itself (pass grad_scale = 1/world for the mean over the ranks' views).
 * All collectives are in place and asynchronous on the ctx stream. */
int bh_comm_unique_id(void* out_id /*host, 128 bytes*/);
int bh_comm_init(bh_ctx* ctx, int rank, int world, const void* unique_id /*host, 128 bytes*/);
int bh_comm_destroy(bh_ctx* ctx);
int bh_comm_world(bh_ctx* ctx); /* 1 without a communicator */
int bh_allreduce_sum_f32(bh_ctx* ctx, float* buf, uint64_t count);
int bh_allreduce_max_f32(bh_ctx* ctx, float* buf, uint64_t count); /* e.g. RefineRecord maxima before refine */
int bh_allgather_bytes(bh_ctx* ctx, const void* send, void* recv /*world * bytes_per_rank*/, uint64_t bytes_per_rank); /* e.g. image strips */
int bh_comm_rank(bh_ctx* ctx);  /* 0 without a communicator */
/* Every RCCL entry point the library binds (all-reduce SUM / MAX, all-gather, grouped send / recv) on small rank-dependent
 * patterns, checked on the host; blocking, collective (every rank calls it).  Run it once after bh_comm_init before trusting an
 * exchange: a build whose communicator has never met more than one rank finds out here, not in a gradient.  With option
 * grad_allreduce = direct set on the ctx it also checks the direct all-reduce against ncclAllReduce's sum. */
int bh_comm_selftest(bh_ctx* ctx);
/* One frame split over the ranks by strips of tile rows, strip-wise loss: fetch the 21 pixel rows above and below
 * this rank's strip [row_begin_px, row_end_px) of img [H,W,4] from the neighbouring ranks, and hand them this strip's first / last
 * 21 rows (grouped ncclSend / ncclRecv on the ctx stream, in place).  Preconditions, identical on all ranks: strips lie in rank
 * order (rank r directly above rank r + 1) and each is at least 21 rows tall.  bh_train_step calls this itself when the batch's
 * camera has a tile-row window, strip_loss != 0, no image_hook is given and the ctx carries a communicator.
 * bh_strip_halo_plan is the host arithmetic behind it (a caller with its own transport can reuse it): up to 4 operations, returns
 * their number. */
typedef struct BhHaloOp { int32_t send; int32_t peer; uint32_t row_begin_px, rows; } BhHaloOp;
int bh_strip_halo_plan(uint32_t img_h, uint32_t row_begin_px, uint32_t row_end_px, int rank, int world, BhHaloOp* out /*host [4]*/);
int bh_exchange_strip_halos(bh_ctx* ctx, float* img_hwc4, uint32_t h, uint32_t w, uint32_t row_begin_px, uint32_t row_end_px);

/* The stochastic terms of step().  bh_sample_background: sample_background_color (train.rs:896-908) from the same
 * counter-based generator — base + U(-strength, strength)^3 clamped to [0,1], a pure function of (seed, step); host only.
 * bh_normal_samples: the [n,3] N(0,1) samples a device_noise step with this (seed, step) draws (tests; callers that
 * want the tensor).  bh_philox4x32_10: the raw generator (known-answer tests against the published vectors); host only. */
void bh_sample_background(uint64_t seed, uint32_t step, const float base[3] /*host*/, float strength, float out[3] /*host*/);
int bh_normal_samples(bh_ctx* ctx, uint64_t seed, uint32_t step, uint64_t n, float* out /*[n,3] device*/);
void bh_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* state->step_count is advanced only when the call succeeds (a failed step applied no update and may be retried).
 * The gradients live in a scratch buffer of the ctx (the exchange buffer above).  With a hook or a communicator the whole
 * buffer is defined when it is handed over (rows of splats the view did not use are zero: render_bwd.rs:123-138).  On one
 * GPU without a hook nobody else reads it, and the step clears only its refine-weight vector: the backward marks the splats
 * whose gradient rows it writes (the sign bit of their refine weight), the update ignores whatever the other rows hold and takes
 * them as zero — same results, without 4 (10 + 3C) N bytes of zero-fill per step. */
int bh_train_step(bh_ctx* ctx, const BhTrainConfig* cfg /*host*/, BhTrainState* state /*host*/,
                  const BhTrainBatch* batch /*host*/, bh_grad_hook hook, void* hook_user, float grad_scale,
                  BhTrainStats* stats /*host*/);
#ifdef BH_TEST_HOOKS
/* NOT part of the shipping library: exported only by libbrush_hip_testhooks.so (built with -DBH_TEST_HOOKS, csrc/Makefile), which
 * also reads the fault-injection variables BH_BREAK_ALLREDUCE and BH_TEST_FAIL_LOSS_AT.  Overwrites the step's gradient scratch
 * with a 32-bit pattern (e.g. a NaN) on the ctx stream — a step that then still produces the zero-filling step's results has not
 * read a row it did not write.  BH_ERR_STATE before the first step. */
int bh_debug_fill_train_scratch(bh_ctx* ctx, uint32_t pattern);
#endif

/* ---- refine (densify / prune) ------------------------------------------------ */
/* SplatTrainer::refine (brush-train/src/train.rs:431-893) in two calls, because the caller
 * owns all tensors and must size the outputs: bh_refine_plan decides on the device what is
 * pruned and what is split (one 64-byte readback returns the counts), the caller allocates
 * `total_splats` rows for every tensor, bh_refine_apply writes them.  All stochastic choices
 * derive from `seed`: data-parallel ranks passing the same seed on their (identical) replicas
 * take identical decisions. */
typedef struct BhRefineConfig {
    uint32_t iter, total_train_iters; /* opacity-decay schedule and growth gating */
    uint32_t growth_stop_iter;        /* 15000 */
    uint32_t max_splats;              /* 10000000 */
    float growth_grad_threshold;      /* 0.0025 */
    float growth_select_fraction;     /* 0.25 */
    float split_at_screen_size;       /* 0.5; 0 disables */
    float opac_decay;                 /* 0.004 */
    float bounds_center[3], bounds_extent[3]; /* current scene bounds (train.rs:485,504-512) */
    uint64_t seed;
} BhRefineConfig;

typedef struct BhRefineStats { /* RefineStats (brush-train/src/msg.rs) + the resample count */
    uint32_t num_added, num_split_oversized, num_split_high_grad, num_pruned, num_pruned_non_finite, total_splats;
    uint32_t num_resampled; /* splits drawn to refill the pruned budget */
} BhRefineStats;

int bh_refine_plan(bh_ctx* ctx, const BhRefineConfig* cfg /*host*/, const BhTrainState* state /*host*/, BhRefineStats* out /*host*/);
/* Plan arrays of the last bh_refine_plan, each [N] u32, valid until bh_refine_apply:
 * 0 keep flag, 1 new row of a kept splat, 2 split flag, 3 child slot (= kept count + this). */
const uint32_t* bh_refine_plan_flags(bh_ctx* ctx, int which);
/* `out`: caller-allocated state with n = total_splats of the plan (same sh_degree).  Kept rows are
 * gathered in order, split parents rewritten, children appended (train.rs:665-806), Adam moments of
 * both halves zeroed, opacity decay applied (train.rs:808-817), the RefineRecord zeroed. */
int bh_refine_apply(bh_ctx* ctx, const BhRefineConfig* cfg /*host*/, const BhTrainState* in /*host*/, BhTrainState* out /*host*/);
/* get_splat_bounds / bounds_from_pos (brush-train/src/splat_init.rs:130-160): per-axis percentile
 * box of the means ([N,10] transforms, columns 0..2), non-finite values ignored; blocking. */
int bh_splat_bounds(bh_ctx* ctx, const float* transforms, uint32_t n, float percentile, float* center /*host[3]*/, float* extent /*host[3]*/);

/* ---- PLY at the edges (brush-serde) -------------------------------------------- */
/* splat_to_ply (brush-serde/src/export.rs:179-204): the INRIA-layout binary_little_endian PLY Brush writes —
 * header comments "Exported from Brush", "Vertical axis: ...", "SH degree: d", "SplatRenderMode: mip|default";
 * per splat x y z scale_0..2 opacity rot_0..3 (normalised) f_dc_0..2 f_rest_0..3(C-1)-1 ([channel][coeff]).
 * min_scale (nullable): the 3D-filter floor is baked into the written scales / opacity (export.rs:183).
 * The rows are packed on the device and arrive in `out` with one D2H copy.  out == NULL: size query
 * (*written = bytes needed).  Blocking.  up_axis: host [3] or NULL ("Vertical axis: y"). */
int bh_splat_to_ply(bh_ctx* ctx, const float* transforms, const float* sh_coeffs, const float* raw_opacities, const float* min_scale,
                    uint32_t n, uint32_t sh_degree, int render_mip, const float* up_axis /*host*/, void* out /*host*/, uint64_t cap,
                    uint64_t* written /*host*/);

/* What parse_ply learns from the header (brush-serde/src/import.rs:172-277). */
typedef struct BhPlyInfo {
    uint64_t num_splats;
    uint32_t sh_degree;   /* from the number of f_dc_/f_rest_ properties */
    uint32_t row_floats;  /* properties per vertex row when all are float; 0 for rows of mixed scalar types /
                             a red-green-blue colour override (ply_gaussian.rs:36-99) and for compressed files */
    uint64_t body_offset; /* first byte after end_header */
    int32_t render_mode;  /* -1 unknown, 0 default, 1 mip ("SplatRenderMode:" comment) */
    int32_t has_up_axis;
    float up_axis[3];     /* "Vertical axis:" comment: x -> +X, y -> -Y, z -> -Z, or three numbers */
    int32_t compressed;   /* 1: a SuperSplat / PlayCanvas compressed file (chunk + packed vertex [+ sh] elements,
                             import.rs:407-600, quant.rs); row_floats is 0 then */
} BhPlyInfo;
/* Host only.  BH_ERR_UNSUPPORTED for files this build does not read (ascii / big-endian, vertex properties of
 * types the format does not use); BH_ERR_INVALID_ARG for malformed ones. */
int bh_ply_parse_header(const void* bytes /*host*/, uint64_t len, BhPlyInfo* info /*host*/);
/* load_splat_from_ply + SplatData::into_splats (import.rs:166-170, 57-75): one H2D copy of the body, columns
 * scattered on the device (compressed files: the packed words are decoded on the device, one thread per splat); absent properties take the reference's defaults (rotation 1,0,0,0; log-scale -4;
 * SH DC 0.5; raw opacity 0).  Outputs sized from bh_ply_parse_header: transforms [N,10], sh_coeffs
 * [N,(d+1)^2,3], raw_opacities [N].  Blocking. */
int bh_splats_from_ply(bh_ctx* ctx, const void* bytes /*host*/, uint64_t len, float* transforms, float* sh_coeffs, float* raw_opacities);
/* The same, keeping only file rows first, first + step, ... (`count` of them; UINT64_MAX = all that exist): the
 * `subsample_points` argument of load_splat_from_ply (import.rs:170-181: first = s - 1, step = s, count = rows / s) and
 * SplatData::subsample (import.rs:49-74: first = 0, step = ceil(rows / max_splats)).  The output tensors hold `count` rows. */
int bh_splats_from_ply_strided(bh_ctx* ctx, const void* bytes, uint64_t len, uint64_t first, uint64_t step, uint64_t count,
                                      float* transforms, float* sh_coeffs, float* raw_opacities);

/* ---- host image -> packed device batch (brush-dataset) ---------------------------- */
/* SceneBatch::img_packed producer: view_to_packed_data (brush-dataset/src/scene.rs:97-136) moved to the
 * device behind a ring of pinned staging slots and a copy stream, replacing the depth-4 channel + wgpu
 * staging upload of scene_loader.rs:59-174.  The decoded RGB8 / RGBA8 bytes cross PCIe unpacked (3 or 4
 * B/pixel); widening (a = 255), byte-space premultiply ((c*a + 127) / 255, AlphaMode::Transparent) and
 * packing run in a kernel on the copy stream while the previous batch trains.
 *
 * Threads: begin / commit / submit may be called from ONE loader thread, acquire / release from the thread
 * that owns `ctx` (internally locked).  Life of a slot: begin -> (fill pinned bytes) -> commit -> acquire
 * (the ctx stream waits for the upload on the device; no host block) -> queue the train step -> release
 * (an event on the ctx stream; begin blocks on it only when the ring wraps onto a still-busy slot). */
typedef struct bh_uploader bh_uploader;
bh_uploader* bh_uploader_create(bh_ctx* ctx, uint64_t max_pixels, uint32_t num_slots /*2..16*/);
void bh_uploader_destroy(bh_uploader* up);
const char* bh_uploader_last_error(bh_uploader* up);
/* Map the next slot: returns its index (>= 0) and the pinned host buffer to decode into (`bytes` <= 4*max_pixels). */
int bh_uploader_begin(bh_uploader* up, uint64_t bytes, void** pinned /*host out*/);
/* Queue H2D + pack of a mapped slot. channels 3 = RGB8, 4 = RGBA8 (tightly packed rows). */
int bh_uploader_commit(bh_uploader* up, int slot, uint32_t w, uint32_t h, uint32_t channels, int premultiply);
/* begin + memcpy + commit for pixels that already live elsewhere; returns the slot index. */
int bh_uploader_submit(bh_uploader* up, const uint8_t* pixels /*host*/, uint32_t w, uint32_t h, uint32_t channels, int premultiply);
int bh_uploader_acquire(bh_uploader* up, int slot, const uint32_t** packed /*device [H,W] rgba8*/, uint32_t* w, uint32_t* h, int* has_alpha);
int bh_uploader_release(bh_uploader* up, int slot);

/* ---- profiling --------------------------------------------------------------- */
/* on = 1: every pipeline stage is bracketed by HIP events on the ctx stream (costs ~0.1 ms of host
 * time per train step); on = 2: only the dominant kernel (rasterize backward); 0: off. */
int bh_profile_enable(bh_ctx* ctx, int on);
/* Fetch (and clear) accumulated per-stage milliseconds and launch counts; returns
 * the number of stages written (<= cap).  names[i] are static strings. */
int bh_profile_fetch(bh_ctx* ctx, const char** names /*host*/, float* ms /*host*/, uint32_t* calls /*host*/, int cap);

#ifdef __cplusplus
}
#endif
#endif /* BRUSH_HIP_H */
