// context.h — internal host-side state of a bh_ctx and the kernel launcher
// prototypes shared between the translation units of libbrush_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/brush_hip.h"
#include "device_math.h"

namespace bh {

// Scratch arena: named slots that only ever grow, so the steady state of a
// training loop performs no hipMalloc/hipFree (the reference allocates ~20
// buffers per render, render.rs:104-268).
enum Slot : int {
    SLOT_COUNTERS = 0,       // 2 (ping-pong) x [COUNTER_SLOTS][2] u64: partial num_visible, num_intersections
    SLOT_DEPTH_KEYS,         // [N] u32 depth bits / sentinel
    SLOT_ISECT_COUNTS,       // [N] u32
    SLOT_MAX_RADIUS,         // [N] f32
    SLOT_SORT_KEYS_A,        // ping-pong buffers for the radix sort
    SLOT_SORT_KEYS_B,
    SLOT_SORT_VALS_A,
    SLOT_SORT_VALS_B,
    SLOT_SORT_HIST,          // [256 * nblocks] u32
    SLOT_COMM_SCRATCH,       // direct all-reduce: the other ranks' versions of this rank's chunk (comm.hip)
    SLOT_BWD_CKPT,           // backward jobs: the forward blend's pixel-state checkpoints (rasterize.hip)
    SLOT_BWD_TOPLIST,        // ... and the bands' top-class job lists
    SLOT_SORT_PARTS,         // tile sort: per part [9][bins] pair counts (sort.hip tile_parts_*)
    SLOT_SCAN_SUMS,          // block sums for the scan
    SLOT_GLOBAL_FROM_COMPACT,
    SLOT_DEPTHS_SORTED,
    SLOT_CUM_TILES_HIT,
    SLOT_PROJECTED,
    SLOT_TILE_IDS,           // unsorted (tile, gid) pairs
    SLOT_ISECT_GIDS,
    SLOT_TILE_IDS_SORTED,
    SLOT_ISECT_GIDS_SORTED,
    SLOT_TILE_OFFSETS,
    SLOT_OUT_IMG,
    SLOT_VISIBLE,
    SLOT_V_COMBINED,
    SLOT_LOSS_PRED,          // train step: pred in CHW
    SLOT_LOSS_MAP,
    SLOT_LOSS_GRAD,          // dL/dpred CHW
    SLOT_V_OUTPUT,           // [H,W,4]
    SLOT_GRADS,              // exchange buffer: visible | v_transforms | v_sh | v_raw_opac | v_refine
    SLOT_STATS,              // screen radius of the last train-step forward
    SLOT_LOSS_SCALAR,
    SLOT_COL_SCALE,          // per-column lr tables
    SLOT_MISC,
    SLOT_REFINE,             // refine plan: control block + 10 x [N] u32
    SLOT_REFINE_BOUNDS,      // percentile bounds: keys / sorted keys / indices
    SLOT_FOLDED_TRANSFORMS,  // train step with a 3D-filter floor: folded [N,10] / [N]
    SLOT_FOLDED_RAW_OPAC,
    SLOT_PLY_ROWS,           // PLY body being packed / unpacked
    SLOT_EXCH_BLOCKS,        // mask-keyed gradient exchange: per-block counts (+ total) / rank -> splat / compact rows
    SLOT_EXCH_IDX,
    SLOT_EXCH_COMPACT,
    SLOT_PROJECTED_BY_GID,   // [N,9] projected records at their splat id (visible splats only)
    SLOT_SLICE,              // depth-sliced forward: 8 control words (SLICE_CTRL_WORDS) | done bits [ceil(T/32)] | far tile offsets [T,2] | far group totals
    SLOT_SLICE_STATE,        // [H,W,4] raw blend state of the tiles the near slice left unsaturated
    SLOT_SLICE_COUNTS,       // [Nv] live-tile hits per far splat / their inclusive scan
    SLOT_SLICE_CUM,
    SLOT_NEAR_COUNTS,        // [N] tiles hit per splat at or in front of the tile's depth cut (per-tile cut lists)
    SLOT_TILE_ORDER,         // [8][ceil(T/8)] the forward blend's block -> tile map: every XCD band's tiles by descending forecast work
    SLOT_COUNT
};

struct Buffer {
    void* ptr = nullptr;
    size_t cap = 0;
};

constexpr int MAX_PROF = 32;

// K1's block totals land in slot (block % COUNTER_SLOTS); the host adds the slots up after the readback
constexpr uint32_t COUNTER_SLOTS = 128;
// one counter set on the device:
//   [COUNTER_SLOTS][4] u64  visible, intersections, near intersections, near splats (K1's block totals; the last two only with
//                           per-tile depth cuts: the pairs the near pass will list and the splats that own them, see ViewState)
//   [COUNTER_SLOTS][3] u32  slicing feedback, written by the blend kernel of the forward BEFORE the one that accumulates into this
//                           set (rasterize.hip SliceArgs::feedback): max exact-list slots a saturated tile needed | pairs listed
//                           for tiles that never saturated | number of such tiles (empty ones included)
//   [COUNTER_SLOTS][2] u32  max depth key, max ~key over the visible splats: the key range the depth sort splits on.
// The first two parts are read back together (one copy), the third stays on the device.
constexpr uint32_t COUNTER_K1_U64 = 4;                        // u64 words per slot of K1's block totals
constexpr size_t COUNTER_SET_BYTES = COUNTER_SLOTS * 8 * COUNTER_K1_U64 + COUNTER_SLOTS * 12 + COUNTER_SLOTS * 8;
constexpr uint32_t COUNTER_SET_U64 = (uint32_t)(COUNTER_SET_BYTES / 8);
constexpr uint32_t COUNTER_FB_WORD = COUNTER_SLOTS * 2 * COUNTER_K1_U64;   // u32 index of the feedback part inside a set: [COUNTER_SLOTS][3]
constexpr uint32_t COUNTER_MINMAX_WORD = COUNTER_FB_WORD + COUNTER_SLOTS * 3;   // ... of the key-range part
// control words of SLOT_SLICE: [0] near-slice splats  [1] near pairs  [2] tiles the near pass left live  [3] far pairs
// [4] live column bands  [5] live row bands (32 bands per axis; the far pass skips splats whose box misses them)  [6..7] spare
constexpr uint32_t SLICE_CTRL_WORDS = 8;
constexpr uint32_t FAR_GROUP_BLOCKS = 64;   // far slice: count-kernel blocks per group total (<= the projection workgroup size)
constexpr size_t COUNTER_READ_BYTES = COUNTER_SLOTS * 8 * COUNTER_K1_U64 + COUNTER_SLOTS * 12;
// ... or, when the depth sort's first kernel adds the slots up on the device: [COUNTER_K1_U64] u64 totals | [3] u32 feedback
constexpr uint32_t HOST_SUM_WORDS = 2 * COUNTER_K1_U64 + 3 + 1;   // (+ the tag word the host polls)
// pinned host block: [16] u32 scalars (refine control block / bounds picks / exchange rows) | the counter slots + feedback | the
// loss word.  The loss has a word of its own BEHIND everything the refine / bounds / exchange readbacks overwrite.
constexpr size_t HOST_LOSS_WORD = 16 + COUNTER_READ_BYTES / 4;
constexpr size_t HOST_GATE_WORD = HOST_LOSS_WORD + 16;   // depth-sliced forward: tiles the near slice left unsaturated
constexpr size_t HOST_GATE_TAG_WORD = HOST_GATE_WORD + 1;   // ... and the tag the kernel queued BEHIND the near blend stores when it starts (= the blend has finished)
constexpr size_t HOST_COUNTERS_BYTES = 64 + COUNTER_READ_BYTES + 64 + 64;

struct Profiler {
    int level = 0;  // 0 off, 1 every stage, 2 only the dominant kernel (2 events per step)
    const char* names[MAX_PROF];
    float ms[MAX_PROF];
    uint32_t calls[MAX_PROF];
    int count = 0;
    // pending (start, stop) event pairs, resolved lazily at fetch / sync time
    struct Pending { int idx; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    // level 2: the dominant kernel's launch carries its own (start, stop) events (hipExtLaunchKernelGGL): the times
    // come from the dispatch packet itself, without the two barrier packets (~5.6 us of bubble each) that
    // hipEventRecord puts in front of and behind the kernel
    hipEvent_t ext_a = nullptr, ext_b = nullptr;
};

// Per-view state of the per-tile depth cut (BH_FLAG_SLICED_LISTS, automatic mode; api.hip has the whole story): for every tile the
// depth key behind which the tile needed no splat the last time THIS view was rendered (+ a margin), 0xFFFFFFFF = list everything.
// Written in place by the blend kernel of every forward of the view, read by K1 / K5 / the far pass of its next one.
// Bit 0 of an entry is not part of the cut: K1 sets it when it meets a pair BEHIND the cut ("this tile's near list is incomplete
// this frame"), the blend kernel reads it and rewrites the entry with the bit clear.  Cuts are compared without it (zcut_near).
constexpr uint32_t ZCUT_ALL = 0xFFFFFFFEu;
BH_DEV bool zcut_near(uint32_t key, uint32_t cut) { return (key >> 1) <= (cut >> 1); }
struct ViewState {
    uint32_t* zcut = nullptr;       // [tile_bw * tile_bh] device; directly behind it: work[tile_bw * tile_bh], the splats every tile blended at the
                                    // view's last frame (the forward blend's tile order, rasterize.hip); behind that: [VIEW_SPL_WORDS] the depth
                                    // sort's splitter tables of the view
    uint32_t tile_bw = 0, tile_bh = 0;
    bool seeded = false;            // a forward of this view has written the table
    uint32_t exact_frames = 0;      // frames to render with complete lists before the cut is trusted again (the forecast kept failing)
    uint32_t penalty = 0;           // one bit per recent cut frame: its far pass had to run (api.hip view_outcome)
    uint64_t last_used = 0;         // LRU stamp (the ctx's forward counter at the view's last frame)
    uint32_t gap = 0;               // forwards between the view's last two frames (a new view: the number of views the ctx knows) —
                                    // how stale the table will be when it is next read: the margin written into it grows with it
    uint32_t last_pairs = 0;        // num_intersections of the view's last frame
    float last_share = 0.0f;        // share of its pairs the view's last CUT frame listed (0: none yet)
    uint32_t complete_frames = 0;   // frames to render with complete lists because cutting saved (almost) nothing last time; then one probe frame
    bool spl_written[2] = {false, false};   // the view's depth-sort splitter tables ([0] complete lists, [1] cut lists) have been written by a frame
    bool casual = false;            // created by a forward-only frame keyed by its camera (a viewer / eval render): these compete for CASUAL_VIEW_STATES tables only
};
constexpr uint32_t CUT_MIN_PAIRS = 1500000u;   // frames with fewer pairs keep complete lists (nothing to save)
// work classes of the blend backward's longest-first order (rasterize.hip): counters [8 XCD bands][LPT_CLASSES] behind the tile table
constexpr uint32_t LPT_CLASSES = 64;
constexpr uint32_t LPT_HEADER_WORDS = 8 * LPT_CLASSES + 64;   // the counters + 64 control words ([0]: checkpoint slots handed out), cleared with the tile table
// Backward JOBS (round 6, rasterize.hip): the blend backward's unit of work is not a tile but a segment of BWD_SEG entries of a
// tile's blended list.  The forward blend leaves the pixel state (colour so far, signed transmittance) of its 256 pixels at every
// segment boundary — a CHECKPOINT, 4 KB — and files one job per segment; a job starts from its checkpoint instead of replaying
// the tile from its first splat.  The launch then lasts as long as its total work, not as long as its heaviest tile.
constexpr uint32_t BWD_SEG = 128;       // list entries per job (two staging batches)
constexpr uint32_t BWD_MAX_SEGS = 64;   // checkpointed segments per tile; a tile's last job takes whatever lies behind
struct BwdJobs {                        // all zero: one job per tile (far-sliced frames, option bwd_jobs = 0)
    float4* ckpt = nullptr;             // [ckpt_cap][4][64] pixel state at a segment's first entry.  Slot of (tile, segment s >= 1) =
                                        // (first list entry of the tile) / BWD_SEG + tile + s - 1: unique without an atomic or a table, because the
                                        // tiles' lists lie one behind the other (< listed pairs / BWD_SEG + tiles slots; a slot >= ckpt_cap does not exist:
                                        // that tile keeps the rest of its list as one job)
    uint32_t ckpt_cap = 0;
    uint32_t* top_list = nullptr;       // [8][top_cap] the bands' TOP class (every full segment lands there: more entries than tiles)
    uint32_t top_cap = 0;
};
// XCD BANDS.  The dispatcher deals consecutive workgroups to the eight XCDs in turn, so block b of a blend launch runs on XCD b & 7 and
// takes the (b >> 3)-th tile of BAND b & 7.  band_mode 0: a band is a contiguous eighth of the tile index range (whole tile rows: the
// XCD's L2 sees one spatially coherent strip of the frame).  band_mode 1 (round 6): bands are dealt in chunks of BAND_CHUNK horizontally
// adjacent tiles, chunk c to band c & 7 — every XCD gets an even share of any frame (an object in front of an empty background leaves
// the top and bottom strips with nothing to do: two of eight XCDs idle through both blend kernels, and the object's heaviest tiles
// share the L2 and the SIMDs of two others).  Every band has band_slots() slots; a slot may name no tile (>= num_tiles).
constexpr uint32_t BAND_CHUNK = 4;   // (2 / 4 / 8 / 16 measured: profiles/EXPERIMENTS.md)
inline __host__ __device__ uint32_t band_slots(uint32_t num_tiles) {
    const uint32_t p = (num_tiles + 7u) / 8u;
    return ((p + BAND_CHUNK - 1u) / BAND_CHUNK) * BAND_CHUNK;
}
inline __host__ __device__ uint32_t band_tile(uint32_t band, uint32_t i, uint32_t per, uint32_t mode) {
    return mode ? ((i / BAND_CHUNK) * 8u + band) * BAND_CHUNK + (i % BAND_CHUNK) : band * per + i;
}
// SPLIT TILES (round 6, rasterize.hip): the forward blend cannot be cut along a tile's list (a pixel's stop rule needs everything in
// front of it), so a launch lasts as long as its heaviest tile — one wave working through its list at a lone wave's pace.  The few
// tiles whose forecast work (the view's last frame) is several times their band's mean are blended by FOUR waves, one per 8 x 8 pixel
// quadrant (four independent one-wave blocks; the last of them to finish does the tile's bookkeeping).  K1's order blocks decide which:
// the first split_count[band] ranks of a band's descending order.  Layout behind the order table [8][per]:
//   split_count[8] | split_scratch[8][SPLIT_MAX][4] = finished quadrants | max last useful entry | max entry reached | 1 = a quadrant is unsaturated
constexpr uint32_t SPLIT_MAX = 128;         // split tiles per XCD band at most (the grid carries 3 * SPLIT_MAX extra blocks per band)
constexpr uint32_t SPLIT_MIN_WORK = 256;    // a tile below this many blended splats is never split
constexpr uint32_t SPLIT_TAIL_WORDS = 8u + 8u * SPLIT_MAX * 4u;
constexpr uint32_t BWD_CKPT_MAX_SLOTS = 96u * 1024u;   // 384 MB of checkpoints at most (frames with > ~11 M listed pairs split only their first tiles)
constexpr uint32_t DSORT_SPL_STRIDE = 260;            // words of one depth-sort splitter table (depth_sort.hip SPLITTERS)
constexpr uint32_t VIEW_SPL_WORDS = 2 * DSORT_SPL_STRIDE;   // a view keeps two behind its tile table: [0] frames with complete lists, [1] cut lists
constexpr size_t MAX_VIEW_STATES = 4096;
constexpr uint64_t DIRECT_ALLREDUCE_MIN_FLOATS = 1u << 16;   // shorter messages are latency-bound: ncclAllReduce
constexpr uint32_t AUTO_EXACT_FRAMES = 12;   // frames a view renders complete lists after a cut frame that listed > auto_exact_share of its pairs
// Forward-only frames that name no view (a viewer's free camera, an eval render, a pose-optimised camera: a new hash every frame) get
// a table only when the same camera is seen a SECOND time, and at most this many of them are kept: such frames cut nothing, the table
// only orders their blend's tiles.
constexpr size_t CASUAL_VIEW_STATES = 32;
constexpr size_t SEEN_KEYS = 64;   // ring of camera hashes met once
constexpr size_t VIEW_TABLE_BYTES = (size_t)256 << 20;   // ... and at most this much device memory in tables (8 B per tile and view: 64 KB at 1080p, 253 KB at 4K)

// scratch of a depth-sliced forward (rasterize.hip SliceArgs); feedback alone may be set for the exact path (phase 0)
struct RasterSlice {
    uint32_t* done_bits = nullptr;
    uint32_t* unsat_count = nullptr;
    uint32_t* gate_host = nullptr;   // pinned host word: the near pass stores 1 there when it parks a tile (the host's copy of unsat_count != 0)
    float* state = nullptr;
    const uint32_t* offsets_near = nullptr;
    const uint32_t* cum = nullptr;
    uint32_t* feedback = nullptr;
    // per-tile depth cut: the view's table (read: is the tile's near list complete?  written: what the tile needed this time),
    // the depth keys in compact order and their number (to turn "last useful splat + margin" into a key); cut_active: the lists of
    // this frame were built against the table (else it is only written)
    uint32_t* zcut = nullptr;
    const uint32_t* depth_keys_sorted = nullptr;
    uint32_t nv = 0;
    bool cut_active = false;
    uint32_t* live_bands = nullptr;   // the two band words of the slice table (SLICE_CTRL_WORDS): sliced phases only
    uint32_t* work = nullptr;              // [T] the view's per-tile blended counts: written by every phase that finishes a tile
    const uint32_t* order = nullptr;       // [8][ceil(Tw/8)] block -> local tile of this launch (K1 sorted each XCD band by the view's last work), or NULL
    uint32_t order_mode = 1;
    uint32_t* split = nullptr;        // split tiles: split_count[8] | split_scratch (behind `order`; written by K1's order blocks), or NULL
    uint32_t margin_pct = 150;        // depth-order margin behind a tile's last useful splat, in % of its rank
    BwdJobs jobs{};                   // BWD_INFO: file the backward's work as jobs with checkpoints (ckpt != NULL)
};

// The far slice of a depth-sliced forward, ready to be queued: everything launch_* needs (api.hip enqueue_far_slice).
struct FarJob {
    bool pending = false;   // the near slice is queued and its gate word is on its way to the host; the far slice is undecided
    ViewUniforms u{};
    float bg[3] = {0, 0, 0};
    bool bwd_info = false, smooth = false;
    uint32_t nv = 0, ni = 0, budget = 0, num_tiles = 0, tile_bits = 0;
    const float* proj_by_gid = nullptr;
    const uint32_t* gfc = nullptr;
    float* projected = nullptr;
    const uint32_t* cum = nullptr;
    uint32_t* slice_info = nullptr;
    uint32_t* done_bits = nullptr;
    uint32_t* tile_offsets_far = nullptr;
    uint32_t* far_counts = nullptr;
    uint32_t* far_block_totals = nullptr;
    uint32_t* far_group_totals = nullptr;   // zeroed by K1 with the slice table
    uint32_t* tile_ids = nullptr;
    uint32_t* isect_gids = nullptr;
    uint32_t* tile_ids_sorted = nullptr;
    uint32_t* isect_gids_sorted = nullptr;
    float* out_f32 = nullptr;
    uint32_t* out_u8 = nullptr;
    float* visible = nullptr;
    uint32_t* lpt = nullptr;
    float class_width = 8.0f;
    RasterSlice rs{};
    // per-tile cut lists: only the splats with a pair in front of some cut were sorted and listed, so there is no far pass — if a
    // tile is still live behind a cut list the forecast has failed and the whole forward is run again with complete lists
    // (api.hip finish_far_slice).  Its arguments, and the train step's redirections that were in force:
    // how the host learns that the near blend has finished: an event behind it (recorded when the job is created, or late, by
    // finish_far_slice), or — bh_train_step — the tag word its loss kernel stores when it starts (no event: a barrier packet
    // costs ~6 us of bubble in front of the next kernel)
    bool gate_event_recorded = false;
    uint32_t gate_tag = 0;
    bool by_cut = false;
    ViewState* view = nullptr;      // whose prediction failed
    bool view_shared = false;       // ... and whether that is the table of view id 0 (shared by all frames without an id)
    BhCamera cam{};
    uint32_t n = 0, sh_degree = 0, flags = 0, view_id = 0;
    const float* transforms = nullptr;
    const float* sh_coeffs = nullptr;
    const float* raw_opacities = nullptr;
    float* ext_visible = nullptr;
    float* ext_max_radius = nullptr;
    size_t ext_visible_floats = 0;
    float* ext_grad_begin = nullptr;
    size_t ext_grad_floats = 0;
};

// Clears done "on the way" by a forward's kernels, and how the train step wants its gradient span treated.
//   span:  NONE        nobody cleared the span: the backward zero-fills every dense output it writes into
//          ZEROED      K1 of forward `generation` cleared the WHOLE ext span it was given (exchange / multi-GPU step)
//          ROW_MARKS   single-GPU train step: only the refine-weight vector is clear (K1 or a fill did it); K18 marks the rows it
//                      writes in that vector's sign bits and the update kernel reads exactly those rows — the backward must NOT
//                      fill the span and must run K18 in marking mode
//   accum: K5 of forward `generation` cleared v_combined [num_listed_splats, 10]
// Transitions: begin_forward() -> (k1_cleared_span | k5_cleared_accum)* -> stamp(generation) -> [mark_rows()] -> take_*() in the
// backward of THAT generation (anything else finds NONE / false).  A backward always leaves the record empty: the accumulator
// and the span are dirty afterwards.
struct GradClears {
    enum Span : uint8_t { NONE = 0, ZEROED, ROW_MARKS };
    Span span = NONE;
    bool accum = false;
    uint64_t generation = 0;   // 0: not stamped yet (the forward is still being queued)
    void begin_forward() { span = NONE; accum = false; generation = 0; }
    void k1_cleared_span() { span = ZEROED; }
    void k5_cleared_accum(bool yes) { accum = yes; }
    void stamp(uint64_t gen) { generation = gen; }
    // train step, single GPU: the refine-weight vector is clear (by K1: span == ZEROED for that sub-span, or by the caller's fill)
    void mark_rows() { span = ROW_MARKS; }
    bool valid_for(uint64_t gen) const { return generation != 0 && generation == gen; }
    Span take_span(uint64_t gen) { const Span s = valid_for(gen) ? span : NONE; span = NONE; return s; }
    bool take_accum(uint64_t gen) { const bool a = valid_for(gen) && accum; accum = false; return a; }
};

// What a backward needs of the forward it belongs to (RenderBackwards' saved state, bwd/burn_glue.rs:336-371): the host side of it.
struct ForwardState {
    BhRenderOut out{};
    ViewUniforms uniforms{};
    uint32_t n = 0, sh_degree = 0, flags = 0;
    float bg[3] = {0, 0, 0};
    uint32_t* lpt = nullptr;   // longest-first tile order of the forward (rasterize.hip), or NULL
    BwdJobs jobs{};            // ... filed as jobs with checkpoints (ckpt != NULL), or as whole tiles
};
// A forward whose arena blocks were detached from the ctx (bh_render_retain): it stays replayable while later forwards run.
constexpr int RETAIN_SLOTS = 14;
struct Retained {
    ForwardState fs;
    Buffer blocks[RETAIN_SLOTS];
};

}  // namespace bh

// The blend backward accumulates RAW per-splat sums into v_combined and the projection backward maps them to the reference's
// RasterizeGrads row (and stores it back) — rasterize.hip / project.hip.

struct bh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string last_error;
    bh::Buffer slots[bh::SLOT_COUNT];
    uint32_t* host_counters = nullptr;  // pinned, HOST_COUNTERS_BYTES
    hipEvent_t readback_ev = nullptr;   // marks the count readback of the forward (the depth sort is queued behind it)
    hipEvent_t gate_ev = nullptr;       // depth-sliced forward: the near slice's blend + the copy of its gate word have run
    bool counters_ready = false;        // the counter pair the next forward accumulates into is known to be zero
    uint32_t counter_phase = 0;         // which half of SLOT_COUNTERS that is
    // state of the last forward (what RenderBackwards saves, bwd/burn_glue.rs:336-371)
    bool have_forward = false;
    BhCamera cam{};
    bh::ViewUniforms uniforms{};
    uint32_t n = 0, sh_degree = 0, flags = 0;
    float bg[3] = {0, 0, 0};
    BhRenderOut last{};
    uint64_t generation = 0;          // stamped into every BhRenderOut a forward of this ctx returns (bh_render_backward_saved checks it)
    std::vector<bh::Retained> retained;   // forwards detached by bh_render_retain, until bh_render_release
    std::vector<bh::Buffer> pool;     // blocks given back by bh_render_release: ensure() takes from here before it asks hipMalloc
    float* ext_visible = nullptr;     // train step: the forward writes visible / max_radius here (stats buffer)
    float* ext_max_radius = nullptr;
    size_t ext_visible_floats = 0;    // floats to clear at ext_visible (its section of the exchange buffer incl. padding)
    float* ext_grad_begin = nullptr;  // train step: v_transforms .. end of the exchange buffer is one span to zero-fill
    size_t ext_grad_floats = 0;
    // What the forward's kernels cleared on their way, so that the backward can skip its fills (K1: the train step's gradient span;
    // K5: the backward's accumulator v_combined).  ONE record with explicit transitions (bh::GradClears below) instead of loose
    // flags: every fact is tied to the forward (generation) whose kernels established it and is consumed exactly once.
    bh::GradClears clears;
    float* pending_loss_dst = nullptr; // where bh_sync delivers the last step's loss
    void* comm = nullptr;             // RCCL communicator (comm.hip), or NULL
    // the library communicator's side stream: the mask-keyed exchange sums the visible flags and lists their union there,
    // beside the backward on the ctx stream (api.hip); comm_ev marks "the forward is done" for it
    hipStream_t comm_stream = nullptr;
    hipEvent_t comm_ev = nullptr;
    int comm_rank = 0, comm_world = 1;
    uint32_t* lpt = nullptr;          // longest-first tile order of the last BWD_INFO forward (rasterize.hip), or NULL
    // depth-sliced lists (BH_FLAG_SLICED_LISTS): share of the pair list the near slice takes.  <= 0: chosen per frame from the
    // previous frame's feedback (bh_set_list_slicing)
    float slice_fraction = 0.0f;
    bool had_forward = false;         // a forward ran on this ctx before (its feedback words are meaningful)
    bool last_one_slice = true;       // ... and built one list per tile (exact path, or a sliced request that chose one slice)
    uint32_t prev_intersections = 0;  // ... and listed this many pairs
    float need_hint = 0.0f;           // fading maximum of the share of the pair list recent frames' slowest saturating tile needed
    float last_slice_share = 1.0f;    // what the last sliced forward used (1 = one slice = the exact lists); diagnostics
    // The far slice costs ~12 launches even when every one of them is a no-op (~4.5 us each on this chip), so whether to queue
    // it is decided on the HOST where possible: far_direct = the previous sliced frame needed it -> queue it right away
    // (device-gated, no host wait; the gate word is copied out to keep learning); otherwise the host reads the gate word —
    // bh_render_forward waits for it, bh_train_step queues the loss kernels first and waits behind them (far_job.pending).
    bool far_direct = false;
    bool bwd_skip_refine = false;     // set by bh_train_step around its backward: step >= BhTrainConfig.growth_stop_iter, nobody reads the refine weight
    bool defer_far = false;           // set by bh_train_step around its forward: return with far_job.pending instead of waiting
    uint32_t readback_tag = 0;        // tag of the last count readback the host polled for (depth_sort.hip counter_sums_to_host)
    uint32_t gate_tag = 0;            // tag of the last deferred far-slice decision
    bool gate_signal_queued = false;  // a kernel that stores far_job.gate_tag when it starts is queued behind the near blend
    bool gate_learn = false;          // a gate word copied out by a far_direct frame has not been looked at yet
    uint32_t far_launches = 0;        // diagnostics: sliced forwards that queued a far slice / had to be run again with complete lists
    uint32_t last_listed_splats = 0;  // compact entries of the last forward (== num_visible unless per-tile cuts listed a subset)
    // per-tile depth cuts (automatic slicing): one table per view id (bh_set_view_id / BhTrainBatch.view_id; 0 = the ctx's own slot)
    // keyed by the caller's view id, or — id 0 — by a hash of the camera (api.hip view_key): an unmodified SplatTrainer::step
    // (train.rs:176: a SceneBatch carries no view index, brush-dataset/src/scene.rs:138-147) gets the same tables
    std::unordered_map<uint64_t, bh::ViewState> views;
    uint64_t seen_keys[bh::SEEN_KEYS] = {};   // camera hashes of forward-only frames that found no table (api.hip casual_view)
    uint32_t seen_pos = 0;
    uint32_t casual_views = 0;                // tables whose ViewState::casual is set
    uint32_t view_id = 0;
    uint64_t view_clock = 0;
    bh::ViewState* gate_view = nullptr;   // the view whose far pass was queued unasked (gate_learn): penalised if it was needed
    // The margin behind a tile's last useful splat adapts (api.hip cut_margin_pct): x (gap / 2)^(1/3) for a view that comes back
    // after `gap` frames, x margin_scale — multiplied by 1.5 when a forecast fails, by 0.998 when one holds (about one second
    // attempt in 200 cut frames at equilibrium), within [0.5, 16].  A scene that still moves fast (early training, many views between
    // two visits) gets deep margins, a settled one tight ones.
    float margin_scale = 1.0f;
    float ctrl_up = 1.5f, ctrl_down = 0.998f, ctrl_floor = 0.5f, ctrl_gap_exp = 1.0f / 3.0f;   // BH_CUT_CTRL="up:down:floor:gap_exp" (A/B)
    uint32_t cut_min_pairs = bh::CUT_MIN_PAIRS;   // bh_set_list_cut_threshold / BH_CUT_MIN_PAIRS
    bool knob_spec_k5 = true;             // option spec_k5: K5 queued in front of the count readback (api.hip)
    bool knob_event_waits = false;        // BH_EVENT_WAITS (A/B): the host's two mid-step waits use events behind the kernels, as before round 5, instead of polled tag words
    bool knob_readback_copy = false;      // BH_READBACK_COPY (A/B): counts and gate word reach the host through copy launches as before round 4
    bool knob_no_view_hash = false;       // BH_NO_VIEW_HASH (A/B): frames without a view id share ONE table (rounds 4's behaviour) instead of being keyed by their camera
    bool knob_fixed_margin = false;       // BH_CUT_MARGIN_FIXED (A/B): the margin is BH_CUT_MARGIN_PCT for every frame (rounds 4's behaviour), not adaptive
    bool knob_cut_sort_all = false;       // BH_CUT_SORT_ALL (A/B): with per-tile cuts, still sort every visible splat
    uint32_t knob_band_mode = 1;          // XCD bands of the blend kernels: 0 contiguous eighths of the tile range, 1 dealt in chunks of 8 tiles
    uint32_t knob_k16_waves = 0;          // forward blend: resident waves per SIMD (0 / 8: all eight; fewer: later tiles are dispatched as earlier ones finish)
    uint32_t knob_k16_split = 250;        // forward blend: split tiles at >= max(SPLIT_MIN_WORK, this / 100 x the band's mean forecast work); 0: never
    uint32_t knob_k16_split_min = bh::SPLIT_MIN_WORK;
    uint32_t knob_k16_split_of_max = 45;  // ... and this many percent of the band's heaviest tile
    const uint32_t* last_split = nullptr; // (test-hooks build: bh_debug_split_counts) the last forward's split counts, or NULL
    uint32_t knob_k16_order = 1;          // BH_K16_ORDER: 0 index order, 1 by the view's last per-tile work (descending), 2 dealt (A/B)
    uint32_t knob_cut_margin_pct = 150;   // BH_CUT_MARGIN_PCT: margin behind a tile's last useful splat, % of its depth rank (A/B)
    bh::FarJob far_job;
    uint32_t refine_n = 0, refine_new_n = 0;  // a bh_refine_plan awaiting its bh_refine_apply
    bool dsort_lds_raised = false;    // likewise dsort_bucket_kernel (depth_sort.hip)
    uint32_t* dsort_spl = nullptr;    // [DSORT_SPL_STRIDE] device: the depth sort's splitter table (depth_sort.hip SPLITTERS) of frames without a view
    bool dsort_spl_written = false;   // a frame has written it (until then a frame sorts a sample first)
    bool knob_dsort_splitters = true; // option dsort_splitters: the split digit from the previous frame's quantiles (0: always the linear split)
    bool adam_lds_raised = false;     // adam_rowreduced_kernel's > 64 KB dynamic-LDS opt-in was made on this ctx's device
    // developer knobs (A/B measurements): bh_set_option
    bool knob_no_lpt = false;         // option no_lpt: backward tiles in index order
    bool knob_bwd_jobs = true;        // option bwd_jobs: the blend backward works on checkpointed segments of the tiles' lists (rasterize.hip)
    bh::BwdJobs jobs{};                   // of the last BWD_INFO forward (with ctx->lpt)
    bool knob_lpt_linear = false;     // option lpt_classes=linear: the work classes of rounds 2-5 (rasterize.hip)
    bool knob_force_exchange = false;       // BH_FORCE_PG: run the gradient-exchange path with a one-rank communicator too (overhead measurement)
    bool knob_break_allreduce = false;      // BH_BREAK_ALLREDUCE: corrupt the library's all-reduce (the bench self-check must notice)
    bool knob_generic_depth_sort = false;   // BH_GENERIC_DEPTH_SORT: the forward's depth order by the generic 32-bit radix sort + scan
    bool knob_zero_grads = false;     // BH_TRAIN_ZERO_GRADS: the single-GPU train step zero-fills its gradient span like the exchange path (A/B)
    uint32_t knob_fail_loss_at = 0;   // BH_TEST_FAIL_LOSS_AT: the k-th bh_train_step on this ctx returns BH_ERR_OOM between its forward and its loss (test hook)
    uint32_t train_steps_seen = 0;
    uint32_t knob_loss_bands = 1;     // BH_LOSS_BANDS=0 (A/B): the fused loss's blocks take their tiles row-major instead of by XCD column bands
    uint32_t knob_tile_sort = 0;      // option tile_sort: 0 auto (bucket sort unless the view's pairs sit in few tiles), 1 bucket, 2 two LSD passes + the offsets kernel
    float auto_exact_share = 0.9f;    // option auto_exact_share: a view whose last cut frame listed more than this share renders complete lists
    bool knob_direct_allreduce = false;   // option grad_allreduce=direct: reduce-scatter + all-gather over grouped send / recv (comm.hip)
    uint32_t knob_k5_exact_spw = 32;  // BH_K5_EXACT_SPW = 16 | 32 | 64 (A/B): splats per wave of K5 for complete lists
    bool knob_no_dormant = false;     // BH_UPDATE_NO_DORMANT (A/B, tests): the update kernel fetches and updates dormant splats like everyone else
    bool knob_update_early = false;   // BH_UPDATE_EARLY (A/B): the update kernel's blocks of SH degree >= 1 issue all their loads up front
    uint32_t knob_update_rows = 0;    // BH_UPDATE_ROWS: 64 | 128 | 256 splats per block of the update kernel
    uint32_t knob_sort_kpt = 0;       // BH_SORT_KPT: 4 | 8 | 16 keys per thread of the radix sort
    // the update kernel's dormant marks (sign of m2_sh, optim.hip) are trusted only on the state this ctx updated last step
    const void* marks_m2_sh = nullptr; const void* marks_m1_t = nullptr; uint32_t marks_n = 0, marks_step = 0;
    bh::Profiler prof;
};

namespace bh {

int set_error(bh_ctx* ctx, int code, const std::string& msg);
int enqueue_far_slice(bh_ctx* ctx, const FarJob& j);
int finish_far_slice(bh_ctx* ctx, bool* launched);
// after ANY host wait on the ctx stream: hand the last train step's loss (pinned staging word) to its BhTrainStats
inline void deliver_pending_loss(bh_ctx* ctx) {
    if (ctx->pending_loss_dst) {
        *ctx->pending_loss_dst = reinterpret_cast<const volatile float*>(ctx->host_counters)[HOST_LOSS_WORD];
        ctx->pending_loss_dst = nullptr;
    }
}
int check_hip(bh_ctx* ctx, hipError_t e, const char* what);
// Grow-only allocation of a scratch slot; returns nullptr (and sets the error) on failure.
void* ensure(bh_ctx* ctx, Slot s, size_t bytes);

struct ProfScope {
    bh_ctx* ctx;
    int idx = -1;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(bh_ctx* c, const char* name, bool dominant = false);
    ~ProfScope();
};

#define BH_HIP(ctx, expr)                                            \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) return bh::check_hip(ctx, _e, #expr);  \
    } while (0)
#define BH_TRY(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc != 0) return _rc; \
    } while (0)
#define BH_LAUNCH_CHECK(ctx, what)                                      \
    do {                                                                \
        hipError_t _e = hipGetLastError();                              \
        if (_e != hipSuccess) return bh::check_hip(ctx, _e, what);      \
    } while (0)

ViewUniforms make_uniforms(const BhCamera& c);

// ---- launchers (each in its own TU) --------------------------------------------
// project.hip

// Buffers K1 clears on the way (every splat thread stores a few zeros: three fill launches fewer per forward).
struct ForwardPrep {
    unsigned long long* next_counters = nullptr;  // the idle half of the counter ping-pong, cleared for the next forward
    uint32_t* visible = nullptr;                  // visible flags (K16 sets them), or NULL
    uint32_t visible_words = 0;
    uint32_t* tile_table = nullptr;               // tile_offsets + the work-class counters behind it
    uint32_t tile_words = 0;
    float4* span = nullptr;                       // train step: the gradient span of the exchange buffer (grid-stride float4 clear)
    uint32_t span_f4 = 0;
    uint32_t* slice_table = nullptr;              // depth-sliced forward: control words + done bits + far tile offsets
    uint32_t slice_words = 0;
    // the forward blend's tile order: blocks 0..7 sort the tiles of XCD band b by the view's last per-tile work (descending)
    const uint32_t* order_work = nullptr;         // [T] (global tile ids), or NULL
    uint32_t* order_out = nullptr;                // [8][ceil(order_tiles/8)] local tile ids, 0xFFFFFFFF behind a short band
    uint32_t order_tiles = 0, order_tile_begin = 0;
    uint32_t order_mode = 1;                      // 1: descending work   2: dealt (consecutive blocks take every 8th rank)
    uint32_t band_mode = 0;                       // context.h XCD BANDS
    uint32_t* split_out = nullptr;                // split_count[8] | split_scratch (SPLIT_TAIL_WORDS), or NULL: no tile is split
    float split_factor = 2.5f;                    // a tile is split when its forecast work >= max(split_min, split_factor * the band's mean)
    uint32_t split_min = SPLIT_MIN_WORK;
    float split_of_max = 0.45f;                   // ... and >= this share of the band's heaviest tile
    bool list_all_visible = false;                // A/B knob BH_CUT_SORT_ALL: per-tile cuts sort and number EVERY visible splat
};
int launch_project_forward(bh_ctx* ctx, const ViewUniforms& u, uint32_t n, bool mip, uint32_t sh_degree, const float* transforms,
                           const float* sh, const float* raw_opac, uint32_t* depth_keys, uint32_t* isect_counts, float* max_radius,
                           float* projected_by_gid, uint32_t* counters, const ForwardPrep& prep, uint32_t* zcut = nullptr,
                           uint32_t* near_counts = nullptr);
int launch_project_visible(bh_ctx* ctx, uint32_t nv, const float* projected_by_gid, const uint32_t* gid, float* projected);
// budget: only splats whose slot range ends at or below it are emitted (the near slice of a depth-sliced forward; 0xFFFFFFFF =
// all); slice_info (device, 2 words) then receives the slice's splat count and pair count.
int launch_map_gaussians(bh_ctx* ctx, uint32_t nv, const ViewUniforms& u, const float* projected_by_gid, const uint32_t* gid,
                         float* projected, const uint32_t* cum_tiles_hit, uint32_t* tile_ids, uint32_t* isect_gids,
                         float4* zero_span = nullptr, uint32_t zero_f4 = 0, uint32_t budget = 0xFFFFFFFFu, uint32_t* slice_info = nullptr,
                         const uint32_t* zcut = nullptr, const uint32_t* depth_keys_sorted = nullptr, const uint32_t* nv_dev = nullptr, uint32_t pair_cap = 0xFFFFFFFFu);
int launch_map_gaussians_far(bh_ctx* ctx, uint32_t nv, const ViewUniforms& u, const float* projected_by_gid, const uint32_t* gid,
                             float* projected, const uint32_t* cum_tiles_hit, uint32_t budget, const uint32_t* done_bits, const uint32_t* gate,
                             uint32_t* counts, uint32_t* block_totals, uint32_t* group_totals, uint32_t* slice_info, uint32_t* tile_ids, uint32_t* isect_gids);
int launch_project_backward(bh_ctx* ctx, const ViewUniforms& u, uint32_t nv, bool mip, uint32_t sh_degree,
                            const float* transforms, const float* sh, const float* raw_opac, const uint32_t* gid,
                            float* v_combined, float* v_transforms, float* v_sh, float* v_raw_opac,
                            float* v_refine, bool mark_written = false, const float* projected = nullptr);
// sort.hip
int radix_argsort(bh_ctx* ctx, const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t bits,
                  uint32_t* out_keys, uint32_t* out_vals);
// the same with the number of pairs in device memory (host: only the bound n_max): n = min(*n_dev, n_max), 0 if *gate == 0
// (gate may be NULL); the result is written *out_base elements into out_keys / out_vals (out_base may be NULL).  Not in place.
int radix_argsort_dev(bh_ctx* ctx, const uint32_t* keys, const uint32_t* vals, uint32_t n_max, const uint32_t* n_dev, const uint32_t* gate,
                      const uint32_t* out_base, uint32_t bits, uint32_t* out_keys, uint32_t* out_vals, uint32_t alloc_n = 0);
// sort.hip — the forward's tile sort AND its offsets table in four launches (high digit first, one block per bucket finishes);
// bits in 9..16 and n <= 16 M (tile_sort_supported), the table zero on entry
bool tile_sort_supported(uint32_t bits, uint32_t n);
int tile_sort_offsets(bh_ctx* ctx, const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t bits, uint32_t num_tiles,
                      uint32_t* out_keys, uint32_t* out_vals, uint32_t* tile_offsets, uint32_t alloc_n = 0);
// depth_sort.hip — the forward's depth ordering: stable argsort of the depth keys + inclusive scan of the tile counts in that
// order, four launches.  minmax: the second part of a counter set (K1).  cum == NULL: no scan.  rb_*: the first launch also adds
// up counter set rb_set into the pinned host words rb_host (HOST_SUM_WORDS) and rb_done is recorded behind it.
bool depth_sort_supported(uint32_t n);
int depth_sort_scan(bh_ctx* ctx, const uint32_t* keys, const uint32_t* minmax, const uint32_t* counts, uint32_t n, uint32_t* out_keys,
                    uint32_t* out_vals, uint32_t* cum, const uint32_t* rb_set = nullptr, uint32_t* rb_host = nullptr, hipEvent_t rb_done = nullptr,
                    uint32_t rb_tag = 0, uint32_t* rb_dev = nullptr, uint32_t* spl = nullptr, bool* spl_written = nullptr);
// scan.hip — inclusive scan; if `gather` != nullptr the input is in[gather[i]]. exclusive: out[i] = sum_{j<i}.
// gate != NULL: a device word; 0 there turns the launches into no-ops (the depth-sliced forward's second slice)
int prefix_sum(bh_ctx* ctx, const uint32_t* in, const uint32_t* gather, uint32_t n, uint32_t* out, bool exclusive, const uint32_t* gate = nullptr);
// rasterize.hip
int launch_tile_offsets(bh_ctx* ctx, const uint32_t* tile_ids_sorted, uint32_t num_isect, uint32_t num_tiles,
                        uint32_t* tile_offsets, bool pre_zeroed = false);
// the list's length in device memory (min(*n_dev, n_max), 0 if *gate == 0); it starts *base entries into tile_ids_sorted and the
// offsets written are absolute (gate / base may be NULL).  The table must already be zero.
int launch_tile_offsets_dev(bh_ctx* ctx, const uint32_t* tile_ids_sorted, uint32_t n_max, const uint32_t* n_dev, const uint32_t* gate,
                            const uint32_t* base, uint32_t num_tiles, uint32_t* tile_offsets);

// lpt: the longest-first tile order scratch (8*16 counters directly behind tile_offsets, then the class lists); NULL = index order
int launch_rasterize(bh_ctx* ctx, const ViewUniforms& u, const float bg[3], bool bwd_info, bool smooth,
                     const uint32_t* isect_gids, uint32_t* tile_offsets, const float* projected,
                     const uint32_t* global_from_compact, float* out_img, uint32_t* out_packed, float* visible,
                     uint32_t* lpt, float class_width, int phase = 0, const RasterSlice* slice = nullptr);
int launch_rasterize_backward(bh_ctx* ctx, const ViewUniforms& u, const float bg[3], bool smooth,
                              const uint32_t* isect_gids, const uint32_t* tile_offsets, const float* projected,
                              const float* out_img, const float* v_output, float* v_combined, const uint32_t* lpt,
                              const uint32_t* tile_offsets_far = nullptr, bool want_refine = true, const BwdJobs* jobs = nullptr);
// loss.hip
int launch_image_loss_forward(bh_ctx* ctx, const float* pred, const uint32_t* gt, uint32_t channels, uint32_t h,
                              uint32_t w, const BhLossConfig& cfg, float* loss_map);
int launch_image_loss_backward(bh_ctx* ctx, const float* pred, const uint32_t* gt, const float* dl_dmap,
                               float dl_const, uint32_t channels, uint32_t h, uint32_t w, const BhLossConfig& cfg,
                               float* dl_dpred);
// layout helpers used by the train step
int launch_hwc4_to_chw(bh_ctx* ctx, const float* img_hwc4, uint32_t channels, uint32_t h, uint32_t w, float* chw);
int launch_chw_to_hwc4(bh_ctx* ctx, const float* chw, uint32_t channels, uint32_t h, uint32_t w, float* hwc4);
int launch_sum(bh_ctx* ctx, const float* x, uint64_t n, float scale, float* out_scalar, bool accumulate);
// optim.hip
int launch_adam(bh_ctx* ctx, float* param, const float* grad, float* m1, float* m2, uint64_t rows, uint32_t row_len,
                const float* col_scale, float lr, uint32_t t, bool reduce_m2, float beta1, float beta2, float eps);
// statistics + the three Adam updates of a train step in one launch (tab_t: per-column lr of `transforms`)
// noise != NULL: the visibility-gated mean noise of train.rs:389-416, drawn on the device (device_rng.h), rides on the same launch
struct NoiseArgs { uint64_t seed; uint32_t step; float scale, clamp_abs; };
int launch_train_update(bh_ctx* ctx, const BhTrainState* st, const float* g_t, const float* g_sh, const float* g_o,
                        const float* refine_weight, const float* visible, const float* screen_radius, float gscale,
                        bool vis_clamp, const float* tab_t, float lr_sh, float sh_rest_scale, float lr_opac, uint32_t t,
                        float beta1, float beta2, float eps, const NoiseArgs* noise = nullptr, bool masked_rows = false);
int launch_gather_stats(bh_ctx* ctx, float* refine_weight_norm, float* vis_weight, float* max_screen_size,
                        const float* refine_weight, const float* visible, const float* screen_radius, uint64_t n);
// samples == NULL: drawn on the device from (seed, step, splat)
int launch_mean_noise(bh_ctx* ctx, float* transforms, const float* raw_opac, const float* visible,
                      const float* samples, uint64_t n, float noise_scale, float clamp_abs, uint64_t seed = 0, uint32_t step = 0);
int launch_normal_samples(bh_ctx* ctx, float* out, uint64_t n, uint64_t seed, uint32_t step);

// filter3d.hip — Mip-Splatting 3D filter (scale floor): fold, its VJP, the floor itself
int launch_fold_min_scale(bh_ctx* ctx, const float* transforms, const float* raw_opac, const float* min_scale, uint32_t n,
                          float* out_transforms, float* out_raw_opac);
int launch_fold_min_scale_backward(bh_ctx* ctx, const float* transforms, const float* raw_opac, const float* min_scale, uint32_t n,
                                   float* v_transforms, float* v_raw_opac);
int launch_compute_min_scale(bh_ctx* ctx, const float* transforms, uint32_t n, const float* view_cams, uint32_t k, float factor, float* out);

// comm.hip — in-place all-reduce of `count` floats over the ctx's RCCL communicator, on the ctx stream
int comm_allreduce(bh_ctx* ctx, float* buf, uint64_t count, bool max_op);   // on ctx->stream
int comm_exchange_strip_halos(bh_ctx* ctx, float* img, uint32_t h, uint32_t w, uint32_t row_begin_px, uint32_t row_end_px);

// exchange.hip — compaction kernels of the mask-keyed gradient exchange
int launch_union_index(bh_ctx* ctx, const float* visible_sum, uint32_t n, uint32_t* block_scratch /*[n/4096+2]*/, uint32_t* count_dev,
                       uint32_t* idx);
int launch_exchange_rows(bh_ctx* ctx, bool gather, const uint32_t* idx, uint32_t count, uint32_t c3, float* g_tr, float* g_sh, float* g_op,
                         float* g_ref /*NULL = no refine-weight column*/, float* compact);

}  // namespace bh
