// api.hip — the extern "C" surface of libbrush_hip.so (include/brush_hip.h):
// context / arena / error plumbing and the host-side orchestration of the
// forward pipeline (render.rs:37-314), the backward (bwd/render_bwd.rs:21-171)
// and SplatTrainer::step (brush-train/src/train.rs:176-429).
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <new>

#include "context.h"
#include "device_rng.h"

namespace bh {

int launch_image_loss_forward_strided(bh_ctx* ctx, const float* pred, uint32_t pix_stride, uint32_t ch_stride, const uint32_t* gt,
                                      uint32_t channels, uint32_t h, uint32_t w, const BhLossConfig& cfg, float* loss_map);
int launch_image_loss_backward_strided(bh_ctx* ctx, const float* pred, uint32_t pix_stride, uint32_t ch_stride, const uint32_t* gt,
                                       const float* dl_dmap, float dl_rgb, float dl_alpha, uint32_t channels, uint32_t h, uint32_t w,
                                       const BhLossConfig& cfg, float* dl_dpred);

int launch_image_loss_fused(bh_ctx* ctx, const float* img_hwc4, const uint32_t* gt, uint32_t h, uint32_t w, const BhLossConfig& cfg,
                            bool alpha_match, float dl_rgb, float dl_alpha, float* loss_out, float* v_output, float* loss_host = nullptr,
                            uint32_t* started_host = nullptr, uint32_t started_tag = 0);
int launch_image_loss_fused_window(bh_ctx* ctx, const float* img_hwc4, const uint32_t* gt, uint32_t h, uint32_t w, const BhLossConfig& cfg,
                                   bool alpha_match, float dl_rgb, float dl_alpha, uint32_t tile_y0, uint32_t tile_y1, float* loss_out,
                                   float* v_output, float* loss_host = nullptr, uint32_t* started_host = nullptr, uint32_t started_tag = 0);

// grid-stride clear of two float4 spans in one launch
__global__ __launch_bounds__(256) void zero_two_kernel(float4* __restrict__ a, size_t na, float4* __restrict__ b, size_t nb) {
    const size_t stride = (size_t)gridDim.x * 256;
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += stride) {
        if (i < na) a[i] = z;
        else b[i - na] = z;
    }
}

int set_error(bh_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    return code;
}

int check_hip(bh_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return set_error(ctx, e == hipErrorOutOfMemory ? BH_ERR_OOM : BH_ERR_HIP, m);
}

// Host wait for a tag word a kernel stores into pinned host memory (instead of an event behind the kernel: the barrier packet an
// event costs is ~6 us of bubble in front of the NEXT kernel).  Spins; every 16 k polls it asks the stream whether it is still
// alive, so a device fault ends in an error instead of a hang.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}
static int wait_host_tag(bh_ctx* ctx, const volatile uint32_t* word, uint32_t want, const char* what) {
    // Spins for at most ~2 ms (a healthy step delivers its tag within tens of microseconds), then stops burning the core: an event
    // behind everything queued so far and a blocking wait on it — the tag's kernel is in front of that event, so afterwards the tag
    // is there, or its kernel was never launched (an error, not a hang).
    for (uint64_t it = 1; it < (1ull << 21); ++it) {
        if (*word == want) return 0;
        if ((it & 0x3FFFull) == 0ull) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) break;   // everything queued has run: the tag must be there (checked below)
            if (q != hipErrorNotReady) return check_hip(ctx, q, what);
        }
        cpu_relax();
    }
    if (*word == want) return 0;
    BH_HIP(ctx, hipEventRecord(ctx->readback_ev, ctx->stream));
    BH_HIP(ctx, hipEventSynchronize(ctx->readback_ev));
    if (*word == want) return 0;
    return set_error(ctx, BH_ERR_STATE, std::string(what) + ": the kernel that signals the host never stored its tag");
}

void* ensure(bh_ctx* ctx, Slot s, size_t bytes) {
    Buffer& b = ctx->slots[s];
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return b.ptr;
    // growing: wait for queued work that may still read the old block
    if (b.ptr) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(b.ptr);
        b.ptr = nullptr;
        b.cap = 0;
    }
    // a block handed back by bh_render_release (the smallest one that fits, at most 2x too large)
    {
        int best = -1;
        for (size_t i = 0; i < ctx->pool.size(); ++i)
            if (ctx->pool[i].cap >= bytes && ctx->pool[i].cap <= 2 * bytes + 4096 && (best < 0 || ctx->pool[i].cap < ctx->pool[(size_t)best].cap)) best = (int)i;
        if (best >= 0) {
            b = ctx->pool[(size_t)best];
            ctx->pool.erase(ctx->pool.begin() + best);
            return b.ptr;
        }
    }
    size_t cap = bytes + bytes / 4;  // head-room so slowly growing scenes do not realloc every step
    cap = (cap + 255) & ~(size_t)255;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        cap = (bytes + 255) & ~(size_t)255;
        e = hipMalloc(&p, cap);
    }
    if (e != hipSuccess) {
        char msg[160];
        snprintf(msg, sizeof msg, "hipMalloc(%zu bytes) for scratch slot %d failed: %s", cap, (int)s, hipGetErrorString(e));
        (void)hipGetLastError();
        set_error(ctx, BH_ERR_OOM, msg);
        return nullptr;
    }
    b.ptr = p;
    b.cap = cap;
    return p;
}

ViewUniforms make_uniforms(const BhCamera& c) {
    ViewUniforms u;
    for (int i = 0; i < 12; ++i) u.vm[i] = c.vm[i];
    u.fx = c.fx; u.fy = c.fy; u.cx = c.cx; u.cy = c.cy;
    u.lim_pos_x = c.lim_pos_x; u.lim_pos_y = c.lim_pos_y; u.lim_neg_x = c.lim_neg_x; u.lim_neg_y = c.lim_neg_y;
    u.cam_x = c.cam_pos[0]; u.cam_y = c.cam_pos[1]; u.cam_z = c.cam_pos[2];
    u.img_w = c.img_w; u.img_h = c.img_h;
    u.tile_bw = (c.img_w + TILE_WIDTH - 1) / TILE_WIDTH;  // render.rs:30-35
    u.tile_bh = (c.img_h + TILE_WIDTH - 1) / TILE_WIDTH;
    const bool whole = c.tile_row_begin == 0 && c.tile_row_end == 0;
    u.tile_y0 = whole ? 0u : c.tile_row_begin;
    u.tile_y1 = whole ? u.tile_bh : c.tile_row_end;
    u.model = c.model;
    for (int i = 0; i < 8; ++i) u.dist[i] = c.dist[i];
    u.half_fov = c.half_max_render_fov;
    return u;
}

// ---- profiling -----------------------------------------------------------------
static int prof_index(Profiler& p, const char* name) {
    for (int i = 0; i < p.count; ++i)
        if (p.names[i] == name || std::strcmp(p.names[i], name) == 0) return i;
    if (p.count >= MAX_PROF) return -1;
    p.names[p.count] = name;
    p.ms[p.count] = 0.0f;
    p.calls[p.count] = 0;
    return p.count++;
}
static hipEvent_t prof_event(Profiler& p) {
    if (!p.pool.empty()) {
        hipEvent_t e = p.pool.back();
        p.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
static void prof_resolve(bh_ctx* ctx) {
    Profiler& p = ctx->prof;
    for (auto& pe : p.pending) {
        (void)hipEventSynchronize(pe.b);
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess && pe.idx >= 0) {
            p.ms[pe.idx] += ms;
            p.calls[pe.idx] += 1;
        }
        p.pool.push_back(pe.a);
        p.pool.push_back(pe.b);
    }
    p.pending.clear();
}

ProfScope::ProfScope(bh_ctx* c, const char* name, bool dominant) : ctx(c) {
    if (c->prof.level == 0 || (c->prof.level == 2 && !dominant)) return;
    idx = prof_index(c->prof, name);
    a = prof_event(c->prof);
    b = prof_event(c->prof);
    if (c->prof.level == 2) {   // handed to the kernel launch inside the scope (Profiler::ext_a)
        c->prof.ext_a = a;
        c->prof.ext_b = b;
        return;
    }
    (void)hipEventRecord(a, c->stream);
}
ProfScope::~ProfScope() {
    if (!a) return;
    if (ctx->prof.level == 2) {
        const bool used = ctx->prof.ext_a == nullptr;   // the launch took the events
        ctx->prof.ext_a = ctx->prof.ext_b = nullptr;
        if (!used) {   // nothing was launched inside the scope
            ctx->prof.pool.push_back(a);
            ctx->prof.pool.push_back(b);
            return;
        }
    } else {
        (void)hipEventRecord(b, ctx->stream);
    }
    ctx->prof.pending.push_back({idx, a, b});
}


// ---- the far slice of a depth-sliced forward (see bh_render_forward) -------------------------------------------------------------
// count -> emit (the scan between them folded into the emit kernel) the remaining splats into the tiles that still have live pixels, sort them behind the near list (absolute
// offsets: one array for the backward), blend from the parked state.  Every kernel is gated on the device by the number of
// unsaturated tiles, so queueing it for a frame that does not need it is correct, just ~50 us of empty launches.
int enqueue_far_slice(bh_ctx* ctx, const FarJob& j) {
    const uint32_t* gate = j.slice_info + 2;
    const uint32_t far_max = j.ni;   // the host's bound; the live count is slice_info[3] on the device (the emit kernel's last block)
    {
        ProfScope ps(ctx, "MapGaussiansToIntersect");
        BH_TRY(launch_map_gaussians_far(ctx, j.nv, j.u, j.proj_by_gid, j.gfc, j.projected, j.cum, j.budget, j.done_bits, gate, j.far_counts, j.far_block_totals,
                                        j.far_group_totals, j.slice_info, j.tile_ids, j.isect_gids));
    }
    {
        ProfScope ps(ctx, "TileSort");
        BH_TRY(radix_argsort_dev(ctx, j.tile_ids, j.isect_gids, far_max, j.slice_info + 3, gate, j.slice_info + 1, j.tile_bits, j.tile_ids_sorted,
                                 j.isect_gids_sorted));
    }
    {
        ProfScope ps(ctx, "GetTileOffsets");
        BH_TRY(launch_tile_offsets_dev(ctx, j.tile_ids_sorted, far_max, j.slice_info + 3, gate, j.slice_info + 1, j.num_tiles, j.tile_offsets_far));
    }
    {
        ProfScope ps(ctx, "Rasterize");
        BH_TRY(launch_rasterize(ctx, j.u, j.bg, j.bwd_info, j.smooth, j.isect_gids_sorted, j.tile_offsets_far, j.projected, j.gfc, j.out_f32, j.out_u8, j.visible,
                                j.lpt, j.class_width, /*phase=*/2, &j.rs));
    }
    ctx->far_launches++;
    return 0;
}

// Which table a frame uses.  A caller that knows its views names them (bh_set_view_id / BhTrainBatch.view_id); one that does not
// — the reference's SplatTrainer::step receives a SceneBatch without a view index (train.rs:176, brush-dataset/src/scene.rs:138-147)
// — is keyed by the camera itself: a dataset's views are fixed cameras, and the same camera gives the same bits every time.
// Bit 63 separates the two key spaces.
static uint64_t view_key(const bh_ctx* ctx, const BhCamera& c) {
    if (ctx->view_id != 0u || ctx->knob_no_view_hash) return (uint64_t)ctx->view_id;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    auto mix = [&](uint32_t w) {   // splitmix64 finaliser over a running sum: order-sensitive, cheap, well spread
        h += (uint64_t)w + 0x9E3779B97F4A7C15ull;
        uint64_t z = h;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        h = z ^ (z >> 31);
    };
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    for (int i = 0; i < 12; ++i) mix(bits(c.vm[i]));
    mix(bits(c.fx)); mix(bits(c.fy)); mix(bits(c.cx)); mix(bits(c.cy));
    mix(c.img_w); mix(c.img_h); mix(c.tile_row_begin); mix(c.tile_row_end); mix(c.model);
    if (c.model != BH_CAMERA_PINHOLE) for (int i = 0; i < 8; ++i) mix(bits(c.dist[i]));
    return h | (1ull << 63);
}

// The per-tile depth-cut table of view `key` for a (tile_bw x tile_bh) grid: created (all ZCUT_ALL = "list everything") on first
// use, re-created when the grid changes; beyond MAX_VIEW_STATES tables (or VIEW_TABLE_BYTES of them) the least recently used view
// gives its table up — to the new view when the grids match (no free, no host wait: the clears are ordered on the stream).
// touch = false: a second attempt at the frame that has just been counted (finish_far_slice): the view's gap and stamp stay
// casual = a forward-only frame keyed by its camera hash (viewer / eval renders): never more than CASUAL_VIEW_STATES such tables,
// and none at all for a camera met for the first time (returns nullptr: the frame runs in index order, nothing is allocated).
static ViewState* view_state(bh_ctx* ctx, uint64_t key, uint32_t tile_bw, uint32_t tile_bh, bool touch = true, bool casual = false) {
    const size_t words = (size_t)tile_bw * tile_bh ? (size_t)tile_bw * tile_bh : 1;
    auto it = ctx->views.find(key);
    uint32_t* recycled = nullptr;
    auto forget = [&](std::unordered_map<uint64_t, ViewState>::iterator v, bool keep_block) {
        if (ctx->gate_view == &v->second) ctx->gate_view = nullptr;
        if (ctx->far_job.view == &v->second) ctx->far_job.view = nullptr;
        if (v->second.casual && ctx->casual_views) ctx->casual_views--;
        if (keep_block) recycled = v->second.zcut;
        else {
            (void)hipStreamSynchronize(ctx->stream);   // queued kernels may still use the block
            (void)hipFree(v->second.zcut);
        }
        ctx->views.erase(v);
    };
    if (it != ctx->views.end() && (it->second.tile_bw != tile_bw || it->second.tile_bh != tile_bh)) {
        forget(it, false);
        it = ctx->views.end();
    }
    if (it == ctx->views.end()) {
        if (casual) {
            bool seen = false;
            for (uint64_t k : ctx->seen_keys) seen = seen || k == key;
            if (!seen) {   // first meeting: remember the camera, allocate nothing
                ctx->seen_keys[ctx->seen_pos++ % SEEN_KEYS] = key;
                return nullptr;
            }
            while (ctx->casual_views >= CASUAL_VIEW_STATES) {   // the least recently used casual table makes room (its block is reused when the grids match)
                auto old = ctx->views.end();
                for (auto k = ctx->views.begin(); k != ctx->views.end(); ++k)
                    if (k->second.casual && (old == ctx->views.end() || k->second.last_used < old->second.last_used)) old = k;
                if (old == ctx->views.end()) { ctx->casual_views = 0; break; }
                forget(old, recycled == nullptr && old->second.tile_bw == tile_bw && old->second.tile_bh == tile_bh);
            }
        }
        const size_t max_views = std::min(MAX_VIEW_STATES, std::max<size_t>(8, VIEW_TABLE_BYTES / ((2 * words + VIEW_SPL_WORDS) * 4)));
        while (ctx->views.size() >= max_views) {
            auto old = ctx->views.begin();
            for (auto k = ctx->views.begin(); k != ctx->views.end(); ++k)
                if (k->second.last_used < old->second.last_used) old = k;
            forget(old, recycled == nullptr && old->second.tile_bw == tile_bw && old->second.tile_bh == tile_bh);
        }
        ViewState vs;
        vs.tile_bw = tile_bw;
        vs.tile_bh = tile_bh;
        vs.casual = casual;
        // [T] depth cuts (all "everything") | [T] per-tile work of the last frame (all zero) | [VIEW_SPL_WORDS] depth-sort splitter tables (none valid)
        vs.zcut = recycled;
        if (!vs.zcut && hipMalloc((void**)&vs.zcut, (2 * words + VIEW_SPL_WORDS) * 4) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(vs.zcut), (int)ZCUT_ALL, words, ctx->stream) != hipSuccess ||
            hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(vs.zcut + words), 0, words + VIEW_SPL_WORDS, ctx->stream) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(vs.zcut);
            return nullptr;
        }
        vs.gap = (uint32_t)ctx->views.size() + 1u;   // (a new view of a dataset: it will come back after about as many frames as there are views)
        vs.last_used = ++ctx->view_clock;
        if (casual) ctx->casual_views++;
        return &ctx->views.emplace(key, vs).first->second;
    }
    if (it->second.casual && !casual) {   // a training frame adopts the table: it now counts as a dataset view
        it->second.casual = false;
        if (ctx->casual_views) ctx->casual_views--;
    }
    if (touch) {
        const uint64_t now = ++ctx->view_clock;
        it->second.gap = (uint32_t)std::min<uint64_t>(now - it->second.last_used, 1u << 20);
        it->second.last_used = now;
    }
    return &it->second;
}

// margin (in % of a tile's depth rank) the blend kernel of this frame writes behind every tile's last useful splat
static uint32_t cut_margin_pct(const bh_ctx* ctx, const ViewState* vs) {
    const float gap = vs && vs->gap > 2u ? (float)vs->gap : 2.0f;
    const float m = (float)ctx->knob_cut_margin_pct * ctx->margin_scale * std::pow(gap * 0.5f, ctx->ctrl_gap_exp);
    return m < 6400.0f ? (m > 10.0f ? (uint32_t)m : 10u) : 6400u;
}

// Outcome of a per-tile-cut frame of `vs`: did the forecast fail for some tile (the frame was then rendered a second time with
// complete lists, which re-seeds the table)?  Every outcome moves the ctx's margin factor (x ctrl_up on a miss, x ctrl_down on a
// hit: about one miss in 200 cut frames at equilibrium).  Six misses within the view's last eight cut frames (a scene that
// changes faster than any margin) and the view's next eight frames are rendered with complete lists from the start.
static void view_outcome(bh_ctx* ctx, ViewState* vs, bool missed, bool shared_table) {
    if (!ctx->knob_fixed_margin) {
        const float s = ctx->margin_scale * (missed ? ctx->ctrl_up : ctx->ctrl_down);
        ctx->margin_scale = s < ctx->ctrl_floor ? ctx->ctrl_floor : (s > 16.0f ? 16.0f : s);
    }
    if (!vs) return;
    vs->penalty = ((vs->penalty << 1) | (missed ? 1u : 0u)) & 0xFFu;   // (the history of the last eight cut frames, one bit each)
    // (BH_NO_VIEW_HASH only: the table of view id 0 shared by every frame that names no view — alternating cameras miss on every
    //  other frame there: three misses are enough, and the table stays untrusted for longer)
    if (__builtin_popcount(vs->penalty) >= (shared_table ? 3 : 6)) {
        vs->exact_frames = shared_table ? 32u : 8u;
        vs->penalty = 0u;
    }
}

static int forward_impl(bh_ctx* ctx, const BhCamera* cam, uint32_t n, uint32_t sh_degree, const float* transforms, const float* sh_coeffs,
                        const float* raw_opacities, const float* background, uint32_t flags, BhRenderOut* out, bool allow_cut);

// A sliced forward that left the decision to the host: wait for the near pass's gate word.  Slot-budget slices: queue the far slice
// if some tile is still unsaturated.  Per-tile cuts: there is no far pass (only the splats that own a near pair were sorted) — a
// tile that is still live behind a cut list means the view's forecast failed, and the whole forward is run again with complete
// lists (which also re-seeds the view's table).  *launched (optional) tells the caller whether out_img changed after the near pass.
int finish_far_slice(bh_ctx* ctx, bool* launched) {
    if (launched) *launched = false;
    if (!ctx->far_job.pending) return 0;
    ctx->far_job.pending = false;
    if (!ctx->far_job.gate_event_recorded && ctx->gate_signal_queued) {
        // (bh_train_step: its loss kernel, queued behind the near blend, stores the job's tag when it starts)
        BH_TRY(wait_host_tag(ctx, reinterpret_cast<const volatile uint32_t*>(ctx->host_counters) + HOST_GATE_TAG_WORD, ctx->far_job.gate_tag, "near-pass gate"));
    } else {
        if (!ctx->far_job.gate_event_recorded) BH_HIP(ctx, hipEventRecord(ctx->gate_ev, ctx->stream));   // (nobody queued a signal: an event behind whatever is queued now)
        BH_HIP(ctx, hipEventSynchronize(ctx->gate_ev));
    }
    ctx->gate_signal_queued = false;
    const uint32_t unsat = reinterpret_cast<const volatile uint32_t*>(ctx->host_counters)[HOST_GATE_WORD];
    if (ctx->far_job.by_cut) {
        FarJob& j = ctx->far_job;
        view_outcome(ctx, j.view, unsat != 0u, j.view_shared);
        if (unsat == 0u) return 0;
        if (launched) *launched = true;
        ctx->far_launches++;
        // the same call again, with the redirections of the train step that were in force and the same view, complete lists
        const FarJob keep = j;
        struct Saved { float* ev; float* er; size_t evf; float* eg; size_t egf; uint32_t vid; bool defer; } sv{ctx->ext_visible, ctx->ext_max_radius, ctx->ext_visible_floats,
                                                                                                      ctx->ext_grad_begin, ctx->ext_grad_floats, ctx->view_id, ctx->defer_far};
        ctx->ext_visible = keep.ext_visible; ctx->ext_max_radius = keep.ext_max_radius; ctx->ext_visible_floats = keep.ext_visible_floats;
        ctx->ext_grad_begin = keep.ext_grad_begin; ctx->ext_grad_floats = keep.ext_grad_floats;
        ctx->view_id = keep.view_id;
        ctx->defer_far = false;
        BhRenderOut again;
        const int rc = forward_impl(ctx, &keep.cam, keep.n, keep.sh_degree, keep.transforms, keep.sh_coeffs, keep.raw_opacities, keep.bg, keep.flags, &again,
                                    /*allow_cut=*/false);
        ctx->ext_visible = sv.ev; ctx->ext_max_radius = sv.er; ctx->ext_visible_floats = sv.evf;
        ctx->ext_grad_begin = sv.eg; ctx->ext_grad_floats = sv.egf;
        ctx->view_id = sv.vid;
        ctx->defer_far = sv.defer;
        return rc;
    }
    ctx->far_direct = unsat != 0u;   // ... and the next sliced frame starts from what this one needed
    if (unsat == 0u) return 0;
    if (launched) *launched = true;
    return enqueue_far_slice(ctx, ctx->far_job);
}

}  // namespace bh

using namespace bh;

extern "C" {

const char* bh_version(void) { return "brush_hip 0.1 (gfx950)"; }

uint32_t bh_abi_version(void) { return BH_ABI_VERSION; }

uint32_t bh_struct_size(uint32_t which) {
    switch (which) {
        case BH_STRUCT_CAMERA: return (uint32_t)sizeof(BhCamera);
        case BH_STRUCT_RENDER_OUT: return (uint32_t)sizeof(BhRenderOut);
        case BH_STRUCT_LOSS_CONFIG: return (uint32_t)sizeof(BhLossConfig);
        case BH_STRUCT_TRAIN_CONFIG: return (uint32_t)sizeof(BhTrainConfig);
        case BH_STRUCT_TRAIN_STATE: return (uint32_t)sizeof(BhTrainState);
        case BH_STRUCT_TRAIN_BATCH: return (uint32_t)sizeof(BhTrainBatch);
        case BH_STRUCT_TRAIN_STATS: return (uint32_t)sizeof(BhTrainStats);
        case BH_STRUCT_REFINE_CONFIG: return (uint32_t)sizeof(BhRefineConfig);
        case BH_STRUCT_REFINE_STATS: return (uint32_t)sizeof(BhRefineStats);
        case BH_STRUCT_PLY_INFO: return (uint32_t)sizeof(BhPlyInfo);
        default: return 0u;
    }
}

bh_ctx* bh_create(int device, void* stream, int own_stream) {
    bh_ctx* ctx = new (std::nothrow) bh_ctx();
    if (!ctx) return nullptr;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        delete ctx;
        return nullptr;
    }
    if (!own_stream) {
        ctx->stream = (hipStream_t)stream;
        ctx->owns_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            delete ctx;
            return nullptr;
        }
        ctx->owns_stream = true;
    }
    if (hipHostMalloc((void**)&ctx->host_counters, bh::HOST_COUNTERS_BYTES, hipHostMallocCoherent)   /* (polled by the host while kernels store into it: never the non-coherent flavour HIP_HOST_COHERENT=0 would make of the default) */ != hipSuccess) {
        (void)hipGetLastError();
        if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return nullptr;
    }
    std::memset(ctx->host_counters, 0, bh::HOST_COUNTERS_BYTES);
    // No environment variable configures the shipping library: every tuning / A-B knob is a documented bh_set_option key.
#ifdef BH_TEST_HOOKS   // libbrush_hip_testhooks.so only (csrc/Makefile): the shipping library neither reads these nor exports the hook below
    ctx->knob_break_allreduce = getenv("BH_BREAK_ALLREDUCE") != nullptr;
    if (ctx->knob_break_allreduce)   // (bench.py's exchange self-check must catch it): never silent
        fprintf(stderr, "brush_hip: BH_BREAK_ALLREDUCE is set - every all-reduce of this context's communicator is deliberately CORRUPTED (test hook)\n");
    if (const char* e = getenv("BH_TEST_FAIL_LOSS_AT")) {   // (tests/test_gpu_sliced.py): the k-th train step on this ctx fails between its forward and its loss
        ctx->knob_fail_loss_at = (uint32_t)atoi(e);
        if (ctx->knob_fail_loss_at) fprintf(stderr, "brush_hip: BH_TEST_FAIL_LOSS_AT=%u - that train step of this context will FAIL on purpose (test hook)\n", ctx->knob_fail_loss_at);
    }
#endif
    if (hipEventCreateWithFlags(&ctx->readback_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->gate_ev, hipEventDisableTiming) != hipSuccess) {
        if (ctx->readback_ev) (void)hipEventDestroy(ctx->readback_ev);
        (void)hipGetLastError();
        (void)hipHostFree(ctx->host_counters);
        if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void bh_destroy(bh_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    prof_resolve(ctx);
    for (hipEvent_t e : ctx->prof.pool) (void)hipEventDestroy(e);
    for (auto& b : ctx->slots)
        if (b.ptr) (void)hipFree(b.ptr);
    for (auto& kv : ctx->views)
        if (kv.second.zcut) (void)hipFree(kv.second.zcut);
    for (auto& rt : ctx->retained)
        for (auto& b : rt.blocks)
            if (b.ptr) (void)hipFree(b.ptr);
    for (auto& b : ctx->pool)
        if (b.ptr) (void)hipFree(b.ptr);
    if (ctx->host_counters) (void)hipHostFree(ctx->host_counters);
    if (ctx->dsort_spl) (void)hipFree(ctx->dsort_spl);
    if (ctx->readback_ev) (void)hipEventDestroy(ctx->readback_ev);
    if (ctx->gate_ev) (void)hipEventDestroy(ctx->gate_ev);
    if (ctx->comm) (void)bh_comm_destroy(ctx);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* bh_last_error(bh_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int bh_sync(bh_ctx* ctx) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (ctx->far_job.pending) {   // (a deferred decision may queue kernels: on the ctx's device)
        BH_HIP(ctx, hipSetDevice(ctx->device));
        BH_TRY(finish_far_slice(ctx, nullptr));
    }
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    deliver_pending_loss(ctx);  // the last train step's loss, staged through pinned memory
    return 0;
}

int bh_profile_enable(bh_ctx* ctx, int on) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    ctx->prof.level = on < 0 ? 0 : (on > 2 ? 1 : on);
    return 0;
}

int bh_profile_fetch(bh_ctx* ctx, const char** names, float* ms, uint32_t* calls, int cap) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    prof_resolve(ctx);
    Profiler& p = ctx->prof;
    const int n = p.count < cap ? p.count : cap;
    for (int i = 0; i < n; ++i) {
        names[i] = p.names[i];
        ms[i] = p.ms[i];
        calls[i] = p.calls[i];
    }
    p.count = 0;
    return n;
}

// ---- options ---------------------------------------------------------------------
// One documented setter for everything earlier revisions read from BH_* environment variables at bh_create (include/brush_hip.h
// lists the keys).  Options select between paths that give the SAME results (A/B measurements, tests of the alternative paths);
// they are host fields read when the next call is queued.
namespace {
struct OptionKey { const char* name; const char* help; };
const OptionKey kOptionKeys[] = {
    {"cut_min_pairs", "u32: a view whose last frame had fewer pairs keeps complete lists (= bh_set_list_cut_threshold)"},
    {"cut_margin_pct", "0..10000: base margin behind a tile's last useful splat, % of its depth rank (default 150)"},
    {"cut_margin_fixed", "0|1: the margin is cut_margin_pct for every frame instead of adaptive"},
    {"cut_ctrl", "up:down:floor:gap_exp — the margin controller's constants (default 1.5:0.998:0.5:0.3333)"},
    {"cut_sort_all", "0|1: with per-tile cuts, still depth-sort every visible splat"},
    {"auto_exact_share", "0..1: a view whose last cut frame listed more than this share of its pairs renders complete lists (default 0.9; 0 = never)"},
    {"no_view_hash", "0|1: frames without a view id share ONE table instead of being keyed by their camera"},
    {"k16_order", "0 index order | 1 by the view's last per-tile work | 2 dealt: the forward blend's tile order"},
    {"band_mode", "0|1: XCD bands of the blend kernels — contiguous eighths of the tile range, or dealt in chunks of 8 adjacent tiles (default 1)"},
    {"k16_waves", "0..8: forward blend: resident one-wave tiles per SIMD (0 = 8 = all resident at once; fewer: the lighter tiles are dispatched as the heavier ones finish)"},
    {"k16_split", "0..1000: forward blend: a tile whose forecast work is at least max(256, k16_split / 100 x its band's mean) is blended by four quadrant waves (default 250; 0: no tile is split)"},
    {"k16_split_of_max", "0..100: ... and at least this many percent of its band's heaviest tile (default 45)"},
    {"k16_split_min", "1..1023: a tile below this many blended splats (forecast) is never split (default 256)"},
    {"k5_exact_spw", "16|32|64: splats per wave of the list builder for complete lists"},
    {"bwd_jobs", "0|1: the blend backward works on checkpointed 128-entry segments of the tiles' lists (default 1) or on whole tiles"},
    {"no_lpt", "0|1: the blend backward takes its tiles in index order"},
    {"lpt_classes", "log|linear: work classes of the backward's longest-first tile order — two per octave of blended splats, or 1/64 of the mean list length wide"},
    {"generic_depth_sort", "0|1: depth order by the generic radix sort + scan instead of the fused split sort"},
    {"dsort_splitters", "0|1: the fused depth sort splits at the 254 depth quantiles of the view's previous frame (default 1) or always linearly over the frame's key range"},
    {"tile_sort", "auto|bucket|lsd: the forward's tile sort (auto: bucket sort unless the view's pairs are concentrated in few tiles)"},
    {"spec_k5", "0|1: queue the list builder before the host has read the frame's counts (default 1; 0: behind the count readback)"},
    {"event_waits", "0|1: the host's mid-step waits use events behind the kernels instead of polled tag words"},
    {"readback_copy", "0|1: counts and gate word reach the host through copy launches"},
    {"force_exchange", "0|1: a one-rank communicator still walks the whole gradient-exchange path (overhead measurement)"},
    {"zero_grads", "0|1: the single-GPU train step zero-fills its gradient span like the exchange path"},
    {"loss_bands", "0|1: the fused loss's blocks take their tiles by XCD column bands (1) or row-major (0)"},
    {"update_rows", "0|64|128|256: splats per block of the update kernel (0 = default)"},
    {"update_early", "0|1: the update kernel's blocks issue all their loads up front"},
    {"no_dormant", "0|1: the update kernel fetches and updates dormant splats like everyone else"},
    {"sort_kpt", "0|4|8|16: keys per thread of the generic radix sort (0 = default)"},
    {"grad_allreduce", "ring|direct: the dense gradient block's collective — ncclAllReduce, or reduce-scatter + all-gather over grouped send/recv"},
};
bool parse_u32(const char* v, uint32_t lo, uint32_t hi, uint32_t* out) {
    if (!v || !*v) return false;
    char* end = nullptr;
    const unsigned long long x = strtoull(v, &end, 10);
    if (*end != '\0' || x < lo || x > hi) return false;
    *out = (uint32_t)x;
    return true;
}
bool parse_flag(const char* v, bool* out) {
    uint32_t x = 0;
    if (!parse_u32(v, 0, 1, &x)) return false;
    *out = x != 0;
    return true;
}
}  // namespace

extern "C" int bh_option_count(void) { return (int)(sizeof(kOptionKeys) / sizeof(kOptionKeys[0])); }
extern "C" const char* bh_option_name(int i) { return i >= 0 && i < bh_option_count() ? kOptionKeys[i].name : nullptr; }
extern "C" const char* bh_option_help(int i) { return i >= 0 && i < bh_option_count() ? kOptionKeys[i].help : nullptr; }

extern "C" int bh_set_option(bh_ctx* ctx, const char* key, const char* value) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!key || !value) return set_error(ctx, BH_ERR_INVALID_ARG, "set_option: null key or value");
    const std::string k(key);
    bool ok = false;
    uint32_t u = 0;
    if (k == "cut_min_pairs") { if ((ok = parse_u32(value, 0, 0xFFFFFFFFu, &u))) ctx->cut_min_pairs = u; }
    else if (k == "cut_margin_pct") { if ((ok = parse_u32(value, 0, 10000, &u))) ctx->knob_cut_margin_pct = u; }
    else if (k == "cut_margin_fixed") ok = parse_flag(value, &ctx->knob_fixed_margin);
    else if (k == "cut_ctrl") {
        float a = 0, b = 0, c = 0, d = 0;
        if (sscanf(value, "%f:%f:%f:%f", &a, &b, &c, &d) == 4 && a >= 1.0f && b > 0.0f && b <= 1.0f && c > 0.0f && d >= 0.0f && d <= 1.0f) {
            ctx->ctrl_up = a; ctx->ctrl_down = b; ctx->ctrl_floor = c; ctx->ctrl_gap_exp = d;
            ok = true;
        }
    }
    else if (k == "cut_sort_all") ok = parse_flag(value, &ctx->knob_cut_sort_all);
    else if (k == "auto_exact_share") {
        char* end = nullptr;
        const float f = strtof(value, &end);
        if (end != value && *end == '\0' && f >= 0.0f && f <= 1.0f) { ctx->auto_exact_share = f; ok = true; }
    }
    else if (k == "no_view_hash") ok = parse_flag(value, &ctx->knob_no_view_hash);
    else if (k == "k16_order") { if ((ok = parse_u32(value, 0, 2, &u))) ctx->knob_k16_order = u; }
    else if (k == "band_mode") { if ((ok = parse_u32(value, 0, 1, &u))) ctx->knob_band_mode = u; }
    else if (k == "k16_waves") { if ((ok = parse_u32(value, 0, 8, &u))) ctx->knob_k16_waves = u; }
    else if (k == "k16_split") { if ((ok = parse_u32(value, 0, 1000, &u))) ctx->knob_k16_split = u; }
    else if (k == "k16_split_of_max") { if ((ok = parse_u32(value, 0, 100, &u))) ctx->knob_k16_split_of_max = u; }
    else if (k == "k16_split_min") { if ((ok = parse_u32(value, 1, 1023, &u))) ctx->knob_k16_split_min = u; }
    else if (k == "k5_exact_spw") { if ((ok = parse_u32(value, 16, 64, &u) && (u == 16 || u == 32 || u == 64))) ctx->knob_k5_exact_spw = u; }
    else if (k == "bwd_jobs") ok = parse_flag(value, &ctx->knob_bwd_jobs);
    else if (k == "no_lpt") ok = parse_flag(value, &ctx->knob_no_lpt);
    else if (k == "lpt_classes") {
        const std::string v(value);
        if (v == "log") { ctx->knob_lpt_linear = false; ok = true; }
        else if (v == "linear") { ctx->knob_lpt_linear = true; ok = true; }
    }
    else if (k == "generic_depth_sort") ok = parse_flag(value, &ctx->knob_generic_depth_sort);
    else if (k == "dsort_splitters") ok = parse_flag(value, &ctx->knob_dsort_splitters);
    else if (k == "tile_sort") {
        const std::string v(value);
        if (v == "auto") { ctx->knob_tile_sort = 0; ok = true; }
        else if (v == "bucket") { ctx->knob_tile_sort = 1; ok = true; }
        else if (v == "lsd") { ctx->knob_tile_sort = 2; ok = true; }
    }
    else if (k == "spec_k5") ok = parse_flag(value, &ctx->knob_spec_k5);
    else if (k == "event_waits") ok = parse_flag(value, &ctx->knob_event_waits);
    else if (k == "readback_copy") ok = parse_flag(value, &ctx->knob_readback_copy);
    else if (k == "force_exchange") ok = parse_flag(value, &ctx->knob_force_exchange);
    else if (k == "zero_grads") ok = parse_flag(value, &ctx->knob_zero_grads);
    else if (k == "loss_bands") { if ((ok = parse_u32(value, 0, 1, &u))) ctx->knob_loss_bands = u; }
    else if (k == "update_rows") { if ((ok = parse_u32(value, 0, 256, &u) && (u == 0 || u == 64 || u == 128 || u == 256))) ctx->knob_update_rows = u; }
    else if (k == "update_early") ok = parse_flag(value, &ctx->knob_update_early);
    else if (k == "no_dormant") ok = parse_flag(value, &ctx->knob_no_dormant);
    else if (k == "sort_kpt") { if ((ok = parse_u32(value, 0, 16, &u) && (u == 0 || u == 4 || u == 8 || u == 16))) ctx->knob_sort_kpt = u; }
    else if (k == "grad_allreduce") {
        const std::string v(value);
        if (v == "ring") { ctx->knob_direct_allreduce = false; ok = true; }
        else if (v == "direct") { ctx->knob_direct_allreduce = true; ok = true; }
    }
    else return set_error(ctx, BH_ERR_INVALID_ARG, "set_option: unknown key '" + k + "' (bh_option_name lists the keys)");
    if (!ok) return set_error(ctx, BH_ERR_INVALID_ARG, "set_option: bad value '" + std::string(value) + "' for '" + k + "'");
    return 0;
}

// ---- lens laws in f64 (brush-render/src/camera.rs:85-198) ------------------------------------
namespace {
struct Kb4Law {  // d(theta) = theta + k1 theta^3 + k2 theta^5 + k3 theta^7 + k4 theta^9   (camera.rs:120-142)
    double k[4];
    explicit Kb4Law(const float* d) { for (int i = 0; i < 4; ++i) k[i] = d ? (double)d[i] : 0.0; }
    double value(double th) const {
        const double t2 = th * th, t3 = t2 * th, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
        return th + k[0] * t3 + k[1] * t5 + k[2] * t7 + k[3] * t9;
    }
    double slope(double th) const {
        const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
        return 1.0 + 3.0 * k[0] * t2 + 5.0 * k[1] * t4 + 7.0 * k[2] * t6 + 9.0 * k[3] * t8;
    }
    // Newton on d(theta) = target, theta in [0, pi] (camera.rs:145-167)
    double invert(double target) const {
        if (target <= 0.0) return 0.0;
        const double pi = 3.14159265358979323846;
        double th = target < pi - 1e-6 ? target : pi - 1e-6;
        for (int it = 0; it < 50; ++it) {
            const double fp = slope(th);
            if (std::fabs(fp) < 1e-12) break;
            double next = th - (value(th) - target) / fp;
            next = next < 0.0 ? 0.0 : (next > pi ? pi : next);
            const bool done = std::fabs(next - th) < 1e-12;
            th = next;
            if (done) break;
        }
        return th;
    }
};
struct Rt8Law {  // radial factor (1 + k1 r^2 + k2 r^4 + k3 r^6) / (1 + k4 r^2 + k5 r^4 + k6 r^6)   (camera.rs:170-178)
    double k[6];
    explicit Rt8Law(const float* d) { for (int i = 0; i < 6; ++i) k[i] = d ? (double)d[i] : 0.0; }
    double radial(double r) const {
        const double r2 = r * r, r4 = r2 * r2, r6 = r4 * r2;
        return (1.0 + k[0] * r2 + k[1] * r4 + k[2] * r6) / (1.0 + k[3] * r2 + k[4] * r4 + k[5] * r6);
    }
    // fixed-point r <- r_d / radial(r)   (camera.rs:182-198)
    double undistort(double r_d) const {
        double r = r_d;
        for (int it = 0; it < 30; ++it) {
            const double f = radial(r);
            if (std::fabs(f) < 1e-12) break;
            const double rn = r_d / f;
            const bool done = std::fabs(rn - r) < 1e-12;
            r = rn;
            if (done) break;
        }
        return r;
    }
};
}  // namespace

double bh_fov_to_focal(double fov, uint32_t pixels, uint32_t model, const float* dist) {
    const double half = fov / 2.0, r_pix = (double)pixels / 2.0;
    switch (model) {
        case BH_CAMERA_PINHOLE: return r_pix / std::tan(half);
        case BH_CAMERA_KANNALA_BRANDT_4:
        case BH_CAMERA_THIN_PRISM_FISHEYE: return r_pix / Kb4Law(dist).value(half);
        case BH_CAMERA_RADIAL_TANGENTIAL_8: { const double r = std::tan(half); return r_pix / (r * Rt8Law(dist).radial(r)); }
        default: return std::nan("");
    }
}

double bh_focal_to_fov(double focal, uint32_t pixels, uint32_t model, const float* dist) {
    const double r_norm = ((double)pixels / 2.0) / focal;
    switch (model) {
        case BH_CAMERA_PINHOLE: return 2.0 * std::atan(r_norm);
        case BH_CAMERA_KANNALA_BRANDT_4:
        case BH_CAMERA_THIN_PRISM_FISHEYE: return 2.0 * Kb4Law(dist).invert(r_norm);
        case BH_CAMERA_RADIAL_TANGENTIAL_8: return 2.0 * std::atan(Rt8Law(dist).undistort(r_norm));
        default: return std::nan("");
    }
}

// brush-render/src/camera.rs:63-101,200-254, render.rs:70-71; glam 0.30 Affine3A/Mat3A restated in f32.
int bh_camera_setup_model(const float* pos, const float* rot_xyzw, double fov_x, double fov_y, float center_u, float center_v,
                          uint32_t img_w, uint32_t img_h, uint32_t model, const float* dist, BhCamera* out) {
    if (!pos || !rot_xyzw || !out || img_w == 0 || img_h == 0 || model > BH_CAMERA_THIN_PRISM_FISHEYE) return BH_ERR_INVALID_ARG;
    const float x = rot_xyzw[0], y = rot_xyzw[1], z = rot_xyzw[2], w = rot_xyzw[3];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx = x * x2, xy = x * y2, xz = x * z2;
    const float yy = y * y2, yz = y * z2, zz = z * z2;
    const float wx = w * x2, wy = w * y2, wz = w * z2;
    // camera-to-world rotation columns (Mat3A::from_quat)
    const float ax[3] = {1.0f - (yy + zz), xy + wz, xz - wy};
    const float ay[3] = {xy - wz, 1.0f - (xx + zz), yz + wx};
    const float az[3] = {xz + wy, yz - wx, 1.0f - (xx + yy)};
    auto cross = [](const float* a, const float* b, float* o) {
        o[0] = a[1] * b[2] - b[1] * a[2];
        o[1] = a[2] * b[0] - b[2] * a[0];
        o[2] = a[0] * b[1] - b[0] * a[1];
    };
    float t0[3], t1[3], t2[3];
    cross(ay, az, t0);
    cross(az, ax, t1);
    cross(ax, ay, t2);
    const float det = (az[0] * t2[0]) + (az[1] * t2[1]) + (az[2] * t2[2]);
    const float inv_det = 1.0f / det;
    // Mat3A::inverse = from_cols(t0, t1, t2) * inv_det, transposed
    float r0[3], r1[3], r2[3];
    for (int i = 0; i < 3; ++i) { r0[i] = t0[i] * inv_det; r1[i] = t1[i] * inv_det; r2[i] = t2[i] * inv_det; }
    const float c0[3] = {r0[0], r1[0], r2[0]}, c1[3] = {r0[1], r1[1], r2[1]}, c2[3] = {r0[2], r1[2], r2[2]};
    for (int i = 0; i < 3; ++i) {
        out->vm[i] = c0[i];
        out->vm[3 + i] = c1[i];
        out->vm[6 + i] = c2[i];
        const float ip = (c0[i] * pos[0] + c1[i] * pos[1]) + c2[i] * pos[2];
        out->vm[9 + i] = -ip;
    }
    out->model = model;
    for (int i = 0; i < 8; ++i) out->dist[i] = (model != BH_CAMERA_PINHOLE && dist) ? dist[i] : 0.0f;
    out->fx = (float)bh_fov_to_focal(fov_x, img_w, model, out->dist);
    out->fy = (float)bh_fov_to_focal(fov_y, img_h, model, out->dist);
    out->cx = center_u * (float)img_w;
    out->cy = center_v * (float)img_h;
    // calculate_jacobian_clamp_limits (camera.rs:200-254): image edges +-15 %, in normalised coordinates
    const float wf = (float)img_w, hf = (float)img_h;
    float lim[4] = {(1.15f * wf - out->cx) / out->fx, (1.15f * hf - out->cy) / out->fy,
                    (-0.15f * wf - out->cx) / out->fx, (-0.15f * hf - out->cy) / out->fy};
    if (model == BH_CAMERA_RADIAL_TANGENTIAL_8) {
        // the clamp bounds the UNDISTORTED coordinate: invert the radial law at each edge
        const Rt8Law law(out->dist);
        for (float& e : lim) {
            const float mag = (float)law.undistort(std::fabs((double)e));
            e = (e != e) ? e : (std::signbit(e) ? -mag : mag);  // * f32::signum(edge)
        }
    } else if (model != BH_CAMERA_PINHOLE) {
        for (float& e : lim) e = 0.0f;  // fisheye Jacobians are not clamped
    }
    out->lim_pos_x = lim[0]; out->lim_pos_y = lim[1]; out->lim_neg_x = lim[2]; out->lim_neg_y = lim[3];
    // render.rs:70-71 (f32)
    const float full = hypotf((float)fov_x, (float)fov_y) * 1.05f;
    const float cap = 2.0f * 3.14159265358979323846f - 1e-6f;
    out->half_max_render_fov = (full < cap ? full : cap) * 0.5f;
    out->cam_pos[0] = pos[0]; out->cam_pos[1] = pos[1]; out->cam_pos[2] = pos[2];
    out->img_w = img_w;
    out->img_h = img_h;
    out->tile_row_begin = 0;
    out->tile_row_end = 0;
    return 0;
}

int bh_camera_setup(const float* pos, const float* rot_xyzw, double fov_x, double fov_y, float center_u, float center_v,
                    uint32_t img_w, uint32_t img_h, BhCamera* out) {
    return bh_camera_setup_model(pos, rot_xyzw, fov_x, fov_y, center_u, center_v, img_w, img_h, BH_CAMERA_PINHOLE, nullptr, out);
}

// ---- forward -------------------------------------------------------------------
int bh_render_forward(bh_ctx* ctx, const BhCamera* cam, uint32_t n, uint32_t sh_degree, const float* transforms,
                      const float* sh_coeffs, const float* raw_opacities, const float* background, uint32_t flags,
                      BhRenderOut* out) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!cam || !out || !background) return set_error(ctx, BH_ERR_INVALID_ARG, "render_forward: null argument");
    if (cam->img_w == 0 || cam->img_h == 0) return set_error(ctx, BH_ERR_INVALID_ARG, "Can't render images with 0 size.");  // render.rs:50-53
    if (sh_degree > 4) return set_error(ctx, BH_ERR_INVALID_ARG, "sh_degree must be 0..4");
    if (cam->model > BH_CAMERA_THIN_PRISM_FISHEYE) return set_error(ctx, BH_ERR_INVALID_ARG, "unknown camera model");
    // (a splat's candidate box is walked with a 24-bit index: tile grids up to 4095 x 4095)
    if (cam->img_w > 65520 || cam->img_h > 65520) return set_error(ctx, BH_ERR_UNSUPPORTED, "images larger than 65520 px per side are not supported (tile grid <= 4095 x 4095)");
    if (n > 0 && (!transforms || !sh_coeffs || !raw_opacities)) return set_error(ctx, BH_ERR_INVALID_ARG, "render_forward: null splat tensor");
    if ((flags & BH_FLAG_SMOOTH_CUTOFF) && !(flags & BH_FLAG_BWD_INFO)) return set_error(ctx, BH_ERR_INVALID_ARG, "smooth cutoff requires the backward pass flag");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->far_job.pending) BH_TRY(finish_far_slice(ctx, nullptr));   // (a deferred decision nobody collected)
    return forward_impl(ctx, cam, n, sh_degree, transforms, sh_coeffs, raw_opacities, background, flags, out, /*allow_cut=*/true);
}

}  // extern "C"

// The forward pipeline (arguments validated by bh_render_forward).  allow_cut = false: complete lists whatever the view's table says
// (the second attempt after a failed forecast, finish_far_slice).
int bh::forward_impl(bh_ctx* ctx, const BhCamera* cam, uint32_t n, uint32_t sh_degree, const float* transforms, const float* sh_coeffs,
                            const float* raw_opacities, const float* background, uint32_t flags, BhRenderOut* out, bool allow_cut) {
    ctx->have_forward = false;
    ctx->clears.begin_forward();   // (filled in below only by the kernels of THIS forward)
    const bool mip = flags & BH_FLAG_MIP, bwd_info = flags & BH_FLAG_BWD_INFO, smooth = flags & BH_FLAG_SMOOTH_CUTOFF;
    const ViewUniforms u = make_uniforms(*cam);
    if (u.tile_y0 >= u.tile_y1 || u.tile_y1 > u.tile_bh) return set_error(ctx, BH_ERR_INVALID_ARG, "tile_row window must satisfy begin < end <= ceil(img_h / 16)");
    const uint32_t num_tiles = u.tile_bw * u.tile_bh;
    const size_t npad = n ? n : 1;

    // two counter pairs: K1 accumulates into one and clears the other for the next forward (no fill launch)
    constexpr size_t counter_set_bytes = COUNTER_SET_BYTES, counter_set_words = COUNTER_SET_BYTES / 4, counter_read_bytes = COUNTER_READ_BYTES;
    auto* counter_pairs = (uint32_t*)ensure(ctx, SLOT_COUNTERS, 2 * counter_set_bytes + 64);   // (+ the device copy of this frame's count sums)
    uint32_t* dev_sums = counter_pairs ? counter_pairs + 2 * counter_set_words : nullptr;
    uint32_t* counters = counter_pairs ? counter_pairs + counter_set_words * (ctx->counter_phase & 1u) : nullptr;
    auto* depth_keys = (uint32_t*)ensure(ctx, SLOT_DEPTH_KEYS, npad * 4);
    auto* isect_counts = (uint32_t*)ensure(ctx, SLOT_ISECT_COUNTS, npad * 4);
    auto* max_radius = ctx->ext_max_radius ? ctx->ext_max_radius : (float*)ensure(ctx, SLOT_MAX_RADIUS, npad * 4);
    auto* proj_by_gid = (float*)ensure(ctx, SLOT_PROJECTED_BY_GID, npad * 9 * 4);
    if (!counters || !depth_keys || !isect_counts || !max_radius || !proj_by_gid) return BH_ERR_OOM;

    auto* gfc = (uint32_t*)ensure(ctx, SLOT_GLOBAL_FROM_COMPACT, npad * 4);
    auto* depths_sorted = (uint32_t*)ensure(ctx, SLOT_DEPTHS_SORTED, npad * 4);
    if (!gfc || !depths_sorted) return BH_ERR_OOM;
    // [T,2] offsets | 8 x LPT_CLASSES work-class counters | [8][LPT_CLASSES][ceil(T/8)] class lists (longest-first tile order of the backward)
    // backward jobs (rasterize.hip): the blend backward works on checkpointed segments of the tiles' lists
    const bool bwd_jobs = bwd_info && ctx->knob_bwd_jobs && !ctx->knob_no_lpt && !((flags & BH_FLAG_SLICED_LISTS) && ctx->slice_fraction > 0.0f);
    const size_t lpt_words = LPT_HEADER_WORDS + (size_t)8 * LPT_CLASSES * band_slots(num_tiles);
    auto* tile_offsets = (uint32_t*)ensure(ctx, SLOT_TILE_OFFSETS, ((size_t)num_tiles * 2 + lpt_words) * 4);
    auto* visible = (bwd_info && ctx->ext_visible) ? ctx->ext_visible : (float*)ensure(ctx, SLOT_VISIBLE, (bwd_info ? npad : 1) * 4);
    if (!tile_offsets || !visible) return BH_ERR_OOM;
    const size_t visible_words = bwd_info ? ((ctx->ext_visible && ctx->ext_visible_floats) ? ctx->ext_visible_floats : npad) : 0;
    // depth-sliced lists: [0] near-slice splats  [1] near-slice pairs  [2] tiles the near slice left unsaturated  [3] far-slice pairs |
    // done bits | far tile offsets [T,2] | far pairs per block group.  Cleared by K1 with the tile table, whether or not this frame ends up slicing.
    const bool want_sliced = (flags & BH_FLAG_SLICED_LISTS) != 0;
    const size_t slice_bit_words = ((size_t)num_tiles + 31) / 32;
    const size_t slice_group_words = (size_t)n / (256 * FAR_GROUP_BLOCKS) + 2;   // far pairs per group of count-kernel blocks
    const size_t slice_words = want_sliced ? SLICE_CTRL_WORDS + slice_bit_words + (size_t)num_tiles * 2 + slice_group_words : 0;
    uint32_t* slice_tab = nullptr;
    if (want_sliced) {
        slice_tab = (uint32_t*)ensure(ctx, SLOT_SLICE, slice_words * 4);
        if (!slice_tab) return BH_ERR_OOM;
    }
    // where THIS forward's blend kernel leaves its slicing hint: the feedback words of the counter set the NEXT forward reads back
    uint32_t* feedback_next = counter_pairs ? counter_pairs + counter_set_words * ((ctx->counter_phase & 1u) ^ 1u) + COUNTER_FB_WORD : nullptr;
    // ---- per-tile depth cuts (BH_FLAG_SLICED_LISTS with the automatic share) ------------------------------------------------
    // A training loop comes back to each of its views every V steps with parameters that moved by a learning rate: how deep every
    // TILE of the view had to go last time is a near-exact forecast of how deep it has to go now.  So every blend launch leaves,
    // per tile, the depth key of the last splat the tile needed + a margin (rasterize.hip) in a table that belongs to the VIEW
    // (bh_set_view_id / BhTrainBatch.view_id), and the view's next frame lists a (splat, tile) pair only if the splat is at or in
    // front of the tile's cut: K1 counts those hits beside the exact ones (same walk), the scan runs over the near counts, K5 /
    // tile sort / offsets handle ~a tenth of the pairs — per tile, so a frame with thin or empty regions (whose tiles keep
    // "everything", i.e. their own short lists) is cut as deep as a frame that saturates everywhere.  Correctness does not rest
    // on the forecast: a tile that is still live behind an incomplete list is counted, and a frame with such a tile is rendered
    // AGAIN with complete lists (finish_far_slice: only the splats in front of the cuts were depth-ordered, so there is no far
    // pass to continue with); that second attempt re-seeds the table.
    ViewState* view = nullptr;
    bool cut_active = false;
    const bool auto_cuts = want_sliced && !(ctx->slice_fraction > 0.0f);
    // A frame with COMPLETE lists (no BH_FLAG_SLICED_LISTS: the reference's exact aux tensors, eval renders, exact_lists steps) takes
    // the view's table too — not to cut anything, but for the forward blend's tile order: its tiles start in descending order of the
    // work they had at the same camera's last frame (K16 193 -> ~150 us at 1 M splats / 1080p), and it refreshes the table.
    const bool order_only = !want_sliced && ctx->knob_k16_order != 0u && n >= 8u * 256u;
    if (n > 0 && (auto_cuts || order_only)) {
        const uint64_t vkey = view_key(ctx, *cam);
        // (a forward-only frame without a view id — a viewer's moving camera, an eval render — must not mint a table per frame)
        const bool casual = order_only && !bwd_info && (vkey >> 63) != 0ull;
        view = view_state(ctx, vkey, u.tile_bw, u.tile_bh, /*touch=*/allow_cut, casual);
        if (!view && auto_cuts) return set_error(ctx, BH_ERR_OOM, "hipMalloc for the per-view tile table failed");
        if (!view) (void)hipGetLastError();   // (ordering is optional: carry on in index order)
        // (a frame with few pairs has nothing to save: the near count in K1 and an occasional second attempt cost more than listing and
        //  sorting them all — 100 k splats at 512 x 512 trained 4 % slower with cuts; the view's last frame tells)
        if (!auto_cuts || !view) {
            // (complete lists by request)
        } else if (!allow_cut) {
            // (the forecast has just failed: this attempt re-seeds the table)
        } else if (view->seeded && view->exact_frames == 0u && view->last_pairs >= ctx->cut_min_pairs) {
            // (a view whose last cut frame listed nearly everything — a scene whose tiles no longer saturate early: a converging
            //  training run ends up there, bench.py train_loop — gains nothing from its cuts and pays for them: the near count in K1,
            //  and a whole second frame whenever a forecast fails.  Such a view renders complete lists, and tries a cut again later)
            if (view->complete_frames) view->complete_frames--;
            else cut_active = true;
        } else if (view->exact_frames) view->exact_frames--;
    }
    uint32_t* near_counts = nullptr;
    uint32_t* tile_order = nullptr;
    uint32_t* tile_split = nullptr;
    if (cut_active) {
        near_counts = (uint32_t*)ensure(ctx, SLOT_NEAR_COUNTS, npad * 4);
        if (!near_counts) return BH_ERR_OOM;
    }

    uint32_t nv = 0, ni = 0, near_total = 0, nv_true = 0;
    uint32_t fb_need = 0;                  // previous forward: most exact-list slots any saturated tile needed
    unsigned long long fb_unsat_pairs = 0; // ... and pairs it listed for tiles that never saturated
    uint32_t fb_unsat_tiles = 0;           // ... and how many such tiles there were (empty ones included)
    bool fused_scan = false;
    uint32_t* cum_early = nullptr;
    bool k5_queued = false;   // K5 was queued before the counts were read (below): valid unless the pairs overflowed its buffers
    uint32_t spec_pair_cap = 0;
    float* spec_projected = nullptr;
    float4* spec_vc = nullptr;
    uint32_t *spec_tile_ids = nullptr, *spec_isect_gids = nullptr;
    if (n > 0) {
        {
            ProfScope ps(ctx, "ProjectSplats");
            if (!ctx->counters_ready) BH_HIP(ctx, hipMemsetAsync(counter_pairs, 0, 2 * counter_set_bytes, ctx->stream));
            ctx->counters_ready = false;
            ForwardPrep prep;
            prep.next_counters = reinterpret_cast<unsigned long long*>(counter_pairs + counter_set_words * ((ctx->counter_phase & 1u) ^ 1u));
            prep.visible = visible_words ? reinterpret_cast<uint32_t*>(visible) : nullptr;
            prep.visible_words = (uint32_t)visible_words;
            prep.tile_table = tile_offsets;
            prep.tile_words = num_tiles * 2 + LPT_HEADER_WORDS;
            prep.slice_table = slice_tab;
            prep.slice_words = (uint32_t)slice_words;
            prep.list_all_visible = ctx->knob_cut_sort_all;
            if (view && ctx->knob_k16_order && n >= 8u * 256u) {   // the forward blend's tile order from the view's last per-tile work (K1's blocks 0..7 sort it: the grid must have them)
                const uint32_t win_t = u.tile_bw * (u.tile_y1 - u.tile_y0);
                tile_order = (uint32_t*)ensure(ctx, SLOT_TILE_ORDER, ((size_t)8 * band_slots(win_t) + SPLIT_TAIL_WORDS) * 4);
                if (!tile_order) return BH_ERR_OOM;
                prep.order_work = view->zcut + (size_t)num_tiles;
                prep.order_out = tile_order;
                prep.order_tiles = win_t;
                prep.order_tile_begin = u.tile_bw * u.tile_y0;
                prep.order_mode = ctx->knob_k16_order;
                prep.band_mode = ctx->knob_band_mode;
                if (ctx->knob_k16_split && ctx->knob_k16_order == 1u) {   // split tiles (context.h SPLIT_MAX): K1's order blocks pick them
                    tile_split = tile_order + (size_t)8 * band_slots(win_t);
                    prep.split_out = tile_split;
                    prep.split_factor = (float)ctx->knob_k16_split * 0.01f;
                    prep.split_min = ctx->knob_k16_split_min;
                    prep.split_of_max = (float)ctx->knob_k16_split_of_max * 0.01f;
                }
            }
            if (bwd_info && ctx->ext_grad_begin && ctx->ext_grad_floats && (ctx->ext_grad_floats & 3u) == 0 &&
                (reinterpret_cast<uintptr_t>(ctx->ext_grad_begin) & 15u) == 0 && ctx->ext_grad_floats / 4 <= 0xFFFFFFFFull) {
                prep.span = reinterpret_cast<float4*>(ctx->ext_grad_begin);   // the train step's gradient span
                prep.span_f4 = (uint32_t)(ctx->ext_grad_floats / 4);
                ctx->clears.k1_cleared_span();
            }
            BH_TRY(launch_project_forward(ctx, u, n, mip, sh_degree, transforms, sh_coeffs, raw_opacities, depth_keys, isect_counts, max_radius,
                                          proj_by_gid, counters, prep, cut_active ? view->zcut : nullptr, near_counts));
            ctx->counter_phase ^= 1u;
            ctx->counters_ready = true;
        }
        // the one mid-pipeline readback (render.rs:146-168).  The depth sort covers all n splats
        // and needs neither count, so it is queued BEFORE the host waits: the GPU sorts while the
        // host reads the counts, sizes the buffers and queues the rest.
        auto* hslots = reinterpret_cast<unsigned long long*>(ctx->host_counters + 16);
        fused_scan = depth_sort_supported(n) && !ctx->knob_generic_depth_sort;
        // (the sort's first kernel adds the counter slots up and stores the sums into the pinned block itself; only the generic
        //  sort path still needs a copy launch between K1 and the sort)
        const bool sums_on_device = fused_scan && !ctx->knob_readback_copy;
        // ... and the host does not wait for an event behind that kernel either: it polls a tag word the kernel stores behind the sums
        const bool poll_tag = sums_on_device && !ctx->knob_event_waits;
        if (!sums_on_device) {
            BH_HIP(ctx, hipMemcpyAsync(hslots, counters, counter_read_bytes, hipMemcpyDeviceToHost, ctx->stream));
            BH_HIP(ctx, hipEventRecord(ctx->readback_ev, ctx->stream));
        }
        {
            ProfScope ps(ctx, "DepthSort");
            // culled splats carry key 0xFFFFFFFF and sort behind every visible one:
            // the stable sort is also the (deterministic) compaction.
            if (fused_scan) {
                // ... and the scan of the tile counts in depth order rides on its last kernel (depth_sort.hip); the arena's
                // cum slot is sized for n here because the visible count is not known yet
                cum_early = (uint32_t*)ensure(ctx, SLOT_CUM_TILES_HIT, npad * 4);
                if (!cum_early) return BH_ERR_OOM;
                // (per-tile cuts: the scan of the NEAR counts = the slot ranges of the near pass's list)
                if (poll_tag && ++ctx->readback_tag == 0u) ctx->readback_tag = 1u;
                BH_TRY(depth_sort_scan(ctx, depth_keys, counters + COUNTER_MINMAX_WORD, cut_active ? near_counts : isect_counts, n, depths_sorted, gfc, cum_early,
                                       sums_on_device ? counters : nullptr, ctx->host_counters + 16, ctx->readback_ev, poll_tag ? ctx->readback_tag : 0u,
                                       sums_on_device ? dev_sums : nullptr,
                                       view ? view->zcut + 2 * (size_t)num_tiles + (cut_active ? DSORT_SPL_STRIDE : 0u) : nullptr,
                                       view ? &view->spl_written[cut_active ? 1 : 0] : nullptr));
            } else {
                BH_TRY(radix_argsort(ctx, depth_keys, nullptr, n, 32, depths_sorted, gfc));
            }
        }
        // ---- speculative K5: the list builder is queued BEFORE the host has read the counts.  Between the depth sort's last kernel and
        // K5 the GPU used to idle ~5.5 us every frame — the host learns the counts ~70 us into the frame, but sizing and launching K5
        // only then made K5's launch latency a bubble (the one idle gap a kernel trace of a step shows, scripts/gap_trace.py).  K5
        // needs two numbers the host does not have yet: the listed splats (it reads them on the device, where the sort's first kernel
        // left the sums) and room for the pairs — the pair buffers' CAPACITY is the bound: the arena only grows, so after a view's
        // first frames it holds what the frame needs; a wave whose slots would not fit emits nothing, the host finds the overflow in
        // the counts it reads anyway and queues K5 again behind the grown buffers.
        if (poll_tag && ctx->knob_spec_k5 && !(want_sliced && ctx->slice_fraction > 0.0f)) {
            const size_t cap_pairs = std::min(ctx->slots[SLOT_TILE_IDS].cap, ctx->slots[SLOT_ISECT_GIDS].cap) / 4;
            if (cap_pairs >= 4096) {
                spec_projected = (float*)ensure(ctx, SLOT_PROJECTED, npad * 9 * 4);   // (rows for up to n listed splats)
                spec_vc = bwd_info ? (float4*)ensure(ctx, SLOT_V_COMBINED, npad * 10 * 4 + 16) : nullptr;
                if (!spec_projected || (bwd_info && !spec_vc)) return BH_ERR_OOM;
                spec_tile_ids = (uint32_t*)ctx->slots[SLOT_TILE_IDS].ptr;
                spec_isect_gids = (uint32_t*)ctx->slots[SLOT_ISECT_GIDS].ptr;
                spec_pair_cap = (uint32_t)std::min<size_t>(cap_pairs, 0xFFFFFFFEull);
                ProfScope ps(ctx, "MapGaussiansToIntersect");
                BH_TRY(launch_map_gaussians(ctx, n, u, proj_by_gid, gfc, spec_projected, cum_early, spec_tile_ids, spec_isect_gids, spec_vc,
                                            spec_vc ? (uint32_t)(((size_t)n * 10 + 3) / 4) : 0u, 0xFFFFFFFFu, cut_active ? slice_tab : nullptr,
                                            cut_active ? view->zcut : nullptr, cut_active ? depths_sorted : nullptr, dev_sums + (cut_active ? 3 : 0), spec_pair_cap));
                k5_queued = true;
            }
        }
        if (poll_tag) BH_TRY(wait_host_tag(ctx, reinterpret_cast<const volatile uint32_t*>(hslots) + HOST_SUM_WORDS - 1u, ctx->readback_tag, "count readback"));
        else BH_HIP(ctx, hipEventSynchronize(ctx->readback_ev));
        if (ctx->gate_learn) {   // the previous sliced frame queued its far slice unasked: did it need it?
            ctx->gate_learn = false;
            ctx->far_direct = reinterpret_cast<const volatile uint32_t*>(ctx->host_counters)[HOST_GATE_WORD] != 0u;
        }
        unsigned long long hc[4] = {0ull, 0ull, 0ull, 0ull};
        if (sums_on_device) {
            const volatile unsigned long long* hs = hslots;
            for (uint32_t c = 0; c < COUNTER_K1_U64; ++c) hc[c] = hs[c];
            const volatile uint32_t* hfb = reinterpret_cast<const volatile uint32_t*>(hslots) + 2 * COUNTER_K1_U64;
            fb_need = hfb[0]; fb_unsat_pairs = hfb[1]; fb_unsat_tiles = hfb[2];   // the previous forward's slicing hint
        } else {
            for (uint32_t k = 0; k < COUNTER_SLOTS; ++k)
                for (uint32_t c = 0; c < COUNTER_K1_U64; ++c) hc[c] += hslots[COUNTER_K1_U64 * k + c];
            // the previous forward's slicing hint came along in the same copy
            const uint32_t* hfb = reinterpret_cast<const uint32_t*>(hslots) + COUNTER_FB_WORD;
            for (uint32_t k = 0; k < COUNTER_SLOTS; ++k) { if (hfb[3 * k] > fb_need) fb_need = hfb[3 * k]; fb_unsat_pairs += hfb[3 * k + 1]; fb_unsat_tiles += hfb[3 * k + 2]; }
        }
        if (hc[1] > 0xFFFFFFFFull) return set_error(ctx, BH_ERR_UNSUPPORTED, "more than 2^32-1 tile intersections");
        nv_true = (uint32_t)hc[0];
        ni = (uint32_t)hc[1];
        near_total = cut_active ? (uint32_t)hc[2] : ni;
        // per-tile cuts: only the splats that own a pair in front of some cut were given a real depth key (K1): the compact
        // arrays, K5's grid, the backward's accumulator and K18 are sized for THEM; num_visible stays the reference's count
        nv = cut_active ? (uint32_t)hc[3] : nv_true;   // (BH_CUT_SORT_ALL: K1 then listed every visible splat, hc[3] == hc[0])
    } else {   // no K1 to clear them on the way
        BH_HIP(ctx, hipMemsetAsync(tile_offsets, 0, ((size_t)num_tiles * 2 + LPT_HEADER_WORDS) * 4, ctx->stream));
        if (visible_words) BH_HIP(ctx, hipMemsetAsync(visible, 0, visible_words * 4, ctx->stream));
        if (counter_pairs) {   // counter_phase does not flip without K1: clear what this frame's blend kernel will add to
            BH_HIP(ctx, hipMemsetAsync(feedback_next, 0, COUNTER_SLOTS * 12, ctx->stream));
        }
    }

    const size_t nvpad = nv ? nv : 1, nipad = ni ? ni : 1;
    auto* cum = fused_scan ? cum_early : (uint32_t*)ensure(ctx, SLOT_CUM_TILES_HIT, nvpad * 4);
    auto* projected = (float*)ensure(ctx, SLOT_PROJECTED, nvpad * 9 * 4);
    auto* tile_ids = (uint32_t*)ensure(ctx, SLOT_TILE_IDS, nipad * 4);
    auto* isect_gids = (uint32_t*)ensure(ctx, SLOT_ISECT_GIDS, nipad * 4);
    auto* tile_ids_sorted = (uint32_t*)ensure(ctx, SLOT_TILE_IDS_SORTED, nipad * 4);
    auto* isect_gids_sorted = (uint32_t*)ensure(ctx, SLOT_ISECT_GIDS_SORTED, nipad * 4);
    const size_t pixels = (size_t)u.img_w * u.img_h;
    void* out_img = ensure(ctx, SLOT_OUT_IMG, pixels * (bwd_info ? 16 : 4));
    if (!cum || !projected || !tile_ids || !isect_gids || !tile_ids_sorted || !isect_gids_sorted || !out_img)
        return BH_ERR_OOM;
    // the speculative K5 stands if its pairs fitted and none of its buffers has moved since
    if (k5_queued && ((cut_active ? near_total : ni) > spec_pair_cap || projected != spec_projected || tile_ids != spec_tile_ids || isect_gids != spec_isect_gids))
        k5_queued = false;

    // ---- depth-sliced lists (BH_FLAG_SLICED_LISTS) --------------------------------------------------------------------------
    // The reference lists EVERY (tile, splat) pair and sorts them all (map_gaussians.rs:15-80, render.rs:228-230), although a tile
    // stops blending once its 256 pixels are saturated (rasterize.rs:116-189): at 1 M splats / 1080p the blend reads 9 % of the
    // sorted list, at 6 M / 4K 1.7 %.  The splats are in depth order and cum_tiles_hit is their slot ranges, so "slot end <= budget"
    // is a NEAR slice of the depth order whose pairs are the first I0 slots of the exact list.  List + sort + blend that slice;
    // tiles whose pixels all saturated are final (one bit each); the FAR rest is listed only into tiles that are not
    // (count -> emit against the bit table), sorted behind the near list and blended from the parked pixel state.
    // Everything the far slice launches is a no-op when no tile is left (a device-side gate: no host round trip).  Per pixel the
    // same splats are folded in the same order: out_img, visible[], the blended part of every list and the gradients are those
    // of the exact path.  The near slice's size comes from the PREVIOUS forward on this ctx (how many slots its slowest
    // saturating tile needed, + 25 %), or from bh_set_list_slicing; scenes that do not saturate keep the single exact list.
    uint32_t budget = ni;   // == ni: one slice, the exact lists
    bool sliced = false;
    if (want_sliced && ni > 0) {
        const float share = ctx->slice_fraction;
        if (share > 0.0f) {   // a fixed share of the pair list (bh_set_list_slicing: tests, A/B): the slot-budget slices
            if (share < 1.0f) {
                const double b = (double)share * (double)ni;
                const uint32_t floor_b = ni < 1024u ? ni : 1024u;
                budget = b < (double)floor_b ? floor_b : (uint32_t)b;
                if (budget > ni) budget = ni;
            }
            sliced = budget < ni;
            ctx->last_slice_share = (float)((double)budget / (double)ni);
        } else if (cut_active && near_total < ni) {   // per-tile depth cuts from this view's last frame
            sliced = true;
            budget = near_total;
            ctx->last_slice_share = (float)((double)near_total / (double)ni);
            view->last_share = ctx->last_slice_share;
            if (ctx->auto_exact_share > 0.0f && view->last_share > ctx->auto_exact_share) view->complete_frames = AUTO_EXACT_FRAMES;
        } else {
            if (cut_active) {   // (the cut removed nothing at all)
                view->last_share = 1.0f;
                if (ctx->auto_exact_share > 0.0f) view->complete_frames = AUTO_EXACT_FRAMES;
            }
            // no history for this view yet (or its forecast keeps failing, or it cut nothing): complete lists; the blend
            // kernel seeds / refreshes the view's table
            ctx->last_slice_share = 1.0f;
        }
    }
    (void)fb_need; (void)fb_unsat_pairs; (void)fb_unsat_tiles;
    const bool by_cut = cut_active && sliced;           // this frame's lists end at the per-tile cuts
    const uint32_t* zcut_lists = cut_active ? view->zcut : nullptr;   // (cut_active but not sliced: the cut removed nothing — K5 still filters, and keeps everything)
    uint32_t* slice_info = slice_tab;
    uint32_t* done_bits = slice_tab ? slice_tab + SLICE_CTRL_WORDS : nullptr;
    uint32_t* tile_offsets_far = slice_tab ? slice_tab + SLICE_CTRL_WORDS + slice_bit_words : nullptr;
    uint32_t* far_group_totals = slice_tab ? slice_tab + SLICE_CTRL_WORDS + slice_bit_words + (size_t)num_tiles * 2 : nullptr;
    uint32_t tile_bits = 0;
    while (tile_bits < 32 && (num_tiles >> tile_bits) != 0) tile_bits++;  // render.rs:228
    RasterSlice rs;
    rs.cum = cum;
    ctx->jobs = BwdJobs{};
    if (bwd_jobs) {
        // checkpoint slots are addressed by list position (context.h BwdJobs): listed pairs / BWD_SEG + tiles of them
        const uint32_t listed = cut_active ? near_total : ni;
        const uint32_t ckpt_cap = (uint32_t)std::min<uint64_t>((uint64_t)listed / BWD_SEG + num_tiles + 1u, BWD_CKPT_MAX_SLOTS);
        rs.jobs.ckpt = (float4*)ensure(ctx, SLOT_BWD_CKPT, (size_t)ckpt_cap * 256 * sizeof(float4));
        rs.jobs.ckpt_cap = ckpt_cap;
        rs.jobs.top_cap = band_slots(num_tiles) + ckpt_cap;   // (a band's full segments: at most one per checkpoint + one per tile)
        rs.jobs.top_list = (uint32_t*)ensure(ctx, SLOT_BWD_TOPLIST, (size_t)8 * rs.jobs.top_cap * 4);
        if (!rs.jobs.ckpt || !rs.jobs.top_list) return BH_ERR_OOM;
        ctx->jobs = rs.jobs;
    }
    rs.feedback = nullptr;   // (the slot-budget heuristics that read it are gone: the automatic mode cuts per tile)
    (void)feedback_next;
    if (view) {   // every forward of a view refreshes its table
        rs.zcut = view->zcut;
        rs.depth_keys_sorted = depths_sorted;
        rs.nv = nv;
        rs.cut_active = by_cut;
        rs.margin_pct = ctx->knob_fixed_margin ? ctx->knob_cut_margin_pct : cut_margin_pct(ctx, view);
        rs.work = view->zcut + (size_t)num_tiles;
        rs.order = tile_order;
        rs.order_mode = ctx->knob_k16_order;
        rs.split = tile_split;
#ifdef BH_TEST_HOOKS
        ctx->last_split = tile_split;
#endif
    }
    // work classes ~1/64 of the mean list length wide (a tile typically blends ~10 % of its list before it saturates)
    const uint32_t win_tiles = u.tile_bw * (u.tile_y1 - u.tile_y0);
    const float class_width_raw = (float)ni / (float)(win_tiles ? win_tiles : 1u) / 64.0f;
    const float class_width = ctx->knob_lpt_linear ? (class_width_raw < 8.0f ? 8.0f : class_width_raw) : 0.0f;   // (0: logarithmic classes, rasterize.hip)
    ctx->lpt = (bwd_info && !ctx->knob_no_lpt) ? tile_offsets + (size_t)num_tiles * 2 : nullptr;
    float* out_f32 = bwd_info ? (float*)out_img : nullptr;
    uint32_t* out_u8 = bwd_info ? nullptr : (uint32_t*)out_img;

    bool offsets_done = false;   // the tile sort wrote the offsets table on its way
    if (nv > 0) {
        if (!fused_scan) {
            ProfScope ps(ctx, "PrefixSumGaussHits");
            BH_TRY(prefix_sum(ctx, cut_active ? near_counts : isect_counts, gfc, nv, cum, false));
        }
        if (ni == 0) {   // nothing to map: only the record gather is left (K5 does it on its way otherwise)
            ProfScope ps(ctx, "ProjectVisible");
            BH_TRY(launch_project_visible(ctx, nv, proj_by_gid, gfc, projected));
        }
        if (ni > 0) {
            {
                ProfScope ps(ctx, "MapGaussiansToIntersect");
                // the backward's accumulator [nv,10] is cleared by K5 on its way (whole float4s: + 16 B of room)
                float4* vc = nullptr;
                if (bwd_info) {
                    vc = (float4*)ensure(ctx, SLOT_V_COMBINED, (size_t)nv * 10 * 4 + 16);
                    if (!vc) return BH_ERR_OOM;
                }
                if (zcut_lists && near_total == 0u) {
                    // no pair in front of any cut: nothing to emit (and K5 does not run to clear v_combined)
                } else if (k5_queued && (!bwd_info || vc == spec_vc)) {
                    ctx->clears.k5_cleared_accum(vc != nullptr);   // (it ran in front of the count readback)
                } else {
                    ctx->clears.k5_cleared_accum(vc != nullptr);
                    BH_TRY(launch_map_gaussians(ctx, nv, u, proj_by_gid, gfc, projected, cum, tile_ids, isect_gids, vc, vc ? (uint32_t)(((size_t)nv * 10 + 3) / 4) : 0u,
                                                (sliced && !by_cut) ? budget : 0xFFFFFFFFu, (sliced || zcut_lists) ? slice_info : nullptr, zcut_lists,
                                                zcut_lists ? depths_sorted : nullptr));
                }
            }
            {
                ProfScope ps(ctx, "TileSort");
                // (scratch sized for the whole list: the near list's length moves from frame to frame)
                // host-known lengths: the sort that also writes the offsets table, four launches instead of seven (sort.hip)
                if ((zcut_lists || !sliced) && tile_sort_supported(tile_bits, zcut_lists ? near_total : ni) && !(ctx->knob_tile_sort == 2u)) {
                    BH_TRY(tile_sort_offsets(ctx, tile_ids, isect_gids, zcut_lists ? near_total : ni, tile_bits, num_tiles, tile_ids_sorted, isect_gids_sorted,
                                             tile_offsets, ni));
                    offsets_done = true;
                } else
                if (zcut_lists) BH_TRY(radix_argsort_dev(ctx, tile_ids, isect_gids, near_total, nullptr, nullptr, nullptr, tile_bits, tile_ids_sorted, isect_gids_sorted, ni));
                else if (sliced) BH_TRY(radix_argsort_dev(ctx, tile_ids, isect_gids, budget, slice_info + 1, nullptr, nullptr, tile_bits, tile_ids_sorted, isect_gids_sorted, ni));
                else BH_TRY(radix_argsort(ctx, tile_ids, isect_gids, ni, tile_bits, tile_ids_sorted, isect_gids_sorted));
            }
        }
    }
    if (!offsets_done) {
        ProfScope ps(ctx, "GetTileOffsets");   // K1 cleared the table (or a fill did, for n == 0)
        if (zcut_lists) BH_TRY(launch_tile_offsets(ctx, tile_ids_sorted, near_total, num_tiles, tile_offsets, /*pre_zeroed=*/true));
        else if (sliced) BH_TRY(launch_tile_offsets_dev(ctx, tile_ids_sorted, budget, slice_info + 1, nullptr, nullptr, num_tiles, tile_offsets));
        else BH_TRY(launch_tile_offsets(ctx, tile_ids_sorted, ni, num_tiles, tile_offsets, /*pre_zeroed=*/true));
    }
    if (!sliced) {
        ProfScope ps(ctx, "Rasterize");   // `visible` was cleared by K1 as well
        BH_TRY(launch_rasterize(ctx, u, background, bwd_info, smooth, isect_gids_sorted, tile_offsets, projected, gfc, out_f32, out_u8, visible, ctx->lpt,
                                class_width, /*phase=*/0, &rs));
    } else {
        auto* state = (float*)ensure(ctx, SLOT_SLICE_STATE, pixels * 16);
        auto* far_counts = (uint32_t*)ensure(ctx, SLOT_SLICE_COUNTS, nvpad * 4);
        auto* far_block_totals = (uint32_t*)ensure(ctx, SLOT_SLICE_CUM, (nvpad / 256 + 2) * 4);
        if (!state || !far_counts || !far_block_totals) return BH_ERR_OOM;
        rs.done_bits = done_bits;
        rs.unsat_count = slice_info + 2;
        if (!ctx->knob_readback_copy) {
            // how many tiles are left, for the host (to decide now, or to learn for the next frame: context.h far_direct): the
            // blend kernel stores into the pinned word itself.  Nothing of an earlier frame can still write it — every frame's
            // count readback waited behind the previous frame's blend.
            rs.gate_host = ctx->host_counters + HOST_GATE_WORD;
            *reinterpret_cast<volatile uint32_t*>(rs.gate_host) = 0u;
        }
        rs.state = state;
        rs.offsets_near = tile_offsets;
        rs.live_bands = slice_info + 4;
        {
            ProfScope ps(ctx, "Rasterize");
            BH_TRY(launch_rasterize(ctx, u, background, bwd_info, smooth, isect_gids_sorted, tile_offsets, projected, gfc, out_f32, out_u8, visible, ctx->lpt,
                                    class_width, /*phase=*/1, &rs));
        }
        FarJob& j = ctx->far_job;
        j.u = u;
        j.bg[0] = background[0]; j.bg[1] = background[1]; j.bg[2] = background[2];
        j.bwd_info = bwd_info; j.smooth = smooth;
        j.nv = nv; j.ni = ni; j.budget = budget; j.num_tiles = num_tiles; j.tile_bits = tile_bits;
        j.proj_by_gid = proj_by_gid; j.gfc = gfc; j.projected = projected; j.cum = cum;
        j.slice_info = slice_info; j.done_bits = done_bits; j.tile_offsets_far = tile_offsets_far;
        j.far_counts = far_counts; j.far_block_totals = far_block_totals; j.far_group_totals = far_group_totals;
        j.tile_ids = tile_ids; j.isect_gids = isect_gids; j.tile_ids_sorted = tile_ids_sorted; j.isect_gids_sorted = isect_gids_sorted;
        j.out_f32 = out_f32; j.out_u8 = out_u8; j.visible = visible; j.lpt = ctx->lpt; j.class_width = class_width; j.rs = rs;
        j.by_cut = by_cut;
        j.view = by_cut ? view : nullptr;
        j.view_shared = ctx->view_id == 0u && ctx->knob_no_view_hash;   // (one table shared by every frame without an id: the A/B knob only)
        if (by_cut) {   // what a second attempt with complete lists needs (finish_far_slice)
            j.cam = *cam;
            j.n = n; j.sh_degree = sh_degree; j.flags = flags; j.view_id = ctx->view_id;
            j.transforms = transforms; j.sh_coeffs = sh_coeffs; j.raw_opacities = raw_opacities;
            j.ext_visible = ctx->ext_visible; j.ext_max_radius = ctx->ext_max_radius; j.ext_visible_floats = ctx->ext_visible_floats;
            j.ext_grad_begin = ctx->ext_grad_begin; j.ext_grad_floats = ctx->ext_grad_floats;
        }
        if (ctx->knob_readback_copy)
            BH_HIP(ctx, hipMemcpyAsync(ctx->host_counters + HOST_GATE_WORD, slice_info + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
        // (per-tile cuts: the forecast is expected to hold — the host decides every time, bh_train_step hides the wait behind its
        //  loss kernels)
        if (ctx->far_direct && !by_cut) {
            BH_TRY(enqueue_far_slice(ctx, j));
            ctx->gate_learn = true;
        } else {
            // bh_train_step (defer_far) queues its loss kernels next and lets the first of them signal "the blend is done" through a
            // tag word: no event behind the blend (finish_far_slice records one late if nobody queued the signal)
            if (++ctx->gate_tag == 0u) ctx->gate_tag = 1u;
            j.gate_tag = ctx->gate_tag;
            ctx->gate_signal_queued = false;
            j.gate_event_recorded = !(ctx->defer_far && !ctx->knob_event_waits && !ctx->knob_readback_copy);
            if (j.gate_event_recorded) BH_HIP(ctx, hipEventRecord(ctx->gate_ev, ctx->stream));
            j.pending = true;
            if (!ctx->defer_far) {
                bool again = false;
                BH_TRY(finish_far_slice(ctx, &again));
                if (by_cut && again) {   // the forecast failed and the frame was rendered a second time, with complete lists: that is the result
                    *out = ctx->last;
                    return 0;
                }
            }
        }
    }

    BhRenderOut r{};
    r.num_visible = nv_true;
    r.num_listed_splats = nv;
    r.num_intersections = ni;
    r.num_tiles = num_tiles;
    r.tile_bw = u.tile_bw;
    r.tile_bh = u.tile_bh;
    r.flags = flags;
    r.out_img = bwd_info ? (float*)out_img : nullptr;
    r.out_img_packed = bwd_info ? nullptr : (uint32_t*)out_img;
    r.visible = bwd_info ? visible : nullptr;
    r.max_radius = max_radius;
    r.tile_offsets = tile_offsets;
    r.projected = projected;
    r.compact_gid_from_isect = isect_gids_sorted;
    r.tile_id_from_isect = tile_ids_sorted;
    r.global_from_compact_gid = gfc;
    r.cum_tiles_hit = cum;
    r.intersect_counts = isect_counts;
    r.depths_sorted = (float*)depths_sorted;
    r.tile_offsets_far = sliced ? tile_offsets_far : nullptr;
    r.list_budget = budget;   // (per-tile cuts: the pairs the near pass listed)
    r.generation = ++ctx->generation;
    ctx->clears.stamp(r.generation);
    *out = r;
    ctx->last = r;
    ctx->last_listed_splats = nv;
    ctx->cam = *cam;
    ctx->uniforms = u;
    ctx->n = n;
    ctx->sh_degree = sh_degree;
    ctx->flags = flags;
    ctx->bg[0] = background[0]; ctx->bg[1] = background[1]; ctx->bg[2] = background[2];
    ctx->have_forward = true;
    ctx->had_forward = true;
    ctx->prev_intersections = ni;
    ctx->last_one_slice = !sliced;
    if (view) {   // this frame's blend kernel leaves what every tile needed
        view->seeded = true;
        view->last_pairs = ni;
    }
    return 0;
}

extern "C" {

int bh_set_list_slicing(bh_ctx* ctx, float near_share) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (near_share != near_share || near_share > 1.0f) return set_error(ctx, BH_ERR_INVALID_ARG, "set_list_slicing: the near slice's share must be <= 1 (<= 0: automatic)");
    ctx->slice_fraction = near_share > 0.0f ? near_share : 0.0f;
    return 0;
}

int bh_set_list_cut_threshold(bh_ctx* ctx, uint32_t min_pairs) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    ctx->cut_min_pairs = min_pairs;
    return 0;
}

int bh_set_view_id(bh_ctx* ctx, uint32_t view_id) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    ctx->view_id = view_id;
    return 0;
}

int bh_forget_views(bh_ctx* ctx) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->far_job.pending) BH_TRY(finish_far_slice(ctx, nullptr));
    // (the depth sort's splitter tables are per-ctx state of the same kind: a run that starts over starts from the linear split)
    if (ctx->dsort_spl) { BH_HIP(ctx, hipMemsetAsync(ctx->dsort_spl, 0, DSORT_SPL_STRIDE * 4, ctx->stream)); ctx->dsort_spl_written = false; }
    if (ctx->views.empty()) return 0;
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));   // queued kernels may still use the tables
    deliver_pending_loss(ctx);
    for (auto& kv : ctx->views)
        if (kv.second.zcut) (void)hipFree(kv.second.zcut);
    ctx->views.clear();
    ctx->casual_views = 0;
    for (uint64_t& k : ctx->seen_keys) k = 0ull;
    ctx->gate_view = nullptr;
    ctx->far_job.view = nullptr;
    ctx->margin_scale = 1.0f;
    return 0;
}

float bh_last_list_share(bh_ctx* ctx) { return ctx ? ctx->last_slice_share : 0.0f; }
uint32_t bh_far_slices_queued(bh_ctx* ctx) { return ctx ? ctx->far_launches : 0u; }
uint32_t bh_view_table_count(bh_ctx* ctx) { return ctx ? (uint32_t)ctx->views.size() : 0u; }

// ---- backward ------------------------------------------------------------------
}  // extern "C"

namespace bh {

static ForwardState latest_forward(const bh_ctx* ctx) {
    ForwardState fs;
    fs.out = ctx->last;
    fs.uniforms = ctx->uniforms;
    fs.n = ctx->n; fs.sh_degree = ctx->sh_degree; fs.flags = ctx->flags;
    fs.bg[0] = ctx->bg[0]; fs.bg[1] = ctx->bg[1]; fs.bg[2] = ctx->bg[2];
    fs.lpt = ctx->lpt;
    fs.jobs = ctx->jobs;
    return fs;
}

// The two backward kernels on the saved state `fs` (bwd/render_bwd.rs:21-171).  What the forward's kernels cleared on their way is
// recorded in ctx->clears under that forward's generation: a backward of any other forward (a retained, older one) finds nothing
// there and clears everything itself.
static int backward_impl(bh_ctx* ctx, const ForwardState& fs, const float* v_output, const float* transforms, const float* sh_coeffs,
                         const float* raw_opacities, float* v_transforms, float* v_sh_coeffs, float* v_raw_opacities, float* v_refine_weight) {
    const BhRenderOut& r = fs.out;
    const uint32_t n = fs.n, nv = r.num_listed_splats, C = (fs.sh_degree + 1) * (fs.sh_degree + 1);
    const size_t nvpad = nv ? nv : 1;
    auto* v_combined = (float*)ensure(ctx, SLOT_V_COMBINED, nvpad * 10 * 4 + 16);   // + room to clear whole float4s
    if (!v_combined) return BH_ERR_OOM;
    bool row_marks = false;
    {
        ProfScope ps(ctx, "ZeroGradBuffers");
        // what the forward's kernels cleared on their way (K5: v_combined; K1: the train step's gradient span) is done
        const bool one_span = n > 0 && ctx->ext_grad_begin == v_transforms && ctx->ext_grad_floats;
        // Consumed here, whichever forward this backward belongs to: the accumulator and the span are dirty from now on, and a
        // backward of ANOTHER forward (a retained one) finds nothing to trust (take_* check the generation).
        const bool vc_done = ctx->clears.take_accum(r.generation);
        const GradClears::Span span = ctx->clears.take_span(r.generation);
        row_marks = span == GradClears::ROW_MARKS;
        // (ROW_MARKS: the single-GPU train step reads only the rows K18 writes and marks — its forward cleared the marks)
        if (row_marks && !one_span) return set_error(ctx, BH_ERR_STATE, "internal: row-marked gradients without the train step's gradient span");
        const bool span_done = one_span && span != GradClears::NONE;
        if (one_span && (ctx->ext_grad_floats & 3u) == 0 && (reinterpret_cast<uintptr_t>(v_transforms) & 15u) == 0) {
            // v_combined and the exchange buffer's gradient span cleared by ONE launch (hipMemsetAsync spends two or
            // three launches on them, each ~5 us of latency beyond the bytes)
            const size_t na = vc_done ? 0 : (nvpad * 10 + 3) / 4, nb = span_done ? 0 : ctx->ext_grad_floats / 4;
            if (na + nb) {
                hipLaunchKernelGGL(zero_two_kernel, dim3(2048), dim3(256), 0, ctx->stream, reinterpret_cast<float4*>(v_combined), na,
                                   reinterpret_cast<float4*>(v_transforms), nb);
                BH_LAUNCH_CHECK(ctx, "zero_two_kernel");
            }
        } else {
            if (!vc_done) BH_HIP(ctx, hipMemsetAsync(v_combined, 0, nvpad * 10 * 4, ctx->stream));
            if (n > 0) {
                // dense outputs are zero-filled; the kernel scatters compact -> global (render_bwd.rs:123-138)
                if (one_span) {
                    BH_HIP(ctx, hipMemsetAsync(v_transforms, 0, ctx->ext_grad_floats * 4, ctx->stream));
                } else {
                    BH_HIP(ctx, hipMemsetAsync(v_transforms, 0, (size_t)n * 10 * 4, ctx->stream));
                    BH_HIP(ctx, hipMemsetAsync(v_sh_coeffs, 0, (size_t)n * C * 3 * 4, ctx->stream));
                    BH_HIP(ctx, hipMemsetAsync(v_raw_opacities, 0, (size_t)n * 4, ctx->stream));
                    BH_HIP(ctx, hipMemsetAsync(v_refine_weight, 0, (size_t)n * 4, ctx->stream));
                }
            }
        }
    }
    {
        ProfScope ps(ctx, "RasterizeBackwards", /*dominant=*/true);
        if (r.num_intersections > 0)
            BH_TRY(launch_rasterize_backward(ctx, fs.uniforms, fs.bg, fs.flags & BH_FLAG_SMOOTH_CUTOFF,
                                             r.compact_gid_from_isect, r.tile_offsets, r.projected, r.out_img, v_output, v_combined, fs.lpt,
                                             r.tile_offsets_far, /*want_refine=*/!ctx->bwd_skip_refine, &fs.jobs));
    }
    {
        ProfScope ps(ctx, "ProjectBackwards");
        BH_TRY(launch_project_backward(ctx, fs.uniforms, nv, fs.flags & BH_FLAG_MIP, fs.sh_degree, transforms, sh_coeffs,
                                       raw_opacities, r.global_from_compact_gid, v_combined, v_transforms, v_sh_coeffs,
                                       v_raw_opacities, v_refine_weight, row_marks, r.projected));
    }
    return 0;
}

// the arena slots a BhRenderOut points into (everything a retained forward must keep alive)
static const Slot kRetainSlots[RETAIN_SLOTS] = {SLOT_OUT_IMG, SLOT_VISIBLE, SLOT_MAX_RADIUS, SLOT_TILE_OFFSETS, SLOT_PROJECTED, SLOT_ISECT_GIDS_SORTED,
                                                SLOT_TILE_IDS_SORTED, SLOT_GLOBAL_FROM_COMPACT, SLOT_CUM_TILES_HIT, SLOT_ISECT_COUNTS, SLOT_DEPTHS_SORTED,
                                                SLOT_SLICE, SLOT_BWD_CKPT, SLOT_BWD_TOPLIST};

}  // namespace bh

extern "C" {

int bh_render_backward(bh_ctx* ctx, const float* v_output, const float* transforms, const float* sh_coeffs,
                       const float* raw_opacities, float* v_transforms, float* v_sh_coeffs, float* v_raw_opacities,
                       float* v_refine_weight) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!ctx->have_forward || !(ctx->flags & BH_FLAG_BWD_INFO))
        return set_error(ctx, BH_ERR_STATE, "render_backward needs a preceding BH_FLAG_BWD_INFO forward on this context");
    if (!v_output || !v_transforms || !v_sh_coeffs || !v_raw_opacities || !v_refine_weight)
        return set_error(ctx, BH_ERR_INVALID_ARG, "render_backward: null argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->far_job.pending) BH_TRY(finish_far_slice(ctx, nullptr));
    return backward_impl(ctx, latest_forward(ctx), v_output, transforms, sh_coeffs, raw_opacities, v_transforms, v_sh_coeffs, v_raw_opacities,
                         v_refine_weight);
}

// SplatBwdOps::{rasterize_bwd, project_bwd} take the forward's saved tensors explicitly (bwd/burn_glue.rs:62-92, 336-371): so does this.
int bh_render_backward_saved(bh_ctx* ctx, const BhRenderOut* saved, const float* v_output, const float* transforms, const float* sh_coeffs,
                             const float* raw_opacities, float* v_transforms, float* v_sh_coeffs, float* v_raw_opacities,
                             float* v_refine_weight) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!saved || !v_output || !v_transforms || !v_sh_coeffs || !v_raw_opacities || !v_refine_weight)
        return set_error(ctx, BH_ERR_INVALID_ARG, "render_backward_saved: null argument");
    if (!(saved->flags & BH_FLAG_BWD_INFO)) return set_error(ctx, BH_ERR_STATE, "render_backward_saved: the saved forward was not a BH_FLAG_BWD_INFO forward");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->far_job.pending) BH_TRY(finish_far_slice(ctx, nullptr));
    for (const Retained& rt : ctx->retained)
        if (rt.fs.out.generation == saved->generation && rt.fs.out.out_img == saved->out_img)
            return backward_impl(ctx, rt.fs, v_output, transforms, sh_coeffs, raw_opacities, v_transforms, v_sh_coeffs, v_raw_opacities,
                                 v_refine_weight);
    if (ctx->have_forward && saved->generation == ctx->last.generation && saved->out_img == ctx->last.out_img)
        return backward_impl(ctx, latest_forward(ctx), v_output, transforms, sh_coeffs, raw_opacities, v_transforms, v_sh_coeffs,
                             v_raw_opacities, v_refine_weight);
    char msg[256];
    snprintf(msg, sizeof msg, "render_backward_saved: forward #%llu is stale (the context's buffers now hold forward #%llu); call bh_render_retain "
                              "on a forward that must outlive the next one", (unsigned long long)saved->generation, (unsigned long long)ctx->generation);
    return set_error(ctx, BH_ERR_STATE, msg);
}

int bh_render_retain(bh_ctx* ctx, const BhRenderOut* out) {
    if (!ctx || !out) return BH_ERR_INVALID_ARG;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->far_job.pending) BH_TRY(finish_far_slice(ctx, nullptr));
    if (!ctx->have_forward || out->generation != ctx->last.generation || out->out_img != ctx->last.out_img || out->out_img_packed != ctx->last.out_img_packed)
        return set_error(ctx, BH_ERR_STATE, "render_retain: only the context's most recent forward can be retained (and only once)");
    Retained rt;
    rt.fs = latest_forward(ctx);
    for (int i = 0; i < RETAIN_SLOTS; ++i) {   // the blocks leave the arena: the next forward gets its own (from the pool, or hipMalloc)
        rt.blocks[i] = ctx->slots[kRetainSlots[i]];
        ctx->slots[kRetainSlots[i]] = Buffer{};
    }
    ctx->retained.push_back(rt);
    ctx->have_forward = false;    // bh_render_backward ("the last forward") has nothing to refer to until the next forward
    ctx->lpt = nullptr;
    return 0;
}

int bh_render_release(bh_ctx* ctx, const BhRenderOut* out) {
    if (!ctx || !out) return BH_ERR_INVALID_ARG;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    for (size_t k = 0; k < ctx->retained.size(); ++k) {
        Retained& rt = ctx->retained[k];
        if (rt.fs.out.generation != out->generation || rt.fs.out.out_img != out->out_img) continue;
        // stream order protects the blocks: whoever gets them next is queued behind the kernels that still read them
        for (int i = 0; i < RETAIN_SLOTS; ++i) {
            if (!rt.blocks[i].ptr) continue;
            Buffer& slot = ctx->slots[kRetainSlots[i]];
            if (!slot.ptr) slot = rt.blocks[i];
            else if (ctx->pool.size() < 64) ctx->pool.push_back(rt.blocks[i]);
            else { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(rt.blocks[i].ptr); }
        }
        ctx->retained.erase(ctx->retained.begin() + (long)k);
        return 0;
    }
    return set_error(ctx, BH_ERR_STATE, "render_release: this forward is not retained on this context");
}

int bh_last_render_out(bh_ctx* ctx, BhRenderOut* out) {
    if (!ctx || !out) return BH_ERR_INVALID_ARG;
    if (!ctx->have_forward) return set_error(ctx, BH_ERR_STATE, "no forward render on this context yet");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->far_job.pending) BH_TRY(finish_far_slice(ctx, nullptr));
    *out = ctx->last;
    return 0;
}

int bh_last_list_counts(bh_ctx* ctx, uint32_t* near_pairs, uint32_t* far_pairs) {
    if (!ctx || !near_pairs || !far_pairs) return BH_ERR_INVALID_ARG;
    if (!ctx->have_forward) return set_error(ctx, BH_ERR_STATE, "no forward render on this context yet");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->far_job.pending) BH_TRY(finish_far_slice(ctx, nullptr));
    *near_pairs = ctx->last.num_intersections;
    *far_pairs = 0u;
    if (!ctx->last.tile_offsets_far) return 0;   // one slice: the exact lists
    // slice_info (SLOT_SLICE): [0] near splats  [1] near pairs  [2] tiles the near slice left unsaturated  [3] far pairs
    uint32_t info[4] = {0u, 0u, 0u, 0u};
    BH_HIP(ctx, hipMemcpyAsync(info, ctx->slots[SLOT_SLICE].ptr, sizeof info, hipMemcpyDeviceToHost, ctx->stream));
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    deliver_pending_loss(ctx);
    *near_pairs = info[1];
    *far_pairs = info[2] ? info[3] : 0u;
    return 0;
}

const float* bh_last_v_combined(bh_ctx* ctx) { return ctx ? (const float*)ctx->slots[SLOT_V_COMBINED].ptr : nullptr; }

// ---- primitives ----------------------------------------------------------------
int bh_radix_argsort(bh_ctx* ctx, const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t bits,
                     uint32_t* out_keys, uint32_t* out_vals) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (n > 0 && (!keys || !out_keys || !out_vals)) return set_error(ctx, BH_ERR_INVALID_ARG, "radix_argsort: null argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return radix_argsort(ctx, keys, vals, n, bits, out_keys, out_vals);
}

int bh_prefix_sum(bh_ctx* ctx, const uint32_t* in, uint32_t n, uint32_t* out) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (n > 0 && (!in || !out)) return set_error(ctx, BH_ERR_INVALID_ARG, "prefix_sum: null argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return prefix_sum(ctx, in, nullptr, n, out, false);
}

int bh_image_loss_forward(bh_ctx* ctx, const float* pred_chw, const uint32_t* gt_packed, uint32_t channels, uint32_t h,
                          uint32_t w, const BhLossConfig* cfg, float* loss_map) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!pred_chw || !gt_packed || !cfg || !loss_map || h == 0 || w == 0) return set_error(ctx, BH_ERR_INVALID_ARG, "image_loss_forward: bad argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return launch_image_loss_forward(ctx, pred_chw, gt_packed, channels, h, w, *cfg, loss_map);
}

int bh_image_loss_backward(bh_ctx* ctx, const float* pred_chw, const uint32_t* gt_packed, const float* dl_dmap,
                           uint32_t channels, uint32_t h, uint32_t w, const BhLossConfig* cfg, float* dl_dpred) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!pred_chw || !gt_packed || !dl_dmap || !cfg || !dl_dpred || h == 0 || w == 0) return set_error(ctx, BH_ERR_INVALID_ARG, "image_loss_backward: bad argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return launch_image_loss_backward(ctx, pred_chw, gt_packed, dl_dmap, 0.0f, channels, h, w, *cfg, dl_dpred);
}

int bh_image_loss_value_and_grad(bh_ctx* ctx, const float* img_hwc4, const uint32_t* gt_packed, uint32_t h, uint32_t w,
                                 const BhLossConfig* cfg, float alpha_weight, float* loss_out, float* v_output) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!img_hwc4 || !gt_packed || !cfg || !loss_out || !v_output || h == 0 || w == 0)
        return set_error(ctx, BH_ERR_INVALID_ARG, "image_loss_value_and_grad: bad argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    const size_t hw = (size_t)h * w;
    const bool alpha = alpha_weight > 0.0f;
    return launch_image_loss_fused(ctx, img_hwc4, gt_packed, h, w, *cfg, alpha, 1.0f / (float)(hw * 3), alpha ? alpha_weight / (float)hw : 0.0f,
                                   loss_out, v_output);
}

int bh_adam_step(bh_ctx* ctx, float* param, const float* grad, float* m1, float* m2, uint64_t rows, uint32_t row_len,
                 const float* col_scale, float lr, uint32_t t, int reduce_m2, float beta1, float beta2, float eps) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (rows > 0 && (!param || !grad || !m1 || !m2)) return set_error(ctx, BH_ERR_INVALID_ARG, "adam_step: null argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return launch_adam(ctx, param, grad, m1, m2, rows, row_len, col_scale, lr, t, reduce_m2 != 0, beta1, beta2, eps);
}

int bh_gather_stats(bh_ctx* ctx, float* refine_weight_norm, float* vis_weight, float* max_screen_size,
                    const float* refine_weight, const float* visible, const float* screen_radius, uint64_t n) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return launch_gather_stats(ctx, refine_weight_norm, vis_weight, max_screen_size, refine_weight, visible, screen_radius, n);
}

// ---- the stochastic terms of step() (device_rng.h) ----------------------------------------
void bh_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const Philox4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// sample_background_color (train.rs:896-908): base + U(-strength, strength)^3, clamped to [0,1]
void bh_sample_background(uint64_t seed, uint32_t step, const float base[3], float strength, float out[3]) {
    float u[3] = {0.5f, 0.5f, 0.5f};
    if (strength > 0.0f) {
        const Philox4 r = philox4x32_10(0u, step, 0u, RNG_STREAM_BACKGROUND, (uint32_t)seed, (uint32_t)(seed >> 32));
        u[0] = unit_open(r.x); u[1] = unit_open(r.y); u[2] = unit_open(r.z);
    }
    for (int k = 0; k < 3; ++k) {
        const float v = strength > 0.0f ? base[k] + (2.0f * u[k] - 1.0f) * strength : base[k];
        out[k] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    }
}

int bh_normal_samples(bh_ctx* ctx, uint64_t seed, uint32_t step, uint64_t n, float* out) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (n > 0 && !out) return set_error(ctx, BH_ERR_INVALID_ARG, "normal_samples: null argument");
    if (n > 0xFFFFFFFFull) return set_error(ctx, BH_ERR_INVALID_ARG, "normal_samples: n must fit 32 bits (the splat index is one counter word)");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return launch_normal_samples(ctx, out, n, seed, step);
}

// ---- Mip-Splatting 3D filter ---------------------------------------------------------
int bh_fold_min_scale(bh_ctx* ctx, const float* transforms, const float* raw_opacities, const float* min_scale, uint32_t n,
                      float* out_transforms, float* out_raw_opacities) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (n > 0 && (!transforms || !raw_opacities || !min_scale || !out_transforms || !out_raw_opacities))
        return set_error(ctx, BH_ERR_INVALID_ARG, "fold_min_scale: null argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return launch_fold_min_scale(ctx, transforms, raw_opacities, min_scale, n, out_transforms, out_raw_opacities);
}

int bh_fold_min_scale_backward(bh_ctx* ctx, const float* transforms, const float* raw_opacities, const float* min_scale, uint32_t n,
                               float* v_transforms, float* v_raw_opacities) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (n > 0 && (!transforms || !raw_opacities || !min_scale || !v_transforms || !v_raw_opacities))
        return set_error(ctx, BH_ERR_INVALID_ARG, "fold_min_scale_backward: null argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return launch_fold_min_scale_backward(ctx, transforms, raw_opacities, min_scale, n, v_transforms, v_raw_opacities);
}

int bh_compute_min_scale(bh_ctx* ctx, const float* transforms, uint32_t n, const float* view_cams, uint32_t num_views, float factor,
                         float* out) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    // train.rs:107-109: disabled (factor <= 0) or no cameras -> there is no floor; the caller keeps min_scale = NULL
    if (!(factor > 0.0f) || num_views == 0) return set_error(ctx, BH_ERR_INVALID_ARG, "compute_min_scale: needs factor > 0 and at least one view camera");
    if (!view_cams || (n > 0 && (!transforms || !out))) return set_error(ctx, BH_ERR_INVALID_ARG, "compute_min_scale: null argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    return launch_compute_min_scale(ctx, transforms, n, view_cams, num_views, factor, out);
}

}  // extern "C"

extern "C" int bh_tile_sort_offsets(bh_ctx* ctx, const uint32_t* tile_ids, const uint32_t* compact_gids, uint32_t n, uint32_t num_tiles,
                                    uint32_t* tile_ids_sorted, uint32_t* compact_gids_sorted, uint32_t* tile_offsets) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!tile_offsets || (n > 0 && (!tile_ids || !compact_gids || !tile_ids_sorted || !compact_gids_sorted)))
        return set_error(ctx, BH_ERR_INVALID_ARG, "tile_sort_offsets: null argument");
    if (num_tiles == 0) return set_error(ctx, BH_ERR_INVALID_ARG, "tile_sort_offsets: num_tiles == 0");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t bits = 0;
    while (bits < 32 && (num_tiles >> bits) != 0) bits++;  // render.rs:228
    BH_HIP(ctx, hipMemsetAsync(tile_offsets, 0, (size_t)num_tiles * 2 * 4, ctx->stream));
    if (n == 0) return 0;
    if (tile_sort_supported(bits, n) && !(ctx->knob_tile_sort == 2u))
        return tile_sort_offsets(ctx, tile_ids, compact_gids, n, bits, num_tiles, tile_ids_sorted, compact_gids_sorted, tile_offsets);
    BH_TRY(radix_argsort(ctx, tile_ids, compact_gids, n, bits, tile_ids_sorted, compact_gids_sorted));
    return launch_tile_offsets(ctx, tile_ids_sorted, n, num_tiles, tile_offsets, /*pre_zeroed=*/true);
}

#ifdef BH_TEST_HOOKS
// the split counts K1 left for the context's last forward (context.h SPLIT_MAX); returns 1 and fills out[8], or 0 when that frame split nothing by construction
extern "C" int bh_debug_split_counts(bh_ctx* ctx, uint32_t* out) {
    if (!ctx || !out) return BH_ERR_INVALID_ARG;
    if (!ctx->last_split) return 0;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    BH_HIP(ctx, hipMemcpyAsync(out, ctx->last_split, 8 * 4, hipMemcpyDeviceToHost, ctx->stream));
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 1;
}
extern "C" int bh_debug_fill_train_scratch(bh_ctx* ctx, uint32_t pattern) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    const Buffer& s = ctx->slots[SLOT_GRADS];
    if (!s.ptr || s.cap < 4) return set_error(ctx, BH_ERR_STATE, "debug_fill_train_scratch: no train step on this context yet");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    BH_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(s.ptr), (int)pattern, s.cap / 4, ctx->stream));
    return 0;
}
#endif

// ---- training step -------------------------------------------------------------
static int train_step_impl(bh_ctx* ctx, const BhTrainConfig* cfg, BhTrainState* st, const BhTrainBatch* batch,
                           bh_grad_hook hook, void* hook_user, float grad_scale, BhTrainStats* stats);

extern "C" int bh_train_step(bh_ctx* ctx, const BhTrainConfig* cfg, BhTrainState* st, const BhTrainBatch* batch,
                             bh_grad_hook hook, void* hook_user, float grad_scale, BhTrainStats* stats) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    const int rc = train_step_impl(ctx, cfg, st, batch, hook, hook_user, grad_scale, stats);
    if (rc != 0) {
        // A step that fails behind its forward must not leave a deferred far-slice decision behind: the job holds the CALLER's
        // parameter pointers (a per-tile-cut job replays the whole forward from them) and the caller is free to release or
        // re-allocate them after a failed step (a refine changes n and every buffer).  Drop it; the frame it belonged to is
        // incomplete, so nothing may be replayed from it either, and the view's next frame is rendered with complete lists.
        if (ctx->far_job.pending) {
            ctx->far_job.pending = false;
            if (ctx->far_job.by_cut && ctx->far_job.view && ctx->far_job.view->exact_frames == 0u) ctx->far_job.view->exact_frames = 1u;
        }
        ctx->far_job.view = nullptr;
        ctx->have_forward = false;
        ctx->ext_visible = nullptr; ctx->ext_max_radius = nullptr; ctx->ext_visible_floats = 0;
        ctx->ext_grad_begin = nullptr; ctx->ext_grad_floats = 0;
        ctx->clears.begin_forward();
        ctx->defer_far = false;
    }
    return rc;
}

static int train_step_impl(bh_ctx* ctx, const BhTrainConfig* cfg, BhTrainState* st, const BhTrainBatch* batch,
                           bh_grad_hook hook, void* hook_user, float grad_scale, BhTrainStats* stats) {
    if (!cfg || !st || !batch || !stats) return set_error(ctx, BH_ERR_INVALID_ARG, "train_step: null argument");
    if (!st->transforms || !st->sh_coeffs || !st->raw_opacities || !st->m1_transforms || !st->m2_transforms || !st->m1_sh ||
        !st->m2_sh || !st->m1_opac || !st->m2_opac || !st->refine_weight_norm || !st->vis_weight || !st->max_screen_size ||
        !batch->gt_packed)
        return set_error(ctx, BH_ERR_INVALID_ARG, "train_step: null state tensor");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = st->n, C = (st->sh_degree + 1) * (st->sh_degree + 1);
    const uint32_t W = batch->camera.img_w, H = batch->camera.img_h;
    // train.rs:183 — committed to `st` only once the update is queued: a step that fails half-way (OOM, hook / RCCL error)
    // applied no update, so Adam's t and the lr_mean schedule must not have advanced for the retry
    const uint32_t step = st->step_count + 1;

    // ---- the exchange buffer: visible[N] | v_transforms[10N] | v_sh[3CN] | v_raw_opac[N] | v_refine[N], every section
    // starting on a 16-byte boundary (padded to a multiple of 4 floats; the padding stays zero) so the update kernel
    // can move it with 128-bit accesses whatever N is.  One buffer = one collective per step for a multi-GPU caller.
    auto pad4 = [](size_t x) { return (x + 3) & ~(size_t)3; };
    const size_t o_tr = pad4(n), o_sh = o_tr + pad4((size_t)n * 10), o_op = o_sh + pad4((size_t)n * 3 * C), o_ref = o_op + pad4(n);
    const size_t exch_count = o_ref + pad4(n);
    auto* exch = (float*)ensure(ctx, SLOT_GRADS, (exch_count ? exch_count : 4) * 4);
    auto* s_radius = (float*)ensure(ctx, SLOT_STATS, (size_t)(n ? n : 1) * 4);
    if (!exch || !s_radius) return BH_ERR_OOM;
    float* s_visible = exch;
    float* s_refine = exch + o_ref;

    // ---- Mip-Splatting 3D filter: render fold_min_scale(params) (bwd/burn_glue.rs:260-270)
    const float* r_transforms = st->transforms;
    const float* r_raw_opac = st->raw_opacities;
    if (st->min_scale && n > 0) {
        ProfScope ps(ctx, "FoldMinScale");
        auto* ft = (float*)ensure(ctx, SLOT_FOLDED_TRANSFORMS, (size_t)n * 10 * 4);
        auto* fo = (float*)ensure(ctx, SLOT_FOLDED_RAW_OPAC, (size_t)n * 4);
        if (!ft || !fo) return BH_ERR_OOM;
        BH_TRY(launch_fold_min_scale(ctx, st->transforms, st->raw_opacities, st->min_scale, n, ft, fo));
        r_transforms = ft;
        r_raw_opac = fo;
    }

    // ---- forward (train.rs:211-215); `visible` lands directly in the exchange buffer
    BhRenderOut ro;
    const uint32_t flags = BH_FLAG_BWD_INFO | (cfg->render_mip ? BH_FLAG_MIP : 0) | (cfg->exact_lists ? 0 : BH_FLAG_SLICED_LISTS);
    ctx->ext_visible = s_visible;
    ctx->ext_visible_floats = o_tr;  // the forward clears the section incl. its padding
    ctx->ext_max_radius = s_radius;
    ctx->ext_grad_begin = exch + o_tr;       // ... and the whole gradient span (padding included): K1 does both on its way
    ctx->ext_grad_floats = exch_count - o_tr;
    // ... unless nobody but this step's own update reads the gradients (one GPU, no hook): then only the refine-weight vector
    // (N floats, the span's last section) is cleared.  K18 writes the rows of the splats that received a gradient and marks them in
    // that vector's sign bit, the update kernel takes every unmarked row as zero — at SH degree 3 the zero-fill was most of K1's
    // HBM traffic (236 of 330 MB per step at 1 M splats).
    // (knob_force_exchange: a one-rank communicator still walks the whole exchange path — its kernels, readback and host logic —
    // with the collectives themselves degenerate: the one-GPU measurement of what the path costs per step)
    const bool exchanging = hook || (ctx->comm && (ctx->comm_world > 1 || ctx->knob_force_exchange));
    // one frame split over the ranks by strips of tile rows: the caller's image hook moves the strips / halos, or — no hook, a
    // communicator on the ctx, a proper tile-row window in the camera — the library exchanges the halos itself (comm.hip)
    const ViewUniforms win_u = make_uniforms(batch->camera);
    const bool window_partial = win_u.tile_y0 != 0u || win_u.tile_y1 != win_u.tile_bh;
    // (a one-rank communicator has no neighbours: a window without strip_loss then runs as it does without a communicator)
    const bool native_tiles = !batch->image_hook && window_partial && ctx->comm != nullptr && !hook && (batch->strip_loss || ctx->comm_world > 1);
    if (native_tiles && !batch->strip_loss)
        return set_error(ctx, BH_ERR_INVALID_ARG, "train_step: a tile-row window without an image hook needs strip_loss = 1 (the library exchanges the strips' halos, not whole frames)");
    const bool tile_mode = batch->image_hook != nullptr || native_tiles;
    const bool masked_grads = !exchanging && !tile_mode && !ctx->knob_zero_grads && n > 0;
    if (masked_grads) {
        ctx->ext_grad_begin = exch + o_ref;
        ctx->ext_grad_floats = exch_count - o_ref;
    }
    // depth-sliced lists: whether the far slice has to run is known once the near slice's blend has; a single-GPU step does not
    // wait for that — the loss kernels are queued behind the near slice first (below) and the host reads the answer while they run
    // (a tile-partitioned frame hands the image to its hook right after the forward: there the forward waits itself)
    ctx->defer_far = !tile_mode;
    const uint32_t caller_view = ctx->view_id;
    ctx->view_id = batch->view_id;
    const int frc = bh_render_forward(ctx, &batch->camera, n, st->sh_degree, r_transforms, st->sh_coeffs, r_raw_opac,
                                      batch->background, flags, &ro);
    ctx->view_id = caller_view;
    ctx->defer_far = false;
    ctx->ext_visible = nullptr;
    ctx->ext_max_radius = nullptr;
    ctx->ext_grad_begin = nullptr;
    ctx->ext_grad_floats = 0;
    BH_TRY(frc);

    // ---- tile-partitioned frame: fetch the other ranks' strips (not in the reference: SURVEY.md §8e)
    if (tile_mode) {
        ProfScope ps(ctx, "ImageExchange");
        const ViewUniforms& wu = win_u;
        const uint32_t r0 = wu.tile_y0 * TILE_WIDTH, r1 = wu.tile_y1 * TILE_WIDTH < H ? wu.tile_y1 * TILE_WIDTH : H;
        if (native_tiles) BH_TRY(comm_exchange_strip_halos(ctx, ro.out_img, H, W, r0, r1));
        else if (batch->image_hook(batch->image_hook_user, ro.out_img, H, W, r0, r1) != 0) return set_error(ctx, BH_ERR_STATE, "image hook failed");
    }

    // ---- loss (train.rs:227-260)
    const bool ssim_on = cfg->ssim_weight > 0.0f;
    BhLossConfig lc{};
    lc.l1_weight = ssim_on ? 1.0f - cfg->ssim_weight : 1.0f;
    lc.ssim_weight = ssim_on ? -cfg->ssim_weight : 0.0f;
    const bool bg_nonzero = batch->background[0] != 0.0f || batch->background[1] != 0.0f || batch->background[2] != 0.0f;
    lc.composite_bg = (batch->has_alpha && bg_nonzero) ? 1 : 0;
    lc.bg[0] = batch->background[0]; lc.bg[1] = batch->background[1]; lc.bg[2] = batch->background[2];
    lc.mask = batch->alpha_is_mask ? 1 : 0;
    const bool alpha_match = batch->has_alpha && !batch->alpha_is_mask && cfg->match_alpha_weight > 0.0f;
    const size_t hw = (size_t)W * H;
    auto* v_output = (float*)ensure(ctx, SLOT_V_OUTPUT, hw * 16);
    auto* loss_dev = (float*)ensure(ctx, SLOT_LOSS_SCALAR, 16);
    if (!v_output || !loss_dev) return BH_ERR_OOM;
    // the loss scalar goes straight into pinned host memory (bh_sync hands it to stats->loss): no copy launch
    float* loss_host = reinterpret_cast<float*>(ctx->host_counters) + HOST_LOSS_WORD;
    const float dl_rgb = 1.0f / (float)(hw * 3);
    const float dl_alpha = alpha_match ? cfg->match_alpha_weight / (float)hw : 0.0f;
    // fused forward + backward of the loss on the rasterizer's [H,W,4] image (loss_fused.hip)
    auto queue_loss = [&]() -> int {
        // a deferred far-slice decision without an event behind the blend: this loss kernel tells the host when it starts
        uint32_t* started_host = nullptr;
        uint32_t started_tag = 0u;
        if (ctx->far_job.pending && !ctx->far_job.gate_event_recorded && !ctx->gate_signal_queued) {
            started_host = ctx->host_counters + HOST_GATE_TAG_WORD;
            started_tag = ctx->far_job.gate_tag;
            ctx->gate_signal_queued = true;
        }
        if (tile_mode && batch->strip_loss) {
            // one frame over several ranks: the hook delivered only the 21-px halos; loss and dL/dimg for this rank's strip
            // (stats->loss is the strip's share of the mean: the ranks' shares add up to the frame's loss)
            const ViewUniforms wu = make_uniforms(batch->camera);
            return launch_image_loss_fused_window(ctx, ro.out_img, batch->gt_packed, H, W, lc, alpha_match, dl_rgb, dl_alpha, wu.tile_y0, wu.tile_y1,
                                                  loss_dev, v_output, loss_host, started_host, started_tag);
        }
        return launch_image_loss_fused(ctx, ro.out_img, batch->gt_packed, H, W, lc, alpha_match, dl_rgb, dl_alpha, loss_dev, v_output, loss_host, started_host,
                                       started_tag);
    };
#ifdef BH_TEST_HOOKS
    if (ctx->knob_fail_loss_at && ++ctx->train_steps_seen == ctx->knob_fail_loss_at)   // (the forward is queued, a deferred far slice may be pending)
        return set_error(ctx, BH_ERR_OOM, "train_step: injected failure between the forward and the loss (BH_TEST_FAIL_LOSS_AT)");
#endif
    // A view whose forecast keeps missing (two of its last eight cut frames) decides FIRST — the host waits for the near pass's blend,
    // ~15 us of bubble — instead of queueing loss kernels that a second attempt would make worthless (~120 us).  One isolated miss
    // does not switch: eight bubbles cost more than the one wasted loss they would insure against at a 3 % miss rate.
    if (ctx->far_job.pending && ctx->far_job.view && __builtin_popcount(ctx->far_job.view->penalty) >= 2) {
        BH_TRY(finish_far_slice(ctx, nullptr));
        ro = ctx->last;   // (a second attempt replaces the frame's outputs)
    }
    BH_TRY(queue_loss());
    if (ctx->far_job.pending) {
        // The loss above ran on the near slice's image, which is the frame's image unless some tile was left unsaturated — the
        // host finds out while it runs.  Then (rare: the near lists carry a margin over what the view needed last time)
        // the far slice is queued and the loss is evaluated again on the finished image.
        bool far_ran = false;
        BH_TRY(finish_far_slice(ctx, &far_ran));
        if (far_ran) {
            ro = ctx->last;
            BH_TRY(queue_loss());
        }
    }

    // ---- multi-GPU exchange, part 1 (mask-keyed mode, exchange.hip): the visible flags are final once the forward (incl. a far
    // slice, if it had to run) is, so they are summed, the union of contributing splats is listed and its size starts travelling
    // to the host NOW — the backward hides the collective's latency and the readback, and the host finds the count ready
    // "sum `cnt` floats at `p` over the ranks, in place": the caller's hook, or the library's communicator
    auto sum_over_ranks = [&](float* p, uint64_t cnt) -> int {
        if (hook) return hook(hook_user, p, cnt) == 0 ? 0 : set_error(ctx, BH_ERR_STATE, "gradient hook failed");
        return comm_allreduce(ctx, p, cnt, false);
    };
    const bool keyed = exchanging && batch->exchange_mode == 1 && n > 0;
    uint32_t* union_idx = nullptr;
    float* compact = nullptr;
    bool flags_on_side = false;
    if (keyed) {
        ProfScope ps(ctx, "FlagExchange");
        const uint32_t nblk = (n + 4095u) / 4096u;
        auto* blocks = (uint32_t*)ensure(ctx, SLOT_EXCH_BLOCKS, ((size_t)nblk + 2) * 4);   // [nblk] block offsets, then the total
        union_idx = (uint32_t*)ensure(ctx, SLOT_EXCH_IDX, (size_t)n * 4);
        if (!blocks || !union_idx) return BH_ERR_OOM;   // (the compact block is sized once the union's size is known, below)
        // The library's communicator runs this on its own stream: the flag sum (4 B per splat over xGMI), the union listing and the
        // count's way to the host then overlap the backward on the ctx stream instead of standing in front of it; nothing on the
        // ctx stream touches the flag section before the update, which waits for the side stream below.  (A caller's hook decides
        // its own streams.)
        flags_on_side = !hook && ctx->comm_stream != nullptr;
        hipStream_t main_stream = ctx->stream;
        if (flags_on_side) {
            BH_HIP(ctx, hipEventRecord(ctx->comm_ev, main_stream));
            BH_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ev, 0));
            ctx->stream = ctx->comm_stream;   // (single-threaded ctx: the launchers below take the stream from it)
        }
        int rc = sum_over_ranks(exch, (uint64_t)o_tr);
        if (rc == 0) rc = launch_union_index(ctx, s_visible, n, blocks, blocks + nblk, union_idx);
        if (rc == 0) rc = check_hip(ctx, hipMemcpyAsync(reinterpret_cast<uint32_t*>(ctx->host_counters) + 8, blocks + nblk, 4, hipMemcpyDeviceToHost, ctx->stream), "flag count readback");
        if (rc == 0) rc = check_hip(ctx, hipEventRecord(ctx->readback_ev, ctx->stream), "flag count event");
        ctx->stream = main_stream;
        BH_TRY(rc);
    }

    // ---- backward (train.rs:278)
    float* g_tr = exch + o_tr;
    float* g_sh = exch + o_sh;
    float* g_op = exch + o_op;
    ctx->ext_grad_begin = g_tr;              // one zero-fill of the whole gradient span (padding included)
    ctx->ext_grad_floats = exch_count - o_tr;
    // masked gradients rely on last step's row marks (sign bits of the refine-weight vector) being gone: K1 of the frame's (last)
    // forward clears the vector on its way when the span is float4-addressable (always, with the pad4 layout and hipMalloc's
    // alignment) - not an implicit invariant: if it could not, clear it here.  Decided HERE, behind every second attempt a failed
    // forecast may have caused: the record describes the forward the backward is about to replay.
    if (masked_grads) {
        if (ctx->clears.span != GradClears::ZEROED) BH_HIP(ctx, hipMemsetAsync(exch + o_ref, 0, (exch_count - o_ref) * sizeof(float), ctx->stream));
        ctx->clears.mark_rows();   // this step's backward runs K18 in marking mode and fills nothing
    }
    // the refine weight's one consumer stops reading it at growth_stop_iter (train.rs:589-614): from then on the blend backward
    // runs without it (refine_weight_norm stays as refine() zeroed it)
    ctx->bwd_skip_refine = cfg->growth_stop_iter != 0u && step >= cfg->growth_stop_iter;
    const int brc = bh_render_backward(ctx, v_output, r_transforms, st->sh_coeffs, r_raw_opac, g_tr, g_sh, g_op, s_refine);
    ctx->bwd_skip_refine = false;
    ctx->ext_grad_begin = nullptr;
    ctx->ext_grad_floats = 0;
    BH_TRY(brc);
    if (st->min_scale && n > 0) {  // chain d/d(folded) -> d/d(raw) through the fold (autodiff of gaussian_splats.rs:86-111)
        ProfScope ps(ctx, "FoldMinScaleBackward");
        BH_TRY(launch_fold_min_scale_backward(ctx, st->transforms, st->raw_opacities, st->min_scale, n, g_tr, g_op));
    }
    // ---- multi-GPU exchange, part 2 (not in the reference: SURVEY.md §8e)
    stats->exchange_rows = 0;
    if (exchanging) {
        ProfScope ps(ctx, "GradExchange");
        if (keyed) {
            // only the gradient rows of splats some rank saw (the count was requested right after the forward).  One frame
            // split over the ranks: the strips' refine weights are partial sums too and travel as one more column.
            float* g_ref = tile_mode ? s_refine : nullptr;
            const uint32_t c3 = 3 * C, k = 11 + c3 + (tile_mode ? 1u : 0u);
            BH_HIP(ctx, hipEventSynchronize(ctx->readback_ev));
            if (flags_on_side) BH_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->readback_ev, 0));   // the union list and the summed flags are the side stream's
            const uint32_t rows = reinterpret_cast<uint32_t*>(ctx->host_counters)[8];
            if (rows == 0) {
                // no rank saw any splat: every gradient row is zero everywhere, nothing to send (same decision on all ranks)
            } else if ((uint64_t)rows * 2 <= n) {
                // sized by what this step sends (grow-only arena): megabytes for a view's worth of rows, not n / 2 rows up front
                compact = (float*)ensure(ctx, SLOT_EXCH_COMPACT, (size_t)rows * k * 4);
                if (!compact) return BH_ERR_OOM;
                BH_TRY(launch_exchange_rows(ctx, true, union_idx, rows, c3, g_tr, g_sh, g_op, g_ref, compact));
                BH_TRY(sum_over_ranks(compact, (uint64_t)rows * k));
                BH_TRY(launch_exchange_rows(ctx, false, union_idx, rows, c3, g_tr, g_sh, g_op, g_ref, compact));
                stats->exchange_rows = rows;
            } else {
                BH_TRY(sum_over_ranks(exch + o_tr, (uint64_t)((tile_mode ? exch_count : o_ref) - o_tr)));
            }
        } else if (tile_mode) {
            BH_TRY(sum_over_ranks(exch, (uint64_t)exch_count));
        } else {
            BH_TRY(sum_over_ranks(exch, (uint64_t)o_ref));
        }
    }
    // ---- refine statistics (train.rs:280-298) + optimizer (train.rs:300-381): one launch
    const double decay = std::pow(cfg->lr_mean_end / cfg->lr_mean, 1.0 / (double)cfg->total_train_iters);
    const double lr_mean = cfg->lr_mean * std::pow(decay, (double)((int)step - 1)) * (double)cfg->median_scene_scale;
    // visibility-gated noise on the means (train.rs:389-416): injected samples, or drawn on the device (device_rng.h)
    const bool noise_on = cfg->mean_noise_weight > 0.0f && n > 0 && (batch->noise_samples || batch->device_noise);
    const float noise_scale = (float)lr_mean * cfg->mean_noise_weight;
    // device-drawn noise without a 3D-filter floor rides on the update launch (the gate reads the opacity that launch just wrote)
    const bool noise_fused = noise_on && !batch->noise_samples && !st->min_scale;
    {
        ProfScope ps(ctx, "OptimizerStep");
        float tab[10];
        for (int i = 0; i < 3; ++i) tab[i] = (float)lr_mean;
        for (int i = 3; i < 7; ++i) tab[i] = (float)cfg->lr_rotation;
        for (int i = 7; i < 10; ++i) tab[i] = (float)cfg->lr_scale;
        const NoiseArgs na{batch->noise_seed, step, noise_scale, cfg->median_scene_scale};
        // sh: DC at full lr, bands >= 1 scaled by 1/lr_coeffs_sh_scale
        BH_TRY(launch_train_update(ctx, st, g_tr, g_sh, g_op, s_refine, s_visible, s_radius, grad_scale, tile_mode, tab,
                                   (float)cfg->lr_coeffs_dc, 1.0f / cfg->lr_coeffs_sh_scale, (float)cfg->lr_opac, step, 0.9f, 0.999f, 1e-15f,
                                   noise_fused ? &na : nullptr, masked_grads));
    }
    st->step_count = step;   // the update is queued: the step counts
    if (noise_on && !noise_fused) {
        ProfScope ps(ctx, "MeanNoise");
        // the gate reads splats.opacities() of the UPDATED parameters (train.rs:389), i.e. through the fold when a floor is set
        const float* gate_opac = st->raw_opacities;
        if (st->min_scale && n > 0) {
            auto* ft = (float*)ctx->slots[SLOT_FOLDED_TRANSFORMS].ptr;
            auto* fo = (float*)ctx->slots[SLOT_FOLDED_RAW_OPAC].ptr;
            BH_TRY(launch_fold_min_scale(ctx, st->transforms, st->raw_opacities, st->min_scale, n, ft, fo));
            gate_opac = fo;
        }
        BH_TRY(launch_mean_noise(ctx, st->transforms, gate_opac, s_visible, batch->noise_samples, n, noise_scale, cfg->median_scene_scale,
                                 batch->noise_seed, step));
    }
    stats->num_visible = ro.num_visible;
    stats->num_intersections = ro.num_intersections;
    stats->lr_mean = lr_mean;
    // the loss kernel wrote the scalar into pinned staging (a copy into the caller's pageable struct would
    // stall the host until the whole step has run); bh_sync moves it into stats->loss
    ctx->pending_loss_dst = &stats->loss;
    stats->loss = 0.0f;
    return 0;
}
