// rasterize.hip — tile offsets, forward alpha compositing, backward compositing.
//
// Reference: brush-render/src/get_tile_offset.rs:11-58, kernels/rasterize.rs:27-190,
// bwd/kernels/rasterize_backwards.rs:101-390 (paths under /root/reference/crates).
//
// MI355X design (not the reference's shapes):
//   * forward: ONE wave64 per 16x16 tile, 4 pixels per lane (a 16x4 strip per lane
//     row, rows ly, ly+4, ly+8, ly+12), so the whole tile is wave-synchronous: no
//     s_barrier in the blend loop, the per-tile "all pixels done" early-out is one
//     ballot, and the per-splat operands are uniform LDS reads.  The reference uses
//     256 threads and a barrier + LDS atomic counter per 256-splat batch.
//   * backward: same one-wave-per-tile, per-PIXEL forward-order replay with the
//     pixel state in registers, and a wave reduction of the 10 per-splat gradients
//     followed by ONE 10-lane float atomic per (splat, tile).  The reference's
//     32-thread per-splat diagonal schedule needs a barrier per step and would idle
//     half of a wave64.
//   * block -> tile mapping is XCD-aware: consecutive workgroup ids land on
//     different XCDs, so each XCD gets a contiguous band of tiles and its L2 sees
//     the spatially coherent part of `projected`.
//   * exp() in the blend loops is exp_blend below (a base-2 polynomial shaped for the gfx950
//     issue rates), restated identically by the oracle: images are bit-identical.
#include <hip/hip_ext.h>

#include <type_traits>

#include "context.h"

namespace bh {

// ---------------------------------------------------------------------------
// K15: get_tile_offsets (get_tile_offset.rs:11-58)
// ---------------------------------------------------------------------------
// Four consecutive intersections per thread (one 16-byte load + the element in front of them).
// dyn (depth-sliced forward): the list's length lives in device memory (n_dev, 0 when *gate == 0) and the list starts *base
// entries into tile_ids; the offsets written are absolute.  All NULL: the plain kernel.
struct OffsetsDyn {
    const uint32_t* n_dev = nullptr;
    const uint32_t* gate = nullptr;
    const uint32_t* base = nullptr;
};
__global__ __launch_bounds__(256) void tile_offsets_kernel(const uint32_t* __restrict__ tile_ids, uint32_t num_isect,
                                                          uint32_t num_tiles, uint32_t* __restrict__ tile_offsets, OffsetsDyn dyn) {
    uint32_t ofs = 0;
    if (dyn.gate && *dyn.gate == 0u) return;
    if (dyn.n_dev) { const uint32_t v = *dyn.n_dev; num_isect = v < num_isect ? v : num_isect; }
    if (dyn.base) { ofs = *dyn.base; tile_ids += ofs; }
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 4u;
    if (i0 >= num_isect) return;
    uint32_t t[4];
    if (i0 + 4u <= num_isect && (ofs & 3u) == 0u) {
        const uint4 v = *reinterpret_cast<const uint4*>(&tile_ids[i0]);
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) t[k] = i0 + k < num_isect ? tile_ids[i0 + k] : 0xFFFFFFFFu;
    }
    uint32_t prev = i0 > 0u ? tile_ids[i0 - 1u] : 0xFFFFFFFFu;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t i = i0 + k;
        if (i < num_isect) {
            const uint32_t tid = t[k];
            if (tid < num_tiles) {  // (sentinel rows are skipped)
                if (i == num_isect - 1u) tile_offsets[tid * 2 + 1] = ofs + i + 1u;
                if (i == 0u) {
                    tile_offsets[tid * 2] = ofs;
                } else if (tid != prev) {
                    if (prev < num_tiles) tile_offsets[prev * 2 + 1] = ofs + i;
                    tile_offsets[tid * 2] = ofs + i;
                }
            } else if (i > 0u && prev < num_tiles) {
                // valid -> sentinel transition: close the last valid tile (the reference leaves its end at 0 here,
                // get_tile_offset.rs:28-57 / SURVEY App. B.2)
                tile_offsets[prev * 2 + 1] = ofs + i;
            }
            prev = tid;
        }
    }
}

int launch_tile_offsets(bh_ctx* ctx, const uint32_t* tile_ids_sorted, uint32_t num_isect, uint32_t num_tiles,
                        uint32_t* tile_offsets, bool pre_zeroed) {
    // the 8 x 16 work-class counters of the backward's tile order sit right behind the table: one fill clears both
    // (pre_zeroed: the forward's K1 already did, project.hip ForwardPrep)
    if (!pre_zeroed) BH_HIP(ctx, hipMemsetAsync(tile_offsets, 0, ((size_t)num_tiles * 2 + 8 * LPT_CLASSES) * 4, ctx->stream));
    if (num_isect == 0) return 0;
    hipLaunchKernelGGL(tile_offsets_kernel, dim3((num_isect + 1023) / 1024), dim3(256), 0, ctx->stream, tile_ids_sorted, num_isect, num_tiles, tile_offsets, OffsetsDyn{});
    BH_LAUNCH_CHECK(ctx, "tile_offsets_kernel");
    return 0;
}

// the table must be zero already; n_max bounds *n_dev
int launch_tile_offsets_dev(bh_ctx* ctx, const uint32_t* tile_ids_sorted, uint32_t n_max, const uint32_t* n_dev, const uint32_t* gate,
                            const uint32_t* base, uint32_t num_tiles, uint32_t* tile_offsets) {
    if (n_max == 0) return 0;
    OffsetsDyn dyn;
    dyn.n_dev = n_dev;
    dyn.gate = gate;
    dyn.base = base;
    hipLaunchKernelGGL(tile_offsets_kernel, dim3((n_max + 1023) / 1024), dim3(256), 0, ctx->stream, tile_ids_sorted, n_max, num_tiles, tile_offsets, dyn);
    BH_LAUNCH_CHECK(ctx, "tile_offsets_kernel");
    return 0;
}

// ---------------------------------------------------------------------------
// shared pieces of the two blend kernels
// ---------------------------------------------------------------------------
struct RasterUniforms {
    uint32_t tile_bw, num_tiles, img_w, img_h;  // num_tiles = tiles of the rendered window
    uint32_t tile_begin;                        // first tile id of the window
    float bg_r, bg_g, bg_b;
    float rcp_class_width;                      // work classes of the backward's longest-first tile order (see LPT below)
    uint32_t band_mode;                         // context.h XCD BANDS
};

// ---- longest-first tile order for the backward ------------------------------------------------------------
// A tile is one wave and the chip holds only ~8 waves per SIMD over the whole launch (8160 tiles at 1080p), so
// WHICH tiles share a SIMD decides the makespan: in index order the slowest SIMD carries ~10 % (up to ~40 % per
// wave slot) more blended splats than the mean at the bench workload (scripts/tile_work_hist.py).  The forward
// learns every tile's exact backward work (its list end is shrunk to the last useful splat, rasterize.rs:183-189),
// so it files the tile, per XCD band, into one of LPT_CLASSES work classes (an atomic append), and the backward
// maps block j of an XCD to that band's j-th tile in DESCENDING class order: heavy tiles start first, light ones
// fill the tail.  The band structure (each XCD keeps a contiguous range of tiles for its L2) is unchanged.
// layout of the LPT scratch: LPT_HEADER_WORDS = [8][LPT_CLASSES] counters + 64 spare words (zeroed with tile_offsets), then
// [8][LPT_CLASSES][per] class lists.  Backward jobs: a band's TOP class (every full segment) has more entries than the band has
// tiles and lives in BwdJobs::top_list.  An entry: local tile (24 bits) | segment << 24 | "runs to the tile's end" << 30.
BH_DEV uint32_t lpt_band_tiles(uint32_t num_tiles) { return band_slots(num_tiles); }
BH_DEV size_t lpt_list_offset(uint32_t num_tiles, uint32_t band, uint32_t cls) {
    return (size_t)LPT_HEADER_WORDS + ((size_t)band * LPT_CLASSES + cls) * lpt_band_tiles(num_tiles);
}
BH_DEV uint32_t ckpt_slot(uint32_t range_lo, uint32_t tile, uint32_t seg) { return range_lo / BWD_SEG + tile + seg - 1u; }
constexpr uint32_t JOB_TILE_MASK = 0x00FFFFFFu, JOB_SEG_SHIFT = 24u, JOB_SEG_MASK = 0x3Fu, JOB_LAST_BIT = 1u << 30;

// One staged splat = 12 floats (48 B, 16-B aligned rows):
//   [0..3] x y c00 c01   [4..7] c11 alpha max(r,0) max(g,0)   [8] max(b,0)
//   [9] sigma_cut  (conservative bound: alpha can only reach the cutoff where sigma <= sigma_cut)
//   [10] colour gate bits (raw r/g/b >= 0)   [11] compact gid
constexpr int SPLAT_STRIDE = 12;
constexpr int BATCH = 64;
constexpr uint32_t SPLIT_GROUP = 4;   // splats a quadrant wave of a split tile takes at a time (rasterize_kernel, NQ == 1)
constexpr float SIGMA_CUT_MARGIN = 0.01f;  // >> the error of bh_logf/exp_blend (~1e-7)

// exp(x) for the blend loops, x = -sigma <= 0 wherever the result is used (lanes that fail the
// sigma pre-test compute a value nobody reads).  Base-2 form chosen for gfx950 issue rates: nine
// full-rate VALU ops (fma, sub, fma, 5 fma, lshl_add) where the Cephes sequence of bh_expf
// takes 14 with three half-rate ones (rndne, cvt, ldexp): k = rint(x*log2e) through the 1.5*2^23
// magic add (fused into the product), 2^f from a degree-5 minimax polynomial on [-0.5, 0.5] (1.6e-7 max rel. error), and the
// exponent spliced in by adding k << 23 to the bit pattern.  The CPU checker used by the tests
// restates the same sequence, so images stay bit-identical to it.
BH_DEV float exp_blend(float x) {
    const float s = __builtin_fmaf(x, 1.44269504088896341f, 12582912.0f);
    const float nkf = 12582912.0f - s;                               // -rint(x log2e), exact (as a subtraction: both fmas keep
    const float f = __builtin_fmaf(x, 1.44269504088896341f, nkf);    //  their constant as a literal, no SGPR operand)
    float p = 1.3274633092805743e-3f;
    p = __builtin_fmaf(p, f, 9.671961888670921e-3f);
    p = __builtin_fmaf(p, f, 5.5506784468889236e-2f);
    p = __builtin_fmaf(p, f, 2.4022234976291656e-1f);
    p = __builtin_fmaf(p, f, 6.931470632553101e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return u2f(f2u(p) + (f2u(s) << 23));
}

// 8 XCDs take workgroups round-robin; each XCD owns a band of tiles (context.h XCD BANDS).  (>= num_tiles: the slot names no tile)
BH_DEV uint32_t tile_of_block(uint32_t b, uint32_t num_tiles, uint32_t band_mode) {
    const uint32_t per = band_slots(num_tiles);
    const uint32_t i = b >> 3;
    return i < per ? band_tile(b & 7u, i, per, band_mode) : 0xFFFFFFFFu;
}

// Stage one batch of up to 64 splats of this tile into LDS (lane i stages splat i).
// Everything that is per-splat rather than per-pixel is done here once by the
// staging lane: colour clamp (rasterize.rs:147-149), the gate bits of the backward
// and the conservative sigma bound used for the wave-uniform quadrant skip.
// HALF_CONIC (the forward): c00 and c11 are staged halved — sigma = 1/2 (c00 dx^2 + c11 dy^2) + c01 dx dy then needs no
// multiply by 1/2 per pixel, and scaling by a power of two commutes with every rounding on the way: bit-identical.
template <bool SMOOTH, bool HALF_CONIC>
BH_DEV uint32_t stage_batch(const uint32_t* __restrict__ isect_gids, const float* __restrict__ projected,
                            uint32_t batch_start, uint32_t cnt, int lane, float* s_splat) {
    uint32_t cg = 0;
    if ((uint32_t)lane < cnt) {
        cg = isect_gids[batch_start + lane];
        const float* p = projected + (size_t)cg * 9;
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = p[k];
        const float thr = SMOOTH ? (ALPHA_CUTOFF_MID - 0.5f * ALPHA_CUTOFF_BAND) : ALPHA_CUTOFF_MID;
        const float cut = __builtin_fmaxf(bh_logf(v[5] / thr) + SIGMA_CUT_MARGIN, 0.0f);
        const uint32_t gate = (v[6] >= 0.0f ? 1u : 0u) | (v[7] >= 0.0f ? 2u : 0u) | (v[8] >= 0.0f ? 4u : 0u);
        float4* d = reinterpret_cast<float4*>(s_splat + lane * SPLAT_STRIDE);
        const float diag = HALF_CONIC ? 0.5f : 1.0f;
        d[0] = make_float4(v[0], v[1], diag * v[2], v[3]);
        d[1] = make_float4(diag * v[4], v[5], __builtin_fmaxf(v[6], 0.0f), __builtin_fmaxf(v[7], 0.0f));
        d[2] = make_float4(__builtin_fmaxf(v[8], 0.0f), cut, u2f(gate), u2f(cg));
    }
    return cg;
}

// Pixel layout of the one wave that owns a 16x16 tile: lane l covers (l&7, l>>3)
// inside each of the four 8x8 quadrants q (qx = q&1, qy = q>>1).  A quadrant is the
// skip unit: if no live pixel of it can reach the alpha cutoff for this splat
// (sigma test, before the exp) the whole wave steps over it with one scalar branch.

// ---------------------------------------------------------------------------
// K16: rasterize (kernels/rasterize.rs:27-190)
// ---------------------------------------------------------------------------
// Depth-sliced lists (BH_FLAG_SLICED_LISTS; api.hip has the whole story).  The per-tile lists are built for a NEAR slice of the
// depth order first; PHASE 1 blends it and a tile whose 256 pixels all saturated is final (bit set in done_bits) — at the
// bench workload that is every tile after a fifth of the pairs.  A tile with live pixels left parks its raw (rgb, T) state and
// is counted in *unsat_count; the FAR slice is then listed for those tiles only and PHASE 2 resumes them.  The per-pixel
// arithmetic is the same sequential fold over the same splats in the same order, so the image is bit-identical to PHASE 0 (the
// exact path: one list per tile).  Every phase also leaves a two-word hint for the next frame's slicing in `feedback`.
struct SliceArgs {
    uint32_t* done_bits = nullptr;          // [ceil(T/32)] bit per tile: its pixels are final
    uint32_t* unsat_count = nullptr;        // tiles PHASE 1 left unsaturated (the gate of everything the far slice launches)
    uint32_t* gate_host = nullptr;          // pinned host word, zeroed by the host before the launch: a parked tile stores 1 (no copy launch for the host's decision)
    float* state = nullptr;                 // [H,W,4] raw rgb + signed T of those tiles
    const uint32_t* offsets_near = nullptr; // PHASE 2: the near slice's [T,2] table (shrunk ends: the tile's backward work so far)
    const uint32_t* cum = nullptr;          // cum_tiles_hit [Nv]: the exact list's slot ranges (feedback)
    uint32_t* feedback = nullptr;           // [COUNTER_SLOTS][3]: max slots a saturated tile needed | listed pairs of unsaturated tiles | their number
    // per-tile depth cuts (api.hip): zcut[tile] = the depth key behind which the tile's NEAR list was cut (ZCUT_ALL: complete).
    // Every phase that finishes a tile writes what it needed THIS time into the table (the view's next frame lists against it);
    // cut_active: this frame's lists were built against the table, so an unsaturated tile with a complete list is final in PHASE 1.
    uint32_t* zcut = nullptr;
    const uint32_t* depth_keys_sorted = nullptr;   // [nv] depth keys in compact (depth) order
    uint32_t nv = 0;
    uint32_t cut_active = 0;
    uint32_t* live_bands = nullptr;         // [2]: 32 bands of tile columns | rows that hold a parked tile (PHASE 1 ORs, the far pass reads)
    uint32_t tile_bh = 0;
    uint32_t margin_pct = 150;
    // tile order: all one-wave tiles of a frame are resident at once, so the launch lasts as long as the SIMD whose eight tiles sum
    // to the most work.  order[band][rank] (K1: every XCD band's tiles sorted by the work they had at the view's last frame) makes
    // consecutive blocks take tiles of steadily decreasing work: whichever regular pattern deals blocks to SIMDs, each SIMD's tiles
    // are spread over the whole range instead of being eight neighbours.  work[tile]: what this frame's tiles blended, for next time.
    const uint32_t* order = nullptr;
    uint32_t order_mode = 1;
    uint32_t* split = nullptr;              // split tiles (context.h SPLIT_MAX): split_count[8] | split_scratch[8][SPLIT_MAX][4], or NULL
    uint32_t* work = nullptr;
    BwdJobs jobs{};                         // BWD_INFO, ckpt != NULL: checkpoint the pixel state every BWD_SEG entries and file the tile's backward work as jobs
};

// One tile by its one wave (NQ == 4: four pixels per lane, one per 8 x 8 quadrant), or ONE QUADRANT of a split tile (NQ == 1: one pixel
// per lane, quadrant qsel; context.h SPLIT_MAX).  The per-pixel arithmetic is the same fold over the same splats in the same order
// either way.  The four quadrant waves of a split tile are independent blocks: each stops when ITS 64 pixels are done, stores its
// pixels (and, where the tile may be parked, their raw state), merges (last useful entry, entry reached, unsaturated) into the tile's
// scratch row with atomics, and the last of them to arrive does the tile's bookkeeping with the merged values — which are the
// whole-tile wave's: a splat is useful to the tile iff it is to a quadrant, the tile is saturated iff all four are.  Checkpoints for
// the backward's jobs: a quadrant writes its 64 pixels of every checkpoint it passes, and when it stops early those of every LATER
// segment of the list too (its state no longer changes) — the last finisher cannot know where the slowest quadrant will stop.
template <bool BWD_INFO, bool SMOOTH, int PHASE, int NQ>
BH_DEV void blend_tile(const RasterUniforms& u, const uint32_t* __restrict__ isect_gids, uint32_t* __restrict__ tile_offsets,
                       const float* __restrict__ projected, const uint32_t* __restrict__ global_from_compact, float* __restrict__ out_img,
                       uint32_t* __restrict__ out_packed, float* __restrict__ visible, uint32_t* __restrict__ lpt, const SliceArgs& sl,
                       float* s_splat, const uint32_t local_tile, const uint32_t qsel, uint32_t* split_row) {
    constexpr int NK = NQ == 4 ? 2 : 1;
    const uint32_t bidx = blockIdx.x;
    const uint32_t tile = u.tile_begin + local_tile;
    if (PHASE == 2) {
        if (*sl.unsat_count == 0u) return;                                     // the near slice finished the frame
        if ((sl.done_bits[tile >> 5] >> (tile & 31u)) & 1u) return;            // ... or this tile
    }
    const int lane = threadIdx.x;
    const uint32_t tx0 = (tile % u.tile_bw) * TILE_WIDTH + (NQ == 1 ? 8u * (qsel & 1u) : 0u);
    const uint32_t ty0 = (tile / u.tile_bw) * TILE_WIDTH + (NQ == 1 ? 8u * (qsel >> 1) : 0u);
    const uint32_t px0 = tx0 + (lane & 7), py0 = ty0 + (lane >> 3);
    float pcx[NK], pcy[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        pcx[k] = (float)(px0 + 8 * k) + 0.5f;
        pcy[k] = (float)(py0 + 8 * k) + 0.5f;
    }
    // transmittance; a finished pixel keeps its final T with the sign flipped
    float tr[NQ], pr[NQ], pg[NQ], pb[NQ];
    auto any_live = [&]() {
        bool l = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q) l = l || tr[q] > 0.0f;
        return l;
    };
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const uint32_t px = px0 + 8 * (q & 1), py = py0 + 8 * (q >> 1);
        const bool inside = px < u.img_w && py < u.img_h;
        tr[q] = inside ? 1.0f : -1.0f;
        pr[q] = pg[q] = pb[q] = 0.0f;
        if (PHASE == 2 && inside) {   // resume where PHASE 1 stopped
            const float4 st = *reinterpret_cast<const float4*>(&sl.state[((size_t)px + (size_t)py * u.img_w) * 4]);
            pr[q] = st.x; pg[q] = st.y; pb[q] = st.z; tr[q] = st.w;
        }
    }
    const uint32_t range_lo = tile_offsets[tile * 2];
    const uint32_t range_hi = tile_offsets[tile * 2 + 1];
    uint32_t last_useful = range_lo;
    uint32_t reached = range_lo;        // one past the last splat the loop looked at (forward-only passes keep no last_useful)
    uint32_t sign_mask = 0x80000000u;   // kept in a VGPR: an SGPR operand halves a VALU op's issue rate
    asm volatile("" : "+v"(sign_mask));

    // (A per-batch variant without the v_min of the 0.999 clamp — the backward's trick — was measured here: the second copy of
    //  the loop costs 16 VGPRs, 8 -> 7 waves per SIMD, 160 -> 170 us.)
    uint32_t batch_start = range_lo;
    for (; batch_start < range_hi; batch_start += BATCH) {
        const bool live = any_live();
        if (__ballot(live) == 0ull) break;
        const uint32_t cnt = min((uint32_t)BATCH, range_hi - batch_start);
        if (BWD_INFO && PHASE != 2) {
            // backward jobs: the pixels' state in front of entry k * BWD_SEG of the list is where job k of this tile starts
            const uint32_t done = batch_start - range_lo;
            if (sl.jobs.ckpt && done != 0u && (done % BWD_SEG) == 0u && done / BWD_SEG < BWD_MAX_SEGS) {   // (wave-uniform, scalar)
                const uint32_t slot = ckpt_slot(range_lo, tile, done / BWD_SEG);
                if (slot < sl.jobs.ckpt_cap) {
                    float4* ck = sl.jobs.ckpt + (size_t)slot * 256u + (uint32_t)lane;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) ck[(NQ == 1 ? qsel : (uint32_t)q) * 64u] = make_float4(pr[q], pg[q], pb[q], tr[q]);
                }
            }
        }
        __syncthreads();  // previous batch fully consumed (single wave: cheap)
        const uint32_t cg = stage_batch<SMOOTH, true>(isect_gids, projected, batch_start, cnt, lane, s_splat);
        __syncthreads();
        unsigned long long contrib_mask = 0ull;
        reached = batch_start + cnt;
        uint32_t t_first = 0u;
        if (NQ == 1) {
            // A quadrant wave runs (nearly) alone on its SIMD at the end of the launch: what it costs is the LATENCY of a splat's
            // dependent chain (sigma -> exp polynomial -> alpha -> T), an op every 6-10 cycles.  Two splats at a time: their sigma /
            // exp / alpha chains are independent of the pixel state and of each other and are written side by side (the scheduler
            // interleaves them); the two state updates follow in list order with exactly the one-splat arithmetic, the second splat's
            // live test on the state the first one left.  The pair is skipped when neither can reach its cutoff anywhere (the second
            // splat's test on the state BEFORE the first: conservative).
            auto apply = [&](const float4& s1, const float2& s2, const float alpha, const bool pre) {
                const float w_cut = SMOOTH ? alpha_cutoff_weight(alpha) : (alpha >= ALPHA_CUTOFF_MID ? 1.0f : 0.0f);
                const bool ok = pre && w_cut > 0.0f;
                const float alpha_eff = SMOOTH ? alpha * w_cut : alpha;
                const float next_t = tr[0] * (1.0f - alpha_eff);
                const bool sat = next_t <= 1.0e-4f;
                const bool contrib = ok && !sat;
                const float vis = contrib ? alpha_eff * tr[0] : 0.0f;
                pr[0] = __builtin_fmaf(s1.z, vis, pr[0]);
                pg[0] = __builtin_fmaf(s1.w, vis, pg[0]);
                pb[0] = __builtin_fmaf(s2.x, vis, pb[0]);
                tr[0] = ok ? (sat ? -tr[0] : next_t) : tr[0];
                return contrib;
            };
            bool stopped = false;
            uint32_t t = 0u;
            constexpr uint32_t G = SPLIT_GROUP;   // splats per group (a power of two <= 8: the saturation test sits behind every 8th splat)
            // (per-lane bookkeeping instead of a ballot and five scalar ops per splat: which splats of the batch this LANE blended — OR-reduced
            //  over the wave behind the batch)
            uint32_t lane_lo = 0u, lane_hi = 0u;
            for (; t + G <= cnt; t += G) {
                float4 g0[G], g1[G];
                float2 g2[G];
                float sig[G], al[G];
                bool cand[G];
#pragma unroll
                for (uint32_t g = 0; g < G; ++g) {
                    g0[g] = *reinterpret_cast<const float4*>(&s_splat[(t + g) * SPLAT_STRIDE]);
                    g1[g] = *reinterpret_cast<const float4*>(&s_splat[(t + g) * SPLAT_STRIDE + 4]);
                    g2[g] = *reinterpret_cast<const float2*>(&s_splat[(t + g) * SPLAT_STRIDE + 8]);
                }
                // sigma of the G splats, step by step ACROSS the group: the scheduler otherwise emits one splat's chain after the other
                // (it schedules for register pressure), and a lone wave then pays the full dependent-op latency on every instruction
                {
                    float dx[G], dy[G], axx[G], bx[G], cy[G];
#pragma unroll
                    for (uint32_t g = 0; g < G; ++g) { dx[g] = pcx[0] - g0[g].x; dy[g] = pcy[0] - g0[g].y; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (uint32_t g = 0; g < G; ++g) { axx[g] = g0[g].z * dx[g]; bx[g] = g0[g].w * dx[g]; cy[g] = g1[g].x * dy[g]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (uint32_t g = 0; g < G; ++g) axx[g] = axx[g] * dx[g];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (uint32_t g = 0; g < G; ++g) axx[g] = __builtin_fmaf(cy[g], dy[g], axx[g]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (uint32_t g = 0; g < G; ++g) sig[g] = __builtin_fmaf(bx[g], dy[g], axx[g]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const uint32_t dead = f2u(tr[0]) & sign_mask;
                bool any_cand = false;
#pragma unroll
                for (uint32_t g = 0; g < G; ++g) {
                    cand[g] = (dead | f2u(sig[g])) <= f2u(g2[g].y);   // (the first splat's is its exact live test; the others' are conservative)
                    any_cand = any_cand || cand[g];
                }
                if (__ballot(any_cand) != 0ull) {
                    // exp_blend(-sigma) of the group, the same way (the arithmetic of exp_blend, statement for statement)
                    {
                        float sx[G], f[G], pp[G];
#pragma unroll
                        for (uint32_t g = 0; g < G; ++g) sx[g] = __builtin_fmaf(-sig[g], 1.44269504088896341f, 12582912.0f);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t g = 0; g < G; ++g) f[g] = 12582912.0f - sx[g];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t g = 0; g < G; ++g) f[g] = __builtin_fmaf(-sig[g], 1.44269504088896341f, f[g]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t g = 0; g < G; ++g) pp[g] = __builtin_fmaf(1.3274633092805743e-3f, f[g], 9.671961888670921e-3f);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t g = 0; g < G; ++g) pp[g] = __builtin_fmaf(pp[g], f[g], 5.5506784468889236e-2f);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t g = 0; g < G; ++g) pp[g] = __builtin_fmaf(pp[g], f[g], 2.4022234976291656e-1f);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t g = 0; g < G; ++g) pp[g] = __builtin_fmaf(pp[g], f[g], 6.931470632553101e-1f);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t g = 0; g < G; ++g) pp[g] = __builtin_fmaf(pp[g], f[g], 1.0f);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t g = 0; g < G; ++g) al[g] = g1[g].y * u2f(f2u(pp[g]) + (f2u(sx[g]) << 23));
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (uint32_t g = 0; g < G; ++g) al[g] = __builtin_fminf(0.999f, al[g]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    uint32_t bits = 0u;
#pragma unroll
                    for (uint32_t g = 0; g < G; ++g) {
                        const bool pre = g == 0u ? cand[0] : ((f2u(tr[0]) & sign_mask) | f2u(sig[g])) <= f2u(g2[g].y);
                        const bool any_g = apply(g1[g], g2[g], al[g], pre);
                        if (BWD_INFO) bits |= any_g ? (1u << g) : 0u;
                    }
                    if (BWD_INFO) {   // (t is a multiple of G: the group's bits do not straddle the two words)
                        if (t < 32u) lane_lo |= bits << t;
                        else lane_hi |= bits << (t - 32u);
                    }
                }
                if (((t + G - 1u) & 7u) == 7u) {   // (the one-splat loop's test behind every 8th splat)
                    const bool still = any_live();
                    if (__ballot(still) == 0ull) { reached = batch_start + t + G; stopped = true; break; }
                }
            }
            if (BWD_INFO) {
                // OR over the wave (DPP within rows of 16, then across rows through the swaps' cheaper cousins: readlane of four row leaders)
                uint32_t lo = lane_lo, hi = lane_hi;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    lo |= (uint32_t)__shfl_xor((int)lo, off);
                    hi |= (uint32_t)__shfl_xor((int)hi, off);
                }
                const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(lo);
                if (m != 0ull) {
                    contrib_mask |= m;
                    last_useful = batch_start + 64u - (uint32_t)__builtin_clzll(m);
                }
            }
            t_first = stopped ? cnt : t;   // an odd batch's last splat goes through the one-splat loop below
        }
        for (uint32_t t = t_first; t < cnt; ++t) {
            const float4 s0 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE]);      // x y c00/2 c01
            const float4 s1 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE + 4]);  // c11/2 a r g
            const float2 s2 = *reinterpret_cast<const float2*>(&s_splat[t * SPLAT_STRIDE + 8]);  // b sigma_cut
            const uint32_t cut_bits = f2u(s2.y);
            float a_xx[NK], b_x[NK], c_y[NK], dy[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const float dx = pcx[k] - s0.x;
                a_xx[k] = (s0.z * dx) * dx;
                b_x[k] = s0.w * dx;
                dy[k] = pcy[k] - s0.y;
                c_y[k] = s1.x * dy[k];
            }
            bool any = false;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int k = q & 1, m = q >> 1;
                const float half_qv = __builtin_fmaf(c_y[m], dy[m], a_xx[k]);   // (the staged diagonal is halved)
                const float sigma = __builtin_fmaf(b_x[k], dy[m], half_qv);
                // live pixel and 0 <= sigma <= sigma_cut: ONE unsigned compare — a finished pixel carries a negative T, and
                // its sign bit ORed into sigma's bit pattern puts the key above every cut (one full-rate v_and_or_b32 instead
                // of a second half-rate v_cmp and a scalar and)
                const bool pre = ((f2u(tr[q]) & sign_mask) | f2u(sigma)) <= cut_bits;
                if (__ballot(pre) != 0ull) {
                    const float alpha = __builtin_fminf(0.999f, s1.y * exp_blend(-sigma));
                    const float w_cut = SMOOTH ? alpha_cutoff_weight(alpha) : (alpha >= ALPHA_CUTOFF_MID ? 1.0f : 0.0f);
                    const bool ok = pre && w_cut > 0.0f;  // pre already implies sigma >= 0
                    const float alpha_eff = SMOOTH ? alpha * w_cut : alpha;
                    const float next_t = tr[q] * (1.0f - alpha_eff);
                    const bool sat = next_t <= 1.0e-4f;
                    const bool contrib = ok && !sat;
                    const float vis = contrib ? alpha_eff * tr[q] : 0.0f;
                    pr[q] = __builtin_fmaf(s1.z, vis, pr[q]);   // (explicit fma: part of the numerical specification, as in the CPU checker)
                    pg[q] = __builtin_fmaf(s1.w, vis, pg[q]);
                    pb[q] = __builtin_fmaf(s2.x, vis, pb[q]);
                    tr[q] = ok ? (sat ? -tr[q] : next_t) : tr[q];
                    any = any || contrib;
                }
            }
            if (BWD_INFO) {
                if (__ballot(any) != 0ull) {
                    contrib_mask |= 1ull << t;
                    last_useful = batch_start + t + 1;
                }
            }
            // every pixel of the tile saturated: the rest of the batch (32 splats on average, ~45 VALU ops each just
            // to fail the quadrant tests) cannot contribute.  Checked every 8th splat; the batch loop's own test ends the tile.
            if ((t & 7u) == 7u) {
                const bool still = any_live();
                if (__ballot(still) == 0ull) { reached = batch_start + t + 1; break; }
            }
        }
        if (BWD_INFO) {
            // rasterize.rs:143-145: mark splats that touched at least one pixel
            if ((contrib_mask >> lane) & 1ull) visible[global_from_compact[cg]] = 1.0f;
        }
    }
    const bool live_end = any_live();
    bool saturated = __ballot(live_end) == 0ull;   // every pixel of the tile is done: no later splat can change it

    // (per-tile depth cuts: a tile whose near list was NOT cut holds everything there is — unsaturated or not, it is final)
    const bool near_complete = PHASE == 1 && sl.cut_active != 0u && (sl.zcut[tile] & 1u) == 0u;   // (bit 0: K1 met a pair behind the cut)
    auto store_pixels = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint32_t px = px0 + 8 * (q & 1), py = py0 + 8 * (q >> 1);
            if (px < u.img_w && py < u.img_h) {
                const float tf = __builtin_fabsf(tr[q]);
                const float fr = pr[q] + tf * u.bg_r;
                const float fg = pg[q] + tf * u.bg_g;
                const float fb = pb[q] + tf * u.bg_b;
                const float fa = 1.0f - tf;
                const size_t pix = (size_t)px + (size_t)py * u.img_w;
                if (BWD_INFO) {
                    *reinterpret_cast<float4*>(&out_img[pix * 4]) = make_float4(fr, fg, fb, fa);
                } else {
                    const uint32_t r8 = (uint32_t)clampf(fr * 255.0f, 0.0f, 255.0f);
                    const uint32_t g8 = (uint32_t)clampf(fg * 255.0f, 0.0f, 255.0f);
                    const uint32_t b8 = (uint32_t)clampf(fb * 255.0f, 0.0f, 255.0f);
                    const uint32_t a8 = (uint32_t)clampf(fa * 255.0f, 0.0f, 255.0f);
                    out_packed[pix] = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
                }
            }
        }
    };
    auto park_pixels = [&]() {   // the raw state; the far slice (listed for the unsaturated tiles only) resumes it in PHASE 2
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint32_t px = px0 + 8 * (q & 1), py = py0 + 8 * (q >> 1);
            if (px < u.img_w && py < u.img_h)
                *reinterpret_cast<float4*>(&sl.state[((size_t)px + (size_t)py * u.img_w) * 4]) = make_float4(pr[q], pg[q], pb[q], tr[q]);
        }
    };
    if (NQ == 1) {
        if (BWD_INFO && PHASE != 2 && sl.jobs.ckpt) {
            // this quadrant's part of every checkpoint behind the point where it stopped (its state is final from here on)
            uint32_t done = ((batch_start - range_lo + BWD_SEG - 1u) / BWD_SEG) * BWD_SEG;
            if (done == 0u) done = BWD_SEG;
            for (; done < range_hi - range_lo && done / BWD_SEG < BWD_MAX_SEGS; done += BWD_SEG) {
                const uint32_t slot = ckpt_slot(range_lo, tile, done / BWD_SEG);
                if (slot >= sl.jobs.ckpt_cap) break;
                sl.jobs.ckpt[(size_t)slot * 256u + qsel * 64u + (uint32_t)lane] = make_float4(pr[0], pg[0], pb[0], tr[0]);
            }
        }
        // whether the TILE is parked is only known to the last finisher: a quadrant leaves both — its raw state (PHASE 2 resumes all
        // 256 pixels from there) and its pixels as they stand (final unless PHASE 2 overwrites them)
        if (PHASE == 1 && !near_complete) park_pixels();
        store_pixels();
        uint32_t ticket = 0u, m_useful = 0u, m_reached = 0u, m_unsat = 0u;
        if (lane == 0) {
            if (last_useful > range_lo) atomicMax(&split_row[1], last_useful);
            if (reached > range_lo) atomicMax(&split_row[2], reached);
            if (!saturated) atomicOr(&split_row[3], 1u);
            __threadfence();
            ticket = atomicAdd(&split_row[0], 1u);
            if (ticket == 3u) {
                m_useful = atomicOr(&split_row[1], 0u);
                m_reached = atomicOr(&split_row[2], 0u);
                m_unsat = atomicOr(&split_row[3], 0u);
            }
        }
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != 3u) return;
        last_useful = max(range_lo, (uint32_t)__builtin_amdgcn_readfirstlane(m_useful));
        reached = max(range_lo, (uint32_t)__builtin_amdgcn_readfirstlane(m_reached));
        saturated = __builtin_amdgcn_readfirstlane(m_unsat) == 0u;
    }
    if (PHASE == 1 && !saturated && !near_complete) {
        if (NQ == 4) park_pixels();
        if (lane == 0) {
            if (BWD_INFO) tile_offsets[tile * 2 + 1] = last_useful;
            atomicAdd(sl.unsat_count, 1u);
            if (sl.gate_host) {
                // system-scope release: the host learns that the blend is done from a tag ANOTHER kernel stores (the loss kernel
                // queued behind this one), possibly from another XCD — this word has to be visible to the host before that tag can be
                // (parked tiles are rare: the fence costs nothing)
                *reinterpret_cast<volatile uint32_t*>(sl.gate_host) = 1u;
                __threadfence_system();
            }
            if (sl.live_bands) {   // where the live tiles are: the far pass walks only splats whose box reaches these bands
                const uint32_t ttx = tile % u.tile_bw, tty = tile / u.tile_bw;
                atomicOr(&sl.live_bands[0], 1u << ((ttx * 32u) / u.tile_bw));
                atomicOr(&sl.live_bands[1], 1u << ((tty * 32u) / sl.tile_bh));
            }
        }
        return;
    }

    if (NQ == 4) store_pixels();
    if (BWD_INFO && PHASE != 2 && lpt && sl.jobs.ckpt) {
        // the tile's backward work as JOBS of BWD_SEG list entries (lane s files segment s).  Checkpoints exist in front of every
        // segment the batch loop entered (a split tile: in front of every segment of its list) — up to BWD_MAX_SEGS, and while their
        // slots exist; the last job takes whatever lies behind the last checkpoint.  Classes: full segments in the top class, the
        // tails by their length.
        const uint32_t jwork = last_useful - range_lo;
        const uint32_t last_batch = reached > range_lo ? ((reached - 1u - range_lo) / BATCH) * BATCH : 0u;   // first entry of the last batch the loop entered
        const uint32_t ck_span = NQ == 1 ? (range_hi > range_lo ? range_hi - 1u - range_lo : 0u) : last_batch;
        uint32_t n_ck = min(ck_span / BWD_SEG, BWD_MAX_SEGS - 1u);
        const uint32_t slot0 = ckpt_slot(range_lo, tile, 1u);
        n_ck = slot0 < sl.jobs.ckpt_cap ? min(n_ck, sl.jobs.ckpt_cap - slot0) : 0u;
        const uint32_t nj = min((jwork + BWD_SEG - 1u) / BWD_SEG, n_ck + 1u);
        if ((uint32_t)lane < nj) {
            const uint32_t lo = (uint32_t)lane * BWD_SEG;
            const bool last = (uint32_t)lane + 1u == nj;
            const uint32_t size = (last ? jwork : lo + BWD_SEG) - lo;
            const uint32_t cls = size >= BWD_SEG ? LPT_CLASSES - 1u : (size * (LPT_CLASSES - 1u)) / BWD_SEG;
            const uint32_t band = bidx & 7u;
            const uint32_t pos = atomicAdd(&lpt[band * LPT_CLASSES + cls], 1u);
            const uint32_t entry = local_tile | ((uint32_t)lane << JOB_SEG_SHIFT) | (last ? JOB_LAST_BIT : 0u);
            if (cls == LPT_CLASSES - 1u) sl.jobs.top_list[(size_t)band * sl.jobs.top_cap + pos] = entry;
            else lpt[lpt_list_offset(u.num_tiles, band, cls) + pos] = entry;
        }
    }
    if (lane == 0) {
        if (PHASE == 1) atomicOr(&sl.done_bits[tile >> 5], 1u << (tile & 31u));
        uint32_t work = (BWD_INFO ? last_useful : reached) - range_lo;
        uint32_t listed = range_hi - range_lo;
        if (PHASE == 2) {   // the tile's backward work / list = what both slices contributed
            const uint32_t n_lo = sl.offsets_near[tile * 2], n_hi = sl.offsets_near[tile * 2 + 1];
            work += n_hi - n_lo;
            listed += n_hi - n_lo;
        }
        if (sl.work) sl.work[tile] = work;   // the forecast of this tile's work at the view's next frame
        // rasterize.rs:183-189: shrink the tile's end to one past the last useful splat
        if (BWD_INFO) {
            tile_offsets[tile * 2 + 1] = last_useful;
            if (lpt && !sl.jobs.ckpt) {  // file the tile under its backward work class (longest-first order, see LPT above)
                // work classes.  Logarithmic, eight per octave of blended splats from 8 up (8, 9, .. 15, 16, 18, .. 30, 32, 36, ..: +-4.5 %
                // whatever the frame looks like; 2048+ share the top class) — the default since round 6.  Rounds 2-5 used LINEAR classes
                // 1/64 of the frame's mean LIST length wide, sixteen of them: fine while tiles blend a tenth of their lists, degenerate once
                // they blend a third — a converging run's late phase filed four tiles in five under the top class and K17 ran 26 % longer
                // than with an order (0.66 vs 0.49 ms).  rcp_class_width > 0 still selects them (option lpt_classes = linear: A/B).
                uint32_t cls;
                if (u.rcp_class_width > 0.0f) {
                    cls = min(LPT_CLASSES - 1u, (uint32_t)((float)work * u.rcp_class_width));
                } else if (work < 8u) {
                    cls = 0u;
                } else {
                    const uint32_t p = 28u - (uint32_t)__builtin_clz(work);   // work in [8 << p, 16 << p)
                    cls = min(LPT_CLASSES - 1u, 1u + 8u * p + ((work >> p) & 7u));
                }
                const uint32_t list = (bidx & 7u) * LPT_CLASSES + cls;
                const uint32_t pos = atomicAdd(&lpt[list], 1u);
                lpt[lpt_list_offset(u.num_tiles, bidx & 7u, cls) + pos] = local_tile | JOB_LAST_BIT;
            }
        }
        // per-tile depth cut for this view's NEXT frame: a saturated tile needs the splats up to its last useful one — plus a margin
        // of margin_pct % of its depth rank (parameters move between two visits of a view) — and nothing behind; a tile that
        // did not saturate needs everything there is.  forward-only passes keep no last_useful: `reached` (>= it) is used.
        if (sl.zcut) {
            uint32_t newcut = ZCUT_ALL;
            if (saturated && sl.nv) {
                const uint32_t stop = BWD_INFO ? last_useful : reached;
                // (the table is only kept in automatic mode, whose frames never run PHASE 2: a frame whose forecast failed is rendered
                //  again with complete lists, api.hip finish_far_slice)
                uint32_t g = 0xFFFFFFFFu;
                if (stop > range_lo) g = isect_gids[stop - 1u];
                if (g != 0xFFFFFFFFu) {
                    // (how deep a tile has to go is set by its SLOWEST pixel — an extreme value that jumps when a few small splats
                    //  move, and the default step moves them on purpose: the margin is generous, the lists still a fraction.  The
                    //  host scales it with how long the view will be away and with how forecasts have fared lately: api.hip cut_margin_pct)
                    const unsigned long long mraw = (unsigned long long)g * sl.margin_pct / 100ull;
                    const uint32_t margin = mraw > 128ull ? (mraw < 0x7FFFFFFFull ? (uint32_t)mraw : 0x7FFFFFFFu) : 128u;
                    const uint32_t g2 = (unsigned long long)g + margin < sl.nv ? g + margin : sl.nv - 1u;
                    newcut = sl.depth_keys_sorted[g2] & ~1u;   // (bit 0 is the "incomplete" mark of the table's next frame: clear)
                }
            }
            sl.zcut[tile] = newcut;
        }
        // hint for the next frame's slicing (read back with its counters): how many slots of the exact list a tile needed before
        // it saturated (max over tiles), and how many pairs are listed for tiles that never saturate
        if (sl.feedback) {
            uint32_t* fb = sl.feedback + 3u * (blockIdx.x & (COUNTER_SLOTS - 1u));
            const uint32_t stop = BWD_INFO ? last_useful : reached;
            if (saturated) {
                if (stop > range_lo) atomicMax(&fb[0], sl.cum[isect_gids[stop - 1u]]);
                else if (PHASE == 2) {   // (saturated by the near slice's last splats, nothing blended here)
                    const uint32_t n_lo = sl.offsets_near[tile * 2], n_hi = sl.offsets_near[tile * 2 + 1];
                    if (n_hi > n_lo) atomicMax(&fb[0], sl.cum[isect_gids[n_hi - 1u]]);
                }
            } else {
                if (listed) atomicAdd(&fb[1], listed);
                atomicAdd(&fb[2], 1u);
            }
        }
    }
}

template <bool BWD_INFO, bool SMOOTH, int PHASE>
__global__ __launch_bounds__(64, 8) void rasterize_kernel(RasterUniforms u, const uint32_t* __restrict__ isect_gids,
                                                      uint32_t* __restrict__ tile_offsets, const float* __restrict__ projected,
                                                      const uint32_t* __restrict__ global_from_compact,
                                                      float* __restrict__ out_img, uint32_t* __restrict__ out_packed,
                                                      float* __restrict__ visible, uint32_t* __restrict__ lpt, SliceArgs sl) {
    __shared__ __attribute__((aligned(16))) float s_splat[BATCH * SPLAT_STRIDE];
    const uint32_t bidx = blockIdx.x;
    uint32_t local_tile, qsel = 4u;   // qsel < 4: this block is one quadrant wave of a split tile
    uint32_t* split_row = nullptr;
    if (sl.order) {
        const uint32_t per = band_slots(u.num_tiles);
        uint32_t j = bidx >> 3;
        if (PHASE != 2 && sl.split) {
            // the first split[band] ranks of the band (its heaviest tiles by forecast) take four blocks each, everyone else moves up
            const uint32_t band = bidx & 7u;
            const uint32_t h = sl.split[band];
            if (j < 4u * h) {
                qsel = j & 3u;
                j >>= 2;
                split_row = sl.split + 8u + (band * SPLIT_MAX + j) * 4u;
            } else {
                j -= 3u * h;
            }
        } else if (sl.order_mode == 2u) {   // dealt: consecutive blocks of a band take every seg-th rank (the grid covers 8 * seg ranks per band)
            const uint32_t seg = (per + 7u) / 8u;
            j = (j & 7u) * seg + (j >> 3);
        }
        local_tile = j < per ? sl.order[(bidx & 7u) * per + j] : 0xFFFFFFFFu;
    } else {
        local_tile = tile_of_block(bidx, u.num_tiles, u.band_mode);
    }
    if (local_tile >= u.num_tiles) return;
    if (PHASE != 2 && qsel < 4u)
        blend_tile<BWD_INFO, SMOOTH, PHASE, 1>(u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible, lpt, sl, s_splat, local_tile, qsel, split_row);
    else
        blend_tile<BWD_INFO, SMOOTH, PHASE, 4>(u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible, lpt, sl, s_splat, local_tile, 0u, nullptr);
}

template <bool BWD_INFO, bool SMOOTH>
static void launch_rasterize_phase(int phase, dim3 grid, uint32_t lds_pad, hipStream_t stream, const RasterUniforms& u, const uint32_t* isect_gids, uint32_t* tile_offsets,
                                   const float* projected, const uint32_t* gfc, float* out_img, uint32_t* out_packed, float* visible, uint32_t* lpt,
                                   const SliceArgs& sl) {
    const dim3 block(64);
    if (phase == 1) hipLaunchKernelGGL((rasterize_kernel<BWD_INFO, SMOOTH, 1>), grid, block, lds_pad, stream, u, isect_gids, tile_offsets, projected, gfc, out_img, out_packed, visible, lpt, sl);
    else if (phase == 2) hipLaunchKernelGGL((rasterize_kernel<BWD_INFO, SMOOTH, 2>), grid, block, lds_pad, stream, u, isect_gids, tile_offsets, projected, gfc, out_img, out_packed, visible, lpt, sl);
    else hipLaunchKernelGGL((rasterize_kernel<BWD_INFO, SMOOTH, 0>), grid, block, lds_pad, stream, u, isect_gids, tile_offsets, projected, gfc, out_img, out_packed, visible, lpt, sl);
}

int launch_rasterize(bh_ctx* ctx, const ViewUniforms& vu, const float bg[3], bool bwd_info, bool smooth,
                     const uint32_t* isect_gids, uint32_t* tile_offsets, const float* projected,
                     const uint32_t* global_from_compact, float* out_img, uint32_t* out_packed, float* visible,
                     uint32_t* lpt, float class_width, int phase, const RasterSlice* slice) {
    RasterUniforms u;
    u.band_mode = ctx->knob_band_mode;
    u.rcp_class_width = class_width > 0.0f ? 1.0f / (class_width > 1.0f ? class_width : 1.0f) : 0.0f;   // (<= 0: logarithmic classes)
    u.tile_bw = vu.tile_bw;
    u.num_tiles = vu.tile_bw * (vu.tile_y1 - vu.tile_y0);
    u.tile_begin = vu.tile_bw * vu.tile_y0;
    u.img_w = vu.img_w;
    u.img_h = vu.img_h;
    u.bg_r = bg[0]; u.bg_g = bg[1]; u.bg_b = bg[2];
    SliceArgs sl;
    if (slice) {
        sl.done_bits = slice->done_bits;
        sl.unsat_count = slice->unsat_count;
        sl.gate_host = phase == 1 ? slice->gate_host : nullptr;
        sl.state = slice->state;
        sl.offsets_near = slice->offsets_near;
        sl.cum = slice->cum;
        sl.feedback = slice->feedback;
        sl.zcut = slice->zcut;
        sl.depth_keys_sorted = slice->depth_keys_sorted;
        sl.nv = slice->nv;
        sl.cut_active = slice->cut_active ? 1u : 0u;
        sl.live_bands = slice->live_bands;
        sl.tile_bh = vu.tile_bh;
        sl.margin_pct = slice->margin_pct;
        sl.order = slice->order;
        sl.order_mode = slice->order_mode;
        sl.split = (phase != 2 && slice->order && slice->order_mode == 1u) ? slice->split : nullptr;
        sl.work = slice->work;
        if (bwd_info && phase != 2 && lpt) sl.jobs = slice->jobs;
    }
    if (sl.zcut && (!sl.depth_keys_sorted && sl.nv)) sl.zcut = nullptr;
    if (phase != 0 && (!sl.done_bits || !sl.unsat_count || !sl.state || (phase == 2 && !sl.offsets_near)))
        return set_error(ctx, BH_ERR_INVALID_ARG, "launch_rasterize: sliced phase without its scratch");
    if (sl.feedback && !sl.cum) sl.feedback = nullptr;
    uint32_t nblocks = band_slots(u.num_tiles) * 8u;
    if (sl.order && sl.order_mode == 2u) nblocks = ((band_slots(u.num_tiles) + 7u) / 8u) * 64u;   // 8 bands x 8 x seg ranks
    if (sl.split) nblocks += 8u * 3u * SPLIT_MAX;   // three more blocks for each tile a band may split (blocks behind the band's last rank leave at once)
    const dim3 grid(nblocks);
    // resident waves per SIMD (option k16_waves): unused dynamic LDS caps how many one-wave blocks a CU holds (160 KB per CU, 4 SIMDs); the
    // blocks behind the first round are dispatched as slots free up, in block order — which with the order table is descending forecast work
    uint32_t lds_pad = 0;
    if (ctx->knob_k16_waves >= 1u && ctx->knob_k16_waves < 8u && sl.order && sl.order_mode == 1u && phase != 2) {
        const uint32_t per_block = (160u * 1024u) / (4u * ctx->knob_k16_waves);
        const uint32_t fixed = BATCH * SPLAT_STRIDE * 4u;
        lds_pad = per_block > fixed + 64u ? ((per_block - fixed - 64u) & ~63u) : 0u;
    }
    if (bwd_info && smooth) launch_rasterize_phase<true, true>(phase, grid, lds_pad, ctx->stream, u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible, lpt, sl);
    else if (bwd_info) launch_rasterize_phase<true, false>(phase, grid, lds_pad, ctx->stream, u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible, lpt, sl);
    else launch_rasterize_phase<false, false>(phase, grid, lds_pad, ctx->stream, u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible, lpt, sl);
    BH_LAUNCH_CHECK(ctx, "rasterize_kernel");
    return 0;
}

// ---------------------------------------------------------------------------
// K17: rasterize_backwards (bwd/kernels/rasterize_backwards.rs:101-390)
// ---------------------------------------------------------------------------
// Issue-rate facts this kernel is shaped by (scripts/micro/valu_rates.hip, valu_latency.hip): a SIMD issues one
// full-rate VALU op per ~2.6 cycles at best and needs >= 4 ready waves for that (one wave alone: an op every 6-10 cycles);
// v_cmp / v_cndmask / v_min and DPP ops issue at HALF rate, v_rcp / v_sqrt / v_exp / v_permlane32_swap at a QUARTER.  The
// kernel runs at ~90 % of the VALU issue bound of its instruction mix (SQ counters), so what counts is the number and the
// kind of VALU instructions per (splat, tile):
//   * the gradient block of a pixel-quadrant runs under an EXEC MASK (a real divergent region), not behind selects: lanes
//     that do not contribute do not execute it, and none of its results needs a v_cndmask to be thrown away;
//   * algebra instead of per-pixel work: the pixel's remaining colour is only ever used dotted with its v_rgb, so the state
//     is that one dot product; v_xy and v_conic are linear maps of sums of v_sigma * (pixel - mean) and its second
//     moments, so those RAW sums are accumulated and the 2x2 conic / the 1/2 factors are applied once per splat;
//     d(alpha0) needs no accumulator of its own: it is -(sum of v_sigma) / alpha0 (gradient block: 42 -> 32 instructions);
//   * splats whose alpha0 cannot reach the 0.999 clamp (all of a batch: decided by one ballot at staging time) run a
//     variant without the v_min / v_cmp / v_cndmask of the clamp;
//   * the ten per-splat sums over the tile's 256 pixels leave the wave through a register butterfly (below) and ONE 10-lane
//     global_atomic_add_f32.  It is the single most expensive piece left: a build without any reduction (wrong results) runs
//     232 us instead of 325 — 28 % of the kernel for ~45 instructions, because v_permlane*_swap and DPP adds are slow AND
//     form a dependent chain.  Three replacements were built, verified against the CPU checker, measured and dropped
//     (DESIGN.md §8): partials parked in LDS and summed by lane pairs behind a barrier (384 us), the same with the reads issued
//     one splat later so nobody waits (446 us: the LDS pipe is not idle enough for 6.5 KB more per splat and wave), and the
//     matrix core (v_mfma_f32_16x16x4_f32 with column selectors: exact, 542 us — the f32 MFMAs do not hide beside the VALU work).

// Register butterfly: wave-wide sum of ten per-lane values in 28 VALU ops — v_permlane32_swap /
// v_permlane16_swap fold two registers into one per step ("transpose-reduce"), then a DPP rotate-add finishes inside each
// 16-lane row.  Afterwards every lane of row r of k[i] holds component comp(i, r): k0 -> g0 g2 g1 g3, k1 -> g4 g6 g5 g7,
// k2 -> g8 - g9 -.
BH_DEV float swap32_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(f2u(a), f2u(b), false, false);
    return u2f(r[0]) + u2f(r[1]);
}
BH_DEV float swap16_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(f2u(a), f2u(b), false, false);
    return u2f(r[0]) + u2f(r[1]);
}
template <int CTRL>
BH_DEV float dpp_rot_add(float x) {
    return x + u2f(__builtin_amdgcn_update_dpp(0u, f2u(x), CTRL, 0xF, 0xF, false));
}
BH_DEV float row_allreduce(float x) {
    x = dpp_rot_add<0x128>(x);  // row_ror:8
    x = dpp_rot_add<0x124>(x);  // row_ror:4
    x = dpp_rot_add<0x122>(x);  // row_ror:2
    x = dpp_rot_add<0x121>(x);  // row_ror:1
    return x;
}

// exp() of the backward's replay is the forward's exp_blend, bit for bit: the replay has to take the forward's decisions
// (alpha >= 1/255, T' <= 1e-4) — with v_exp_f32 instead (BH_BWD_HW_EXP, measurement variant: 13 us faster) a pixel sitting
// within an ulp of a threshold takes the other branch, and at 1 M splats / 1080p the conic gradient of one splat came out
// 2.2e-4 * max|g| off the CPU checker (far pixels carry dx^2), beyond the 1e-4 the gradients are specified to.
BH_DEV float blend_exp_bwd(float x) {
    return exp_blend(x);
}

constexpr int BWD_WAVES = 5;   // waves per SIMD the default variant (hard cutoff, refine weight) is compiled for: 96 VGPRs; the variant without the refine weight gets 6 (80 VGPRs), the smooth-cutoff variants one fewer each.  (Rounds 2-5 compiled for 3 and the allocator happened to stop at 95 / 79; any change to the job prologue moved it to 99-109, i.e. to FOUR waves.)
// REFINE = false: the refine weight (…:340-349: a per-pixel norm — two fmas, a quarter-rate v_sqrt, an fma per contributing
// pixel-quadrant, a third of the gradient block — and the two per-pixel registers its 1/A factors live in) is left out: its one
// consumer, refine()'s growth selection, stops reading it at growth_stop_iter (train.rs:589-614), i.e. for the second half of a
// default training run.  bh_train_step selects it from BhTrainConfig.growth_stop_iter; bh_render_backward* always compute it.
// The nine other sums are bit-identical either way (same instructions in the same order).
// JOBS = true (round 6): the unit of work is a SEGMENT of BWD_SEG entries of a tile's blended list (context.h BwdJobs).  A block
// takes jobs j, j + blocks per band, ... of its XCD band in descending class order (full segments first, then the tiles' tails by
// length); a job of segment s > 0 starts from the checkpoint the forward blend left in front of the segment's first entry instead
// of replaying the tile from its first splat: the pixels' transmittance as it was there (bit for bit: the replay's decisions are
// the forward's), and the remaining colour = final colour - colour so far.  Gradients of the segments of a tile add up in the
// accumulator like those of different tiles.  Why: one wave per TILE made the launch last as long as its heaviest tile (a lone
// wave retires an op every ~5 cycles, one of five on a SIMD every ~13): 409 us for a frame whose heaviest tile blends 879 splats
// while the mean tile blends 58 (an object in front of an empty background), and 1.6 rounds of whole tiles on the uniform frame.
template <bool SMOOTH, bool REFINE, bool JOBS>
__global__ __launch_bounds__(64, (BWD_WAVES - (SMOOTH ? 1 : 0) + (REFINE ? 0 : 1))) void rasterize_backward_kernel(RasterUniforms u, const uint32_t* __restrict__ isect_gids,
                                                               const uint32_t* __restrict__ tile_offsets,
                                                               const float* __restrict__ projected,
                                                               const float* __restrict__ out_img,
                                                               const float* __restrict__ v_output,
                                                               float* __restrict__ v_combined, const uint32_t* __restrict__ lpt,
                                                               const uint32_t* __restrict__ tile_offsets_far, BwdJobs jb) {
    __shared__ __attribute__((aligned(16))) float s_splat[BATCH * SPLAT_STRIDE];
  for (uint32_t jidx = blockIdx.x >> 3;; jidx += gridDim.x >> 3) {   // (JOBS: a block's jobs; otherwise one tile, left by `return`)
    uint32_t local_tile, seg = 0u;
    bool to_end = true;
    if (lpt) {
        // block j of XCD x takes the j-th entry of band x in descending work-class order (wave-uniform scalar code)
        const uint32_t xcd = blockIdx.x & 7u;
        uint32_t j = jidx;
        const uint32_t* cnt = lpt + xcd * LPT_CLASSES;
        // Which class holds the band's j-th entry, in descending class order: the counters are fetched EIGHT at a time (independent scalar
        // loads, one wait).  Rounds 2-6 walked them with one dependent load per class: nothing for a job of the top class, but up to 64
        // round trips — 14-25 us under load (per-block trace of the launch, scripts/k17_trace.py) — for the short tails that sit in the
        // low classes and are taken LAST: the launch's own tail was made of waves looking for their job.  (One counter per lane and a
        // wave scan finds the class in one round trip, but its vector code cost the hot loop registers: 95 -> 109 VGPRs, measured slower.)
        uint32_t cls = LPT_CLASSES;
        bool found = false;
        static_assert(LPT_CLASSES % 8u == 0u, "classes are walked in groups of eight");
        for (uint32_t base = LPT_CLASSES; base > 0u && !found; base -= 8u) {   // (uniform)
            uint32_t k[8];
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) k[i] = cnt[base - 1u - i];
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) {
                if (!found) {
                    if (j < k[i]) { cls = base - 1u - i; found = true; }
                    else j -= k[i];
                }
            }
        }
        if (!found) return;  // no entry left in this band
        const uint32_t entry = (JOBS && cls == LPT_CLASSES - 1u) ? jb.top_list[(size_t)xcd * jb.top_cap + j] : lpt[lpt_list_offset(u.num_tiles, xcd, cls) + j];
        local_tile = entry & JOB_TILE_MASK;
        if (JOBS) { seg = (entry >> JOB_SEG_SHIFT) & JOB_SEG_MASK; to_end = (entry & JOB_LAST_BIT) != 0u; }
    } else {
        local_tile = tile_of_block(blockIdx.x, u.num_tiles, u.band_mode);
        if (local_tile >= u.num_tiles) return;
    }
    const uint32_t tile = u.tile_begin + local_tile;
    // the tile's blended splats, front to back: one list (the exact path), or the near slice's followed by the far slice's
    // (depth-sliced forward; the far table is all zero for a tile the near slice finished)
    uint32_t seg_lo0 = tile_offsets[tile * 2], seg_hi0 = tile_offsets[tile * 2 + 1];
    const uint32_t list_lo = seg_lo0;
    uint32_t seg_lo1 = 0, seg_hi1 = 0;
    if (!JOBS && tile_offsets_far) { seg_lo1 = tile_offsets_far[tile * 2]; seg_hi1 = tile_offsets_far[tile * 2 + 1]; }
    if (JOBS) {   // this job's part of the tile's list
        seg_lo0 += seg * BWD_SEG;
        if (!to_end) seg_hi0 = seg_lo0 + BWD_SEG;
        if (seg_hi0 <= seg_lo0) continue;
    } else if (seg_hi0 <= seg_lo0 && seg_hi1 <= seg_lo1) return;
    int lane = threadIdx.x;
    if (JOBS) asm volatile("" : "+v"(lane));   // (per job: nothing derived from the lane stays in a register across the job loop)
    const uint32_t tx0 = (tile % u.tile_bw) * TILE_WIDTH, ty0 = (tile / u.tile_bw) * TILE_WIDTH;
    const uint32_t px0 = tx0 + (lane & 7), py0 = ty0 + (lane >> 3);
    const float pcx[2] = {(float)px0 + 0.5f, (float)(px0 + 8) + 0.5f};
    const float pcy[2] = {(float)py0 + 0.5f, (float)(py0 + 8) + 0.5f};
    const float img_w_f = (float)u.img_w, img_h_f = (float)u.img_h;
    // pixel replay state (rasterize_backwards.rs:186-228).  The reference carries the remaining colour (rgb_final - T_final * bg
    // minus what has been replayed) and only ever uses it inside a dot product with the pixel's v_rgb, so the state kept here
    // is that dot product: S = remaining . v_rgb (one register and one fma per update instead of three).  T = 0: finished.
    float sS[4], sw[4];
    float vox[4], voy[4], voz[4], w2q[4], h2q[4];   // w2q / h2q: (W / A)^2, (H / A)^2 of the pixel — the refine weight's 1 / max(A, 1e-5) folded into its norm
    // (the eight pixel loads side by side, from clamped — always valid — addresses, masked afterwards: under `if (inside)` they
    //  were four dependent global round trips at the head of every tile)
    float4 o4[4], vo4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t px = px0 + 8 * (q & 1), py = py0 + 8 * (q >> 1);
        const uint32_t pxc = px < u.img_w ? px : u.img_w - 1u, pyc = py < u.img_h ? py : u.img_h - 1u;
        const size_t pix = ((size_t)pxc + (size_t)pyc * u.img_w) * 4;
        o4[q] = *reinterpret_cast<const float4*>(&out_img[pix]);
        vo4[q] = *reinterpret_cast<const float4*>(&v_output[pix]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t px = px0 + 8 * (q & 1), py = py0 + 8 * (q >> 1);
        if (px < u.img_w && py < u.img_h) {
            const float4 o = o4[q];
            const float4 vo = vo4[q];
            const float t_final = 1.0f - o.w;
            // ... minus the pixel's constant term v_o_w = (v_A - bg . v_rgb) T_final, which only ever appears added to it (…:300-310):
            // S' = remaining . v_rgb - v_o_w   (one register and one add per pixel-quadrant less)
            const float v_o_w = (vo.w - (u.bg_r * vo.x + u.bg_g * vo.y + u.bg_b * vo.z)) * t_final;
            sS[q] = __builtin_fmaf(o.z - t_final * u.bg_b, vo.z, __builtin_fmaf(o.y - t_final * u.bg_g, vo.y, (o.x - t_final * u.bg_r) * vo.x)) - v_o_w;
            sw[q] = 1.0f;
            vox[q] = vo.x; voy[q] = vo.y; voz[q] = vo.z;
            if (REFINE) {
                const float inv_fa = 1.0f / __builtin_fmaxf(o.w, 1.0e-5f);
                w2q[q] = (img_w_f * inv_fa) * (img_w_f * inv_fa);
                h2q[q] = (img_h_f * inv_fa) * (img_h_f * inv_fa);
            } else {
                w2q[q] = h2q[q] = 0.0f;
            }
        } else {
            sS[q] = sw[q] = 0.0f;
            vox[q] = voy[q] = voz[q] = 0.0f;
            w2q[q] = h2q[q] = 0.0f;
        }
    }
    if (JOBS && seg != 0u) {
        // start from the forward's checkpoint in front of this segment: (colour so far, signed transmittance) per pixel
        const float4* ck = jb.ckpt + (size_t)ckpt_slot(list_lo, tile, seg) * 256u + (uint32_t)lane;
        float4 c4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c4[q] = ck[q * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // remaining . v_rgb = (final - so far) . v_rgb: the tile's start value minus what the earlier segments subtracted
            sS[q] -= __builtin_fmaf(c4[q].z, voz[q], __builtin_fmaf(c4[q].y, voy[q], c4[q].x * vox[q]));
            sw[q] = (sw[q] != 0.0f && c4[q].w > 0.0f) ? c4[q].w : 0.0f;   // (a finished pixel carries its final T with the sign flipped; outside the image: 0)
        }
    }

    // The ten per-lane partial sums of the splat in flight, RAW: aP aQ = sums of v_sigma (pixel - mean), aR2 aR3 aR4 = its second
    // moments, aCr aCg aCb = rgb, aVs = sum of v_sigma, aRf = refine; the per-splat linear maps that turn them into the
    // reference's ten gradients are applied once per splat, just before the wave reduction.  They live ACROSS the splat loop
    // and are cleared only after a reduction: a splat that touches no pixel leaves them at zero, so the common "no contribution" path
    // carries no re-initialisation at all.
    float aP = 0.f, aQ = 0.f, aR2 = 0.f, aR3 = 0.f, aR4 = 0.f, aCr = 0.f, aCg = 0.f, aCb = 0.f, aVs = 0.f, aRf = 0.f;
    // One staged batch.  CLAMP = false: every alpha0 of the batch is <= 0.999, so min(0.999, alpha0 * G) is the identity and
    // the "below the clamp" gate of the geometry gradients (…:332) is always open — one v_min, one v_cmp and one v_cndmask
    // (all half-rate) less per pixel-quadrant.  Which variant runs is decided per BATCH (a ballot at staging time), not per
    // splat: a per-splat two-way branch makes every accumulator a phi of two paths inside the hot loop, and the copies
    // that resolves into cost more than the three instructions saved.
    auto run_batch = [&](auto clamp_tag, const uint32_t cnt) {
        constexpr bool CLAMP = decltype(clamp_tag)::value;
        for (uint32_t t = 0; t < cnt; ++t) {
            const float4 s0 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE]);      // x y c00/2 c01
            const float4 s1 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE + 4]);  // c11/2 a r g (clamped)
            const float2 s2 = *reinterpret_cast<const float2*>(&s_splat[t * SPLAT_STRIDE + 8]);  // b cut
            // (the diagonal is staged HALVED, as in the forward: sigma needs no multiply by 1/2 per pixel-quadrant, and the replay's
            //  arithmetic is the forward's instruction for instruction; the full diagonal terms the refine weight wants are 2 x these)
            const float h00 = s0.z, c01 = s0.w, h11 = s1.x, color_a = s1.y;
            const float cr = s1.z, cgc = s1.w, cb = s2.x;
            const uint32_t cut_bits = f2u(s2.y);
            float dxp[2], dyp[2], a_xx[2], b_x[2], c_y[2], e_x[2], e_y[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                dxp[k] = pcx[k] - s0.x;  // pixel - mean (forward convention)
                e_x[k] = h00 * dxp[k];          // (c00 / 2) dx
                a_xx[k] = e_x[k] * dxp[k];
                b_x[k] = c01 * dxp[k];
                dyp[k] = pcy[k] - s0.y;
                c_y[k] = h11 * dyp[k];          // (c11 / 2) dy
                e_y[k] = c01 * dyp[k];
            }
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = q & 1, m = q >> 1;
                // --- replay: identical arithmetic to the forward kernel -------------
                const float half_qv = __builtin_fmaf(c_y[m], dyp[m], a_xx[k]);
                const float sigma = __builtin_fmaf(b_x[k], dyp[m], half_qv);
                // live pixel and 0 <= sigma <= sigma_cut  (the forward's single-compare form of this test — sign bit of a negative
                // "finished" T ORed into the key — was measured here too: +1 %, four more VGPRs push the kernel over 96)
                const bool pre = sw[q] > 0.0f && f2u(sigma) <= cut_bits;
                if (__ballot(pre) != 0ull) {   // wave-uniform: one scalar branch steps over a quadrant that cannot reach the cutoff
                    const float gaussian = blend_exp_bwd(-sigma);
                    const float alpha_raw = color_a * gaussian;
                    const float alpha = CLAMP ? __builtin_fminf(0.999f, alpha_raw) : alpha_raw;
                    const float w_cut = SMOOTH ? alpha_cutoff_weight(alpha) : 1.0f;
                    const bool ok = pre && (SMOOTH ? (w_cut > 0.0f) : (alpha >= ALPHA_CUTOFF_MID));
                    const float alpha_eff = SMOOTH ? alpha * w_cut : alpha;
                    const float one_m = 1.0f - alpha_eff;
                    const float T = sw[q];
                    const float next_t = T * one_m;
                    const bool live = !(next_t <= 1.0e-4f);
                    sw[q] = (ok && !live) ? 0.0f : T;   // the pixel is done WITHOUT this splat (rasterize.rs:155-160)
                    if (ok && live) {                   // the ONE exec-masked region: lanes that do not contribute skip the gradient block
                        __asm__ volatile("" ::: "memory");
                        // --- gradients (rasterize_backwards.rs:286-381); tolerance-checked, so explicit fma / v_rcp /
                        //     v_sqrt and algebraically regrouped sums are used freely here -----------------------------
                        const float vis = alpha_eff * T;
                        aCr = __builtin_fmaf(vis, vox[q], aCr);
                        aCg = __builtin_fmaf(vis, voy[q], aCg);
                        aCb = __builtin_fmaf(vis, voz[q], aCb);
                        const float ra = __builtin_amdgcn_rcpf(one_m);
                        // (T c - remaining) . v_rgb = T (c . v_rgb) - S
                        const float cv = __builtin_fmaf(cb, voz[q], __builtin_fmaf(cgc, voy[q], cr * vox[q]));
                        const float v_alpha_eff = __builtin_fmaf(T, cv, -sS[q]) * ra;
                        const float v_alpha = SMOOTH ? v_alpha_eff * (w_cut + alpha * alpha_cutoff_weight_deriv(alpha)) : v_alpha_eff;
                        // geometry / opacity / refine grads only below the alpha clamp (…:332)
                        const float v_sigma = (!CLAMP || alpha_raw <= 0.999f) ? -alpha * v_alpha : 0.0f;
                        // sums of v_sigma * (pixel - mean) and of its second moments; mapped to v_xy / v_conic once per splat (below)
                        const float ux = v_sigma * dxp[k], uy = v_sigma * dyp[m];
                        aP += ux;
                        aQ += uy;
                        aR2 = __builtin_fmaf(ux, dxp[k], aR2);
                        aR3 = __builtin_fmaf(ux, dyp[m], aR3);
                        aR4 = __builtin_fmaf(uy, dyp[m], aR4);
                        aVs += v_sigma;
                        // refine weight: |(v_xy.x W, v_xy.y H)| / max(A, 1e-5), v_xy = -v_sigma conic (pixel - mean)   (…:340-349)
                        if (REFINE) {
                            const float ex = __builtin_fmaf(2.0f, e_x[k], e_y[m]), ey = __builtin_fmaf(2.0f, c_y[m], b_x[k]);   // conic (pixel - mean)
                            // (|(W vx, H vy)| / A = sqrt((W/A)^2 vx^2 + (H/A)^2 vy^2): the per-pixel 1/A lives in w2q / h2q)
                            const float n2 = __builtin_fmaf(h2q[q] * ey, ey, w2q[q] * (ex * ex));
                            aRf = __builtin_fmaf(__builtin_fabsf(v_sigma), __builtin_amdgcn_sqrtf(n2), aRf);
                        }
                        // --- state update ---------------------------------------------------------
                        sS[q] = __builtin_fmaf(-vis, cv, sS[q]);
                        sw[q] = next_t;
                        any = true;
                    }
                }
            }
            if (__ballot(any) != 0ull) {
                // The ten RAW sums leave as they are: the per-splat linear maps that turn them into the reference's ten gradients
                // (conic / 1/2 factors, colour gates, -1/alpha0: ~20 issue slots incl. a quarter-rate rcp) commute with the sum over
                // tiles, so K18 applies them ONCE per splat to the accumulated row instead of this kernel once per (splat, tile).
                const float h0 = swap32_add(aP, aQ), h1 = swap32_add(aR2, aR3), h2 = swap32_add(aR4, aCr);
                const float h3 = swap32_add(aCg, aCb), h4 = swap32_add(aVs, REFINE ? aRf : 0.0f);
                const float k0 = row_allreduce(swap16_add(h0, h1));
                const float k1 = row_allreduce(swap16_add(h2, h3));
                const float k2 = row_allreduce(swap16_add(h4, 0.0f));
                const int ri = lane & 15, rrow = lane >> 4;
                const int comp = ri * 4 + (((rrow & 1) << 1) | (rrow >> 1));
                const float mine = ri == 0 ? k0 : (ri == 1 ? k1 : k2);
                const uint32_t cg = f2u(s_splat[t * SPLAT_STRIDE + 11]);
                if (ri < 3 && comp < 10) unsafeAtomicAdd(&v_combined[(size_t)cg * 10 + comp], mine);
                aP = aQ = aR2 = aR3 = aR4 = aCr = aCg = aCb = aVs = aRf = 0.0f;
            }
        }
    };

#pragma nounroll
    for (int part = 0; part < (JOBS ? 1 : 2); ++part) {   // ONE copy of the batch loop for both segments (its body is ~1.6 k instructions)
        const uint32_t range_lo = part ? seg_lo1 : seg_lo0, range_hi = part ? seg_hi1 : seg_hi0;
        for (uint32_t batch_start = range_lo; batch_start < range_hi; batch_start += BATCH) {
            const uint32_t cnt = min((uint32_t)BATCH, range_hi - batch_start);
            __syncthreads();
            stage_batch<SMOOTH, true>(isect_gids, projected, batch_start, cnt, lane, s_splat);
            __syncthreads();
            const bool mine_clamps = (uint32_t)lane < cnt && s_splat[lane * SPLAT_STRIDE + 5] > 0.999f;
            if (__ballot(mine_clamps) != 0ull) run_batch(std::true_type{}, cnt);
            else run_batch(std::false_type{}, cnt);
        }
    }
    if (!JOBS) return;
  }
}

template <bool SMOOTH, bool REFINE, bool JOBS>
static void launch_rasterize_backward_t(hipStream_t stream, dim3 grid, hipEvent_t ea, hipEvent_t eb, const RasterUniforms& u, const uint32_t* isect_gids,
                                        const uint32_t* tile_offsets, const float* projected, const float* out_img, const float* v_output, float* v_combined,
                                        const uint32_t* lpt, const uint32_t* tile_offsets_far, const BwdJobs& jb) {
    const dim3 block(64);
    if (ea)   // profiling level 2 (bench.py's timed region): the launch carries its own start / stop events (context.h)
        hipExtLaunchKernelGGL((rasterize_backward_kernel<SMOOTH, REFINE, JOBS>), grid, block, 0, stream, ea, eb, 0, u, isect_gids, tile_offsets, projected, out_img, v_output, v_combined, lpt, tile_offsets_far, jb);
    else
        hipLaunchKernelGGL((rasterize_backward_kernel<SMOOTH, REFINE, JOBS>), grid, block, 0, stream, u, isect_gids, tile_offsets, projected, out_img, v_output, v_combined, lpt, tile_offsets_far, jb);
}

int launch_rasterize_backward(bh_ctx* ctx, const ViewUniforms& vu, const float bg[3], bool smooth,
                              const uint32_t* isect_gids, const uint32_t* tile_offsets, const float* projected,
                              const float* out_img, const float* v_output, float* v_combined, const uint32_t* lpt,
                              const uint32_t* tile_offsets_far, bool want_refine, const BwdJobs* jobs) {
    RasterUniforms u;
    u.band_mode = ctx->knob_band_mode;
    u.rcp_class_width = 1.0f;
    u.tile_bw = vu.tile_bw;
    u.num_tiles = vu.tile_bw * (vu.tile_y1 - vu.tile_y0);
    u.tile_begin = vu.tile_bw * vu.tile_y0;
    u.img_w = vu.img_w;
    u.img_h = vu.img_h;
    u.bg_r = bg[0]; u.bg_g = bg[1]; u.bg_b = bg[2];
    // the forward filed jobs (segments with checkpoints) or whole tiles: the backward reads the lists the way they were written
    // (jobs are only filed by frames whose far segments are empty: complete lists and per-tile cuts, whose far table is all zero — api.hip)
    const bool by_jobs = jobs && jobs->ckpt && lpt;
    const BwdJobs jb = by_jobs ? *jobs : BwdJobs{};
    // whole tiles: one block per tile.  Jobs: two blocks per tile (a typical frame has 1.5 - 2.5 jobs per tile); a block takes
    // every (blocks per band)-th job of its band, so any number of jobs is covered
    const uint32_t per = band_slots(u.num_tiles);
    const uint32_t nblocks = (by_jobs ? 2u : 1u) * per * 8u;
    const dim3 grid(nblocks);
    hipEvent_t ea = ctx->prof.ext_a, eb = ctx->prof.ext_b;
    ctx->prof.ext_a = ctx->prof.ext_b = nullptr;
#define BH_K17(S, R, J) launch_rasterize_backward_t<S, R, J>(ctx->stream, grid, ea, eb, u, isect_gids, tile_offsets, projected, out_img, v_output, v_combined, lpt, tile_offsets_far, jb)
    if (by_jobs) {
        if (smooth && want_refine) BH_K17(true, true, true);
        else if (smooth) BH_K17(true, false, true);
        else if (want_refine) BH_K17(false, true, true);
        else BH_K17(false, false, true);
    } else {
        if (smooth && want_refine) BH_K17(true, true, false);
        else if (smooth) BH_K17(true, false, false);
        else if (want_refine) BH_K17(false, true, false);
        else BH_K17(false, false, false);
    }
#undef BH_K17
    BH_LAUNCH_CHECK(ctx, "rasterize_backward_kernel");
    return 0;
}

}  // namespace bh

