// project.hip — per-splat kernels: project_forward (cull + count tiles),
// project_visible (projected records + SH colour), map_gaussians_to_intersect
// (emit (tile, splat) pairs) and project_backward (chain rasterizer grads back
// to means / quats / log-scales / SH / opacity).
//
// Reference: brush-render/src/kernels/{project_forward,project_visible,map_gaussians}.rs,
// brush-render/src/bwd/kernels/project_backwards.rs (paths under /root/reference/crates).
//
// MI355X notes: these are HBM-bound one-thread-per-splat kernels.  Where the
// reference appends visible splats through a global atomic slot counter
// (project_forward.rs:122-124, nondeterministic order), this build writes a
// per-splat depth key (0xFFFFFFFF when culled) in place and lets the stable depth
// sort do the compaction: no per-splat atomics, deterministic tie order (by id).
#include "context.h"
#include "device_camera.h"
#include "device_sh.h"

namespace bh {

constexpr int PROJ_WG = 256;
constexpr int PROJ_WAVES = PROJ_WG / 64;
constexpr uint32_t ORDER_BINS = 1024;   // the tile-order sort's bins: per-tile work (splats blended) clamped to 1023

// ---------------------------------------------------------------------------
// Load-balanced tile walk shared by K1 (count) and K5 (emit).
//
// The reference walks each splat's tile bounding box with one thread
// (helpers.rs:204-223, map_gaussians.rs:46-72): a wave then runs as long as its
// largest splat.  Here a wave flattens the boxes of its 64 splats into one
// candidate list (wave prefix sum of the box areas), and lane l tests candidates
// l, l+64, ...: every lane is busy whatever the size distribution.
//
// Which splat owns candidate c?  The splats with a non-empty box are compacted to
// ranks 0..K-1 (their parameters are stored in LDS by RANK); rank r owns the
// candidates [start_r, start_r + area_r), and the starts are strictly increasing.
// One step covers the 64 consecutive candidates base..base+63, so
//     owner(base + l) = A + popcount(marks & lanes<=l) - 1
// with A = number of ranks that start before `base` (a running scalar) and `marks`
// the 64-bit mask of the starts that fall inside this step — built by the owning
// lanes dropping one byte each into a 64-byte LDS strip that the wave reads back
// with a ballot.  That is ONE LDS round trip per step where a binary search over
// the prefix sums took six dependent ones (the walk is latency-, not ALU-bound).
// The per-(splat, tile) test is the same inlined will_primitive_contribute for both
// kernels, so count and emit cannot disagree.
// ---------------------------------------------------------------------------
struct WalkLds {
    uint32_t count[64];   // by rank: hits per splat (K1) / emit cursor (K5)
    float mx[64], my[64], c00[64], c01[64], c11[64], pt[64];   // by rank
    uint32_t magic[64];   // by rank: ceil(2^32 / box width), 0 = divide the long way (walk_row)
    uint32_t box[64];     // by rank: min_x | min_y << 16   (tile grids up to 4095 x 4095: api.hip's image-size limit)
    uint32_t boxw[64];    // by rank: box width in tiles
    uint32_t start[64];   // by rank: first candidate of the splat
    uint32_t lane_of[64]; // by rank: the lane (splat slot) it came from
    uint32_t zkey[64];    // by rank: the splat's depth key (per-tile depth cuts: near / far is decided per (splat, tile))
    uint32_t near[64];    // by rank: hits at or in front of their tile's cut (K1)
    uint8_t flags[64];    // scratch strip for the per-step start marks
};

// i / bw for a candidate index inside a box.  The walk does this once per candidate, so it is a multiplication: with
// m = ceil(2^32 / bw), mul_hi(i, m) = floor(i / bw) exactly as long as i * bw < 2^32 (e = m bw - 2^32 < bw and i e < 2^32 is the
// classic condition) — walk_magic returns m when every index of the box satisfies that (any box of a 4K frame does by orders
// of magnitude), else 0 and walk_row divides.  (Before: a float quotient + two exec-masked corrections, ~15 VALU ops and three
// SALU exec swaps per candidate.)
BH_DEV uint32_t walk_magic(uint32_t bw, uint32_t nb) {
    if (bw < 2u || (unsigned long long)nb * bw >= (1ull << 32)) return 0u;
    return 0xFFFFFFFFu / bw + 1u;
}
BH_DEV uint32_t walk_row(uint32_t i, uint32_t bw, uint32_t magic) {
    if (magic) return __umulhi(i, magic);
    if (bw < 2u) return i;
    // the float quotient is at most one off for anything a 4095 x 4095 tile grid can hold (i < 2^24); one correction each way
    uint32_t row = (uint32_t)(((float)i + 0.5f) / (float)bw);
    if (row * bw > i) --row;
    else if ((row + 1u) * bw <= i) ++row;
    return row;
}

BH_DEV uint32_t wave_inclusive_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// Visits every (splat, tile) candidate of the wave; calls on_hit(rank, tile_x, tile_y) for the
// contributing ones.  `nb` = box area of this lane's splat (0 = none).  Returns this lane's rank
// (valid when nb > 0): per-splat results are read back from w.count[rank].
// keep(tx, ty) is evaluated BEFORE the contribution test: the depth-sliced forward's second slice walks only the tiles that are
// still unsaturated (a bit test against a 1 KB table instead of the ellipse-rectangle test); K1 / the exact path keep everything.
struct KeepAllTiles { static constexpr bool EAGER = false; BH_DEV bool operator()(const WalkLds&, uint32_t, uint32_t, uint32_t) const { return true; } };
struct KeepLiveTiles {   // bit (tile) of done_bits set = the tile's pixels are final
    static constexpr bool EAGER = false;
    const uint32_t* done_bits;
    uint32_t tile_bw;
    BH_DEV bool operator()(const WalkLds&, uint32_t, uint32_t tx, uint32_t ty) const {
        const uint32_t t = tx + ty * tile_bw;
        return ((done_bits[t >> 5] >> (t & 31u)) & 1u) == 0u;
    }
};
// Per-tile depth cuts (api.hip): the list of a tile holds the splats whose depth key is <= the tile's cut
struct KeepNearOfCut {
    const uint32_t* zcut;
    uint32_t tile_bw;
    // EAGER (flat_tile_walk_emit): the cut is fetched, the contribution test runs WHILE the load is in flight, then the two are combined —
    // instead of load, wait, compare, branch, test.  The near slice's K5 is bound by its longest wave's chain of walk steps, not by issue
    // slots: the tests of the candidates behind their cuts (two thirds of them) cost nothing that shows.
    static constexpr bool EAGER = true;
    BH_DEV uint32_t fetch(uint32_t tx, uint32_t ty) const { return zcut[tx + ty * tile_bw]; }
    BH_DEV bool decide(const WalkLds& w, uint32_t r, uint32_t cut) const { return zcut_near(w.zkey[r], cut); }
    BH_DEV bool operator()(const WalkLds& w, uint32_t r, uint32_t tx, uint32_t ty) const { return decide(w, r, fetch(tx, ty)); }
};

// (One LDS atomic per hit lane in the counting callers: a variant in which the first lane of each splat's run of candidates adds
// the run's hits — ballot + popcount, one plain LDS update per splat and step — measured the same, 58.1 vs 57.6 us: the walk is
// bound by the ~60 VALU instructions of the test, two IEEE divisions among them, not by the LDS.)
// pre(tx, ty) is evaluated for every candidate BEFORE the contribution test and its value handed to on_hit: a load issued there (K1's
// read of the tile's depth cut) is in flight during the ~100 instructions of the test instead of a dependent round trip behind it.
struct NoPrefetch { BH_DEV uint32_t operator()(uint32_t, uint32_t) const { return 0u; } };
template <class OnHit, class Keep = KeepAllTiles, class Pre = NoPrefetch>
BH_DEV uint32_t flat_tile_walk(WalkLds& w, int lane, uint32_t nb, float mx, float my, Sym2 conic, float pt, TileBbox bb,
                               OnHit on_hit, Keep keep = Keep{}, uint32_t zkey = 0u, Pre pre = Pre{}) {
    const uint32_t incl = wave_inclusive_scan_u32(nb, lane);
    const uint32_t start = incl - nb;
    const bool nz = nb > 0u;
    const unsigned long long nzmask = __ballot(nz);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t rank = (uint32_t)__popcll(nzmask & below);
    const uint32_t bb_w = bb.max_x - bb.min_x;
    if (nz) {
        w.count[rank] = 0;
        w.near[rank] = 0;
        w.zkey[rank] = zkey;
        w.mx[rank] = mx; w.my[rank] = my;
        w.c00[rank] = conic.c00; w.c01[rank] = conic.c01; w.c11[rank] = conic.c11;
        w.pt[rank] = pt;
        w.magic[rank] = walk_magic(bb_w, nb);
        w.box[rank] = bb.min_x | (bb.min_y << 16);
        w.boxw[rank] = bb_w;
        w.start[rank] = start;
        w.lane_of[rank] = (uint32_t)lane;
    }
    // The lanes of one wave talk to each other through LDS below.  In hardware a wave's LDS operations retire in
    // order, so no s_barrier is needed — but the COMPILER reasons per lane and would forward a lane's own store to
    // its later load: compiler-only memory barriers (no instructions) keep every cross-lane read a real ds_read.
    // (NOT `volatile`: a volatile pointer into the struct loses the LDS address space — the strip was being accessed
    // with flat_store_byte / flat_load_ubyte sc0 sc1, each followed by s_waitcnt vmcnt(0), i.e. every step of the walk
    // also waited for its own global stores to land.)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    uint8_t* flags = w.flags;
    const uint32_t tot = __shfl(incl, 63);
    const unsigned long long le = below | (1ull << lane);
    uint32_t before = 0;  // ranks that start before `base` (wave-uniform)
    for (uint32_t base = 0; base < tot; base += 64) {
        // marks: which of the 64 candidates of this step is the first tile of a splat
        flags[lane] = 0;
        const uint32_t rel = start - base;   // wraps for starts before base -> fails the range test
        if (nz && rel < 64u) flags[rel] = 1;
        asm volatile("" ::: "memory");
        const unsigned long long marks = __ballot(flags[lane] != 0);
        asm volatile("" ::: "memory");
        const uint32_t c = base + (uint32_t)lane;
        if (c < tot) {
            const uint32_t r = before + (uint32_t)__popcll(marks & le) - 1u;
            const uint32_t i = c - w.start[r];
            const uint32_t box = w.box[r];
            const uint32_t bw = w.boxw[r];
            const uint32_t row = walk_row(i, bw, w.magic[r]);
            const uint32_t tx = (box & 0xFFFFu) + (i - __umul24(row, bw));   // row * bw <= i < 2^24
            const uint32_t ty = (box >> 16) + row;
            const uint32_t fetched = pre(tx, ty);
            if (keep(w, r, tx, ty) && will_primitive_contribute(tx, ty, w.mx[r], w.my[r], Sym2{w.c00[r], w.c01[r], w.c11[r]}, w.pt[r])) on_hit(r, tx, ty, fetched);
        }
        before += (uint32_t)__popcll(marks);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the callers read w.count[] of other lanes' hits next
    return rank;
}

// The same walk for K5, emitting instead of counting.  The wave's 64 splats own ONE contiguous range of the output
// (their slot ranges are consecutive: cum_tiles_hit is the prefix sum in exactly this order), and the walk visits the
// candidates in (splat, tile) order — so the n-th hit of the wave simply goes to slot wave_base + n.  A ballot prefix
// replaces the per-splat LDS cursor (an atomic with return per hit) and compacts the stores: lanes with a hit write
// consecutive addresses.  Relies on K1 having counted with the same inlined test (it has: will_primitive_contribute).
template <class Keep = KeepAllTiles>
BH_DEV uint32_t flat_tile_walk_emit(WalkLds& w, int lane, uint32_t nb, float mx, float my, Sym2 conic, float pt, TileBbox bb,
                                    uint32_t wave_base, uint32_t wave_total, uint32_t tile_bw, uint32_t cg0, uint32_t* __restrict__ tile_ids,
                                    uint32_t* __restrict__ isect_gids, Keep keep = Keep{}, uint32_t zkey = 0u) {
    const uint32_t incl = wave_inclusive_scan_u32(nb, lane);
    const uint32_t start = incl - nb;
    const bool nz = nb > 0u;
    const unsigned long long nzmask = __ballot(nz);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t rank = (uint32_t)__popcll(nzmask & below);
    const uint32_t bb_w = bb.max_x - bb.min_x;
    if (nz) {
        w.zkey[rank] = zkey;
        w.mx[rank] = mx; w.my[rank] = my;
        w.c00[rank] = conic.c00; w.c01[rank] = conic.c01; w.c11[rank] = conic.c11;
        w.pt[rank] = pt;
        w.magic[rank] = walk_magic(bb_w, nb);
        w.box[rank] = bb.min_x | (bb.min_y << 16);
        w.boxw[rank] = bb_w;
        w.start[rank] = start;
        w.lane_of[rank] = (uint32_t)lane;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // see flat_tile_walk
    uint8_t* flags = w.flags;
    const uint32_t tot = __shfl(incl, 63);
    const unsigned long long le = below | (1ull << lane);
    uint32_t before = 0;   // ranks that start before `base` (wave-uniform)
    uint32_t emitted = 0;  // hits of this wave so far (wave-uniform)
    for (uint32_t base = 0; base < tot; base += 64) {
        flags[lane] = 0;
        const uint32_t rel = start - base;
        if (nz && rel < 64u) flags[rel] = 1;
        asm volatile("" ::: "memory");
        const unsigned long long marks = __ballot(flags[lane] != 0);
        asm volatile("" ::: "memory");
        const uint32_t c = base + (uint32_t)lane;
        bool hit = false;
        uint32_t tile = 0, owner = 0;
        if (c < tot) {
            const uint32_t r = before + (uint32_t)__popcll(marks & le) - 1u;
            const uint32_t i = c - w.start[r];
            const uint32_t box = w.box[r];
            const uint32_t bw = w.boxw[r];
            const uint32_t row = walk_row(i, bw, w.magic[r]);
            const uint32_t tx = (box & 0xFFFFu) + (i - __umul24(row, bw));   // row * bw <= i < 2^24
            const uint32_t ty = (box >> 16) + row;
            if constexpr (Keep::EAGER) {
                const uint32_t cut = keep.fetch(tx, ty);
                uint32_t contributes = will_primitive_contribute(tx, ty, w.mx[r], w.my[r], Sym2{w.c00[r], w.c01[r], w.c11[r]}, w.pt[r]) ? 1u : 0u;
                asm volatile("" : "+v"(contributes));   // (the test is not to be sunk behind the load's wait)
                hit = contributes != 0u && keep.decide(w, r, cut);
            } else {
                hit = keep(w, r, tx, ty) && will_primitive_contribute(tx, ty, w.mx[r], w.my[r], Sym2{w.c00[r], w.c01[r], w.c11[r]}, w.pt[r]);
            }
            tile = tx + ty * tile_bw;
            owner = cg0 + w.lane_of[r];
        }
        const unsigned long long hits = __ballot(hit);
        if (hit) {
            // never outside the wave's own slot range [wave_base, wave_base + wave_total), whatever K1 counted
            const uint32_t local = emitted + (uint32_t)__popcll(hits & below);
            if (local < wave_total) {
                tile_ids[wave_base + local] = tile;
                isect_gids[wave_base + local] = owner;
            }
        }
        emitted += (uint32_t)__popcll(hits);
        before += (uint32_t)__popcll(marks);
    }
    // map_gaussians.rs:73-79: leftover budget becomes sentinel rows (tile id 0xFFFFFFFF sorts behind every tile and is
    // skipped by tile_offsets_kernel).  Dead code as long as K1's count and this walk agree — which they do: both inline the
    // same will_primitive_contribute on the same stored values — but a drift would otherwise leave uninitialised rows.
    for (uint32_t k = emitted + (uint32_t)lane; k < wave_total; k += 64u) {
        tile_ids[wave_base + k] = 0xFFFFFFFFu;
        isect_gids[wave_base + k] = 0u;
    }
    return emitted;
}

// ---------------------------------------------------------------------------
// K1: project_forward  (kernels/project_forward.rs:22-125)
// ---------------------------------------------------------------------------
// The projected record of a visible splat (project_visible.rs:23-88: xy, conic, alpha, colour) is computed
// here too and stored at its splat id: every term but the colour is needed for the cull anyway, so the
// reference's second per-splat pass (K4: re-read transforms / opacity / SH through the depth permutation —
// three random gathers — and redo the projection) shrinks to a 36-byte row gather.
template <bool MIP, bool PINHOLE, int DEG>
__global__ __launch_bounds__(PROJ_WG) void project_forward_kernel(
    ViewUniforms u, uint32_t n, const float* __restrict__ transforms, const float* __restrict__ coeffs,
    const float* __restrict__ raw_opacities, uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ isect_counts,
    float* __restrict__ max_radius, float* __restrict__ projected_by_gid, unsigned long long* __restrict__ counters, ForwardPrep prep,
    uint32_t* __restrict__ zcut, uint32_t* __restrict__ near_counts) {
    __shared__ WalkLds s_walk[PROJ_WAVES];
    __shared__ uint32_t s_vis[PROJ_WAVES];
    __shared__ uint32_t s_hit[PROJ_WAVES];
    __shared__ uint32_t s_near[PROJ_WAVES];
    __shared__ uint32_t s_listed[PROJ_WAVES];
    __shared__ uint32_t s_kmax[PROJ_WAVES];
    __shared__ uint32_t s_nmax[PROJ_WAVES];
    const uint32_t gid = blockIdx.x * PROJ_WG + threadIdx.x;
    // The forward blend keeps all its one-wave tiles resident at once (8160 tiles on 8192 wave slots at 1080p): its duration is the
    // SIMD whose eight tiles sum to the most work.  With the view's per-tile work of its last frame as the forecast, blocks
    // 0..7 sort the tiles of XCD band b by descending work (a counting sort in LDS) — consecutive blocks of a band then take
    // tiles of steadily decreasing work, so whatever regular pattern the dispatcher deals blocks to SIMDs with, every SIMD's
    // eight tiles are a stratified sample instead of eight neighbours.  8 of ~4000 blocks spend a few microseconds on it.
    if (prep.order_out && blockIdx.x < 8u) {   // block-uniform
        __shared__ uint32_t s_bins[ORDER_BINS];
        // (slot i of band b names local tile band_tile(b, i): context.h XCD BANDS; a slot behind the window's last tile names none)
        const uint32_t per = band_slots(prep.order_tiles);
        __shared__ uint32_t s_band_work, s_band_cnt, s_band_tail, s_band_max;
        for (uint32_t i = threadIdx.x; i < ORDER_BINS; i += PROJ_WG) s_bins[i] = 0u;
        if (threadIdx.x == 0) s_band_work = s_band_cnt = s_band_tail = s_band_max = 0u;
        __syncthreads();
        uint32_t my_work = 0u, my_cnt = 0u, my_max = 0u;
        for (uint32_t i = threadIdx.x; i < per; i += PROJ_WG) {
            const uint32_t lt = band_tile(blockIdx.x, i, per, prep.band_mode);
            if (lt >= prep.order_tiles) continue;
            const uint32_t wk = prep.order_work[prep.order_tile_begin + lt];
            atomicAdd(&s_bins[ORDER_BINS - 1u - (wk < ORDER_BINS ? wk : ORDER_BINS - 1u)], 1u);   // bin 0 = the most work
            my_work += wk < ORDER_BINS ? wk : ORDER_BINS - 1u;
            my_max = max(my_max, wk < ORDER_BINS ? wk : ORDER_BINS - 1u);
            my_cnt++;
        }
        if (prep.split_out && my_work) atomicAdd(&s_band_work, my_work);
        if (my_cnt) atomicAdd(&s_band_cnt, my_cnt);
        if (prep.split_out && my_max) atomicMax(&s_band_max, my_max);
        __syncthreads();
        const uint32_t cnt = s_band_cnt;
        // exclusive scan of the bins (one thread per 4 bins + a wave scan over the 256 partial sums)
        {
            __shared__ uint32_t s_part[PROJ_WG];
            const uint32_t b0 = threadIdx.x * (ORDER_BINS / PROJ_WG);
            uint32_t loc[ORDER_BINS / PROJ_WG], sum = 0;
#pragma unroll
            for (uint32_t k = 0; k < ORDER_BINS / PROJ_WG; ++k) { loc[k] = sum; sum += s_bins[b0 + k]; }
            s_part[threadIdx.x] = sum;
            __syncthreads();
            if (threadIdx.x == 0) {   // 256 partial sums: a serial pass is ~1 us
                uint32_t run = 0;
                for (uint32_t k = 0; k < (uint32_t)PROJ_WG; ++k) { const uint32_t t = s_part[k]; s_part[k] = run; run += t; }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t k = 0; k < ORDER_BINS / PROJ_WG; ++k) s_bins[b0 + k] = s_part[threadIdx.x] + loc[k];
        }
        __syncthreads();
        // split tiles (context.h SPLIT_MAX): the band's leading ranks whose forecast work is several times the band's mean are blended
        // by four quadrant waves each.  s_bins[b] = tiles with MORE work than bin b's: the start of the bin of `thr - 1` counts the
        // tiles with work >= thr.  The band's scratch rows are cleared here, by the block that decides.
        if (prep.split_out) {
            if (threadIdx.x == 0) {
                const float mean = cnt ? (float)s_band_work / (float)cnt : 0.0f;
                // (relative to the band's heaviest tile as well: a quadrant wave walks an entry in ~0.19 us where a whole-tile wave takes ~0.44 beside
                //  it, so tiles below ~0.45 x the maximum finish before the split maximum does — splitting them only costs instructions)
                const float thr_f = fmaxf(fmaxf((float)prep.split_min, prep.split_factor * mean), prep.split_of_max * (float)s_band_max);
                uint32_t h = 0u;
                if (prep.order_mode == 1u && thr_f < (float)(ORDER_BINS - 1u)) {
                    const uint32_t thr = (uint32_t)ceilf(thr_f);
                    h = s_bins[ORDER_BINS - thr];
                }
                prep.split_out[blockIdx.x] = h < SPLIT_MAX ? h : SPLIT_MAX;
            }
            uint32_t* scr = prep.split_out + 8u + blockIdx.x * (SPLIT_MAX * 4u);
            for (uint32_t i = threadIdx.x; i < SPLIT_MAX * 4u; i += PROJ_WG) scr[i] = 0u;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < per; i += PROJ_WG) {
            const uint32_t lt = band_tile(blockIdx.x, i, per, prep.band_mode);
            if (lt < prep.order_tiles) {
                const uint32_t wk = prep.order_work[prep.order_tile_begin + lt];
                const uint32_t rank = atomicAdd(&s_bins[ORDER_BINS - 1u - (wk < ORDER_BINS ? wk : ORDER_BINS - 1u)], 1u);
                prep.order_out[blockIdx.x * per + rank] = lt;
            } else {
                prep.order_out[blockIdx.x * per + cnt + atomicAdd(&s_band_tail, 1u)] = 0xFFFFFFFFu;   // (slots without a tile: the ranks behind the band's last tile stay empty)
            }
        }
        __syncthreads();
    }
    // housekeeping for the kernels behind this one (coalesced stores; nobody reads these buffers before K1 retires)
    if (gid < prep.visible_words) prep.visible[gid] = 0u;
    if (gid < prep.tile_words) prep.tile_table[gid] = 0u;
    if (gid < prep.slice_words) prep.slice_table[gid] = 0u;
    if (gid < COUNTER_SET_U64 && prep.next_counters) prep.next_counters[gid] = 0ull;
    for (size_t i = gid; i < prep.span_f4; i += (size_t)gridDim.x * PROJ_WG) prep.span[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t key = 0xFFFFFFFFu;
    float radius = 0.0f;
    bool visible = false;
    float mx = 0.0f, my = 0.0f, pt = 0.0f, opac = 0.0f;
    Sym2 conic = Sym2{0.0f, 0.0f, 0.0f};
    TileBbox bb = TileBbox{0, 0, 0, 0};
    Vec3A mean = v3(0.0f, 0.0f, 0.0f);
    // Every input of the splat is fetched up front, unconditionally (a clamped row for the grid's tail): behind the cull tests the
    // loads were SEVEN dependent global round trips (mean -> test -> scale x -> scale y -> scale z -> quaternion -> opacity -> SH),
    // one wait each.  The 14 % of splats that are culled early read 44 bytes they would not have needed.
    const uint32_t gsafe = gid < n ? gid : (n ? n - 1u : 0u);
    float tr[10];
    {
        const float* trp = transforms + (size_t)gsafe * 10;
#pragma unroll
        for (int k = 0; k < 10; ++k) tr[k] = trp[k];
    }
    float raw_opac = raw_opacities[gsafe];
    constexpr int C_ALL = (DEG + 1) * (DEG + 1);
    // the SH DC term of every splat; the higher bands (up to 72 more floats) only for the visible ones, below
    float sh_dc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) sh_dc[k] = coeffs[(size_t)gsafe * C_ALL * 3 + k];
    // (an empty asm that "uses" all fourteen values: without it the compiler sinks the later loads back behind the first cull test)
    asm volatile("" : "+v"(tr[0]), "+v"(tr[1]), "+v"(tr[2]), "+v"(tr[3]), "+v"(tr[4]), "+v"(tr[5]), "+v"(tr[6]), "+v"(tr[7]), "+v"(tr[8]), "+v"(tr[9]),
                      "+v"(raw_opac), "+v"(sh_dc[0]), "+v"(sh_dc[1]), "+v"(sh_dc[2]));
    if (gid < n) {
        do {
            mean = v3(tr[0], tr[1], tr[2]);
            const Vec3A mean_c = world_to_cam(mean, u);
            if (!(finite3(mean_c) && mean_c.z <= 1.0e10f)) break;
            if (!in_front_of_camera<PINHOLE>(mean_c, u)) break;  // project_forward.rs:47-61
            const Vec3A scl = v3(bh_expf(tr[7]), bh_expf(tr[8]), bh_expf(tr[9]));
            if (!finite3(scl)) break;
            const Quat qu = Quat{tr[3], tr[4], tr[5], tr[6]};
            const float qn = qdot(qu, qu);
            if (!(qn >= 1.0e-6f && is_finite_f32(qn))) break;
            if (!is_finite_f32(raw_opac)) break;
            const Quat q = qnormalize(qu);
            const Sym2 raw_cov = calc_cov2d<PINHOLE>(scl, q, mean_c, u);
            float filter_comp;
            const Sym2 cov = compensate_cov2d<MIP>(raw_cov, filter_comp);
            opac = sigmoid(raw_opac) * filter_comp;
            if (!sym2_finite(cov)) break;
            project_point<PINHOLE>(mean_c, u, mx, my);
            if (!(opac >= 1.0f / 255.0f)) break;
            pt = bh_logf(opac * 255.0f);
            conic = sym2_inverse(cov);
            float ex, ey;
            compute_bbox_extent(conic, pt, ex, ey);
            if (!(ex >= 0.0f && ey >= 0.0f)) break;
            const float wf = (float)u.img_w, hf = (float)u.img_h;
            const bool on_screen = mx + ex > 0.0f && mx - ex < wf && my + ey > 0.0f && my - ey < hf;
            if (!on_screen) break;
            bb = get_tile_bbox(mx, my, ex, ey, u.tile_bw, u.tile_y0, u.tile_y1);
            // strip render: a splat that misses every tile row of the window is not this rank's
            const bool whole = u.tile_y0 == 0u && u.tile_y1 == u.tile_bh;
            if (!whole && (bb.max_y <= bb.min_y || bb.max_x <= bb.min_x)) break;
            radius = __builtin_fmaxf(ex / wf, ey / hf);
            key = f2u(mean_c.z);
            visible = true;
        } while (false);
    }
    // helpers.rs:204-223 count_contributing_tiles, load-balanced over the wave
    const uint32_t nb = visible ? (bb.max_y - bb.min_y) * (bb.max_x - bb.min_x) : 0u;
    WalkLds& w = s_walk[wave];
    // per-tile depth cuts (zcut != NULL, wave-uniform): a hit also counts for the NEAR list if the splat is at or in front of its
    // tile's cut - K5 lists exactly those pairs (same keys, same table, same test).  A hit BEHIND the cut sets bit 0 of the tile's
    // entry — the bit is not part of the cut, and the other 31 bits do not change while K1 runs, so every writer stores the SAME
    // word: a plain store (an atomic OR here cost 36 us: the value just loaded comes from a CU's L1, stays stale there, and
    // every later hit of the tile repeated the atomic).  The blend kernel then knows whose near list is incomplete — an
    // unmarked tile holds everything there is, however its cut reads.
    const uint32_t tile_bw = u.tile_bw;
    struct CutOfTile {   // (zcut == NULL: no load)
        const uint32_t* zcut;
        uint32_t tile_bw;
        BH_DEV uint32_t operator()(uint32_t tx, uint32_t ty) const { return zcut ? zcut[tx + ty * tile_bw] : 0u; }
    };
    const uint32_t wrank = flat_tile_walk(w, lane, nb, mx, my, conic, pt, bb, [&](uint32_t r, uint32_t tx, uint32_t ty, uint32_t cut) {
        atomicAdd(&w.count[r], 1u);
        if (zcut) {
            if (zcut_near(w.zkey[r], cut)) atomicAdd(&w.near[r], 1u);
            else if ((cut & 1u) == 0u) zcut[tx + ty * tile_bw] = cut | 1u;
        }
    }, KeepAllTiles{}, key, CutOfTile{zcut, tile_bw});
    const uint32_t tiles_hit = nb ? w.count[wrank] : 0u;
    const uint32_t near_hit = (nb && zcut) ? w.near[wrank] : 0u;
    // per-tile cuts: a visible splat without a single pair in front of a cut takes no part in this frame's lists — it gets the
    // culled key, so the depth sort (whose stable order IS the compaction) numbers and orders only the splats that own a near
    // pair: a quarter of the visible ones at the bench workload.  It still counts as visible (the reference's num_visible).
    const bool listed = visible && (!zcut || near_hit > 0u || prep.list_all_visible);
    // The splat's colour and its projected row (what K5 gathers by depth rank) — behind the walk since round 4: with per-tile cuts
    // only the LISTED splats are ever gathered, a quarter of the visible ones, and the SH bands above DC (180 bytes per splat at
    // degree 3) are only fetched for them.  Without cuts every visible splat is listed.
    if (listed) {  // project_visible.rs:56-87
        const Vec3A v = normalize(sub(mean, camera_pos(u)));
        constexpr int C = (DEG + 1) * (DEG + 1);
        const Vec3A raw = sh_coeffs_to_color_dc<DEG>(coeffs + (size_t)gid * C * 3, v, sh_dc);
        const float cr = raw.x + 0.5f, cgc = raw.y + 0.5f, cb = raw.z + 0.5f;
        float* o = projected_by_gid + (size_t)gid * 9;
        o[0] = mx;
        o[1] = my;
        o[2] = conic.c00;
        o[3] = conic.c01;
        o[4] = conic.c11;
        o[5] = opac;
        o[6] = clampf(is_finite_f32(cr) ? cr : 0.0f, -100.0f, 100.0f);
        o[7] = clampf(is_finite_f32(cgc) ? cgc : 0.0f, -100.0f, 100.0f);
        o[8] = clampf(is_finite_f32(cb) ? cb : 0.0f, -100.0f, 100.0f);
    }
    if (gid < n) {
        depth_keys[gid] = listed ? key : 0xFFFFFFFFu;
        isect_counts[gid] = tiles_hit;
        max_radius[gid] = radius;
        if (zcut) near_counts[gid] = near_hit;
    }
    // block totals -> two global atomics per block (the reference does two per splat), spread over COUNTER_SLOTS
    // (visible, hits) pairs that the host adds up: 7814 atomics on ONE pair of addresses serialise at ~7 ns each on
    // this chip (cross-XCD atomics execute at the memory side) and held the kernel's retirement back by 53 us
    const unsigned long long ball = __ballot(visible);
    const unsigned long long lball = __ballot(listed);
    uint32_t wave_hits = tiles_hit;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wave_hits += __shfl_down(wave_hits, off);
    uint32_t wave_near = 0;
    if (zcut) {   // wave-uniform
        wave_near = near_hit;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wave_near += __shfl_down(wave_near, off);
    }
    // range of the visible depth keys, for the depth sort's split (depth_sort.hip): maxima of key and of ~key
    uint32_t kmax = listed ? key : 0u, nmax = listed ? ~key : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off));
        nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, off));
    }
    if (lane == 0) {
        s_vis[wave] = (uint32_t)__popcll(ball);
        s_hit[wave] = wave_hits;
        s_near[wave] = wave_near;
        s_listed[wave] = (uint32_t)__popcll(lball);
        s_kmax[wave] = kmax;
        s_nmax[wave] = nmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t v = 0, h = 0, nh = 0, ls = 0, km = 0, nm = 0;
#pragma unroll
        for (int k = 0; k < PROJ_WAVES; ++k) { v += s_vis[k]; h += s_hit[k]; nh += s_near[k]; ls += s_listed[k]; km = max(km, s_kmax[k]); nm = max(nm, s_nmax[k]); }
        const uint32_t sl = blockIdx.x & (COUNTER_SLOTS - 1u);
        unsigned long long* slot = counters + COUNTER_K1_U64 * sl;
        if (v) atomicAdd(&slot[0], (unsigned long long)v);
        if (h) atomicAdd(&slot[1], (unsigned long long)h);
        if (nh) atomicAdd(&slot[2], (unsigned long long)nh);
        if (ls && zcut) atomicAdd(&slot[3], (unsigned long long)ls);
        if (ls) {
            uint32_t* mm = reinterpret_cast<uint32_t*>(counters) + COUNTER_MINMAX_WORD + 2u * sl;
            atomicMax(&mm[0], km);
            atomicMax(&mm[1], nm);
        }
    }
}

template <bool MIP, bool PINHOLE>
static int launch_pf_deg(bh_ctx* ctx, const ViewUniforms& u, uint32_t n, uint32_t deg, const float* t, const float* sh, const float* ro,
                         uint32_t* keys, uint32_t* counts, float* radius, float* proj, unsigned long long* c64, const ForwardPrep& prep,
                         uint32_t* zcut, uint32_t* near_counts) {
    const dim3 grid((n + PROJ_WG - 1) / PROJ_WG), block(PROJ_WG);
    switch (deg) {
        case 0: hipLaunchKernelGGL((project_forward_kernel<MIP, PINHOLE, 0>), grid, block, 0, ctx->stream, u, n, t, sh, ro, keys, counts, radius, proj, c64, prep, zcut, near_counts); break;
        case 1: hipLaunchKernelGGL((project_forward_kernel<MIP, PINHOLE, 1>), grid, block, 0, ctx->stream, u, n, t, sh, ro, keys, counts, radius, proj, c64, prep, zcut, near_counts); break;
        case 2: hipLaunchKernelGGL((project_forward_kernel<MIP, PINHOLE, 2>), grid, block, 0, ctx->stream, u, n, t, sh, ro, keys, counts, radius, proj, c64, prep, zcut, near_counts); break;
        case 3: hipLaunchKernelGGL((project_forward_kernel<MIP, PINHOLE, 3>), grid, block, 0, ctx->stream, u, n, t, sh, ro, keys, counts, radius, proj, c64, prep, zcut, near_counts); break;
        case 4: hipLaunchKernelGGL((project_forward_kernel<MIP, PINHOLE, 4>), grid, block, 0, ctx->stream, u, n, t, sh, ro, keys, counts, radius, proj, c64, prep, zcut, near_counts); break;
        default: return set_error(ctx, BH_ERR_INVALID_ARG, "sh_degree must be 0..4");
    }
    BH_LAUNCH_CHECK(ctx, "project_forward_kernel");
    return 0;
}

int launch_project_forward(bh_ctx* ctx, const ViewUniforms& u, uint32_t n, bool mip, uint32_t sh_degree, const float* transforms,
                           const float* sh, const float* raw_opac, uint32_t* depth_keys, uint32_t* isect_counts, float* max_radius,
                           float* projected_by_gid, uint32_t* counters, const ForwardPrep& want, uint32_t* zcut, uint32_t* near_counts) {
    if (zcut && !near_counts) return set_error(ctx, BH_ERR_INVALID_ARG, "project_forward: a depth-cut table needs the near-count output");
    // what the grid cannot cover (tiny scenes under a large tile table, n == 0) is cleared with plain fills
    ForwardPrep prep = want;
    const size_t covered = (size_t)((n + PROJ_WG - 1) / PROJ_WG) * PROJ_WG;
    if (prep.visible && prep.visible_words > covered) {
        BH_HIP(ctx, hipMemsetAsync(prep.visible, 0, (size_t)prep.visible_words * 4, ctx->stream));
        prep.visible_words = 0;
    }
    if (prep.next_counters && COUNTER_SET_U64 > covered) {
        BH_HIP(ctx, hipMemsetAsync(prep.next_counters, 0, COUNTER_SET_BYTES, ctx->stream));
        prep.next_counters = nullptr;
    }
    if (prep.tile_table && prep.tile_words > covered) {
        BH_HIP(ctx, hipMemsetAsync(prep.tile_table, 0, (size_t)prep.tile_words * 4, ctx->stream));
        prep.tile_words = 0;
    }
    if (prep.slice_table && prep.slice_words > covered) {
        BH_HIP(ctx, hipMemsetAsync(prep.slice_table, 0, (size_t)prep.slice_words * 4, ctx->stream));
        prep.slice_words = 0;
    }
    if (n == 0) {
        if (prep.next_counters) BH_HIP(ctx, hipMemsetAsync(prep.next_counters, 0, COUNTER_SET_BYTES, ctx->stream));
        if (prep.span && prep.span_f4) BH_HIP(ctx, hipMemsetAsync(prep.span, 0, (size_t)prep.span_f4 * 16, ctx->stream));
        return 0;
    }
    auto* c64 = reinterpret_cast<unsigned long long*>(counters);
    const bool pinhole = u.model == CAM_PINHOLE;
    if (mip && pinhole) return launch_pf_deg<true, true>(ctx, u, n, sh_degree, transforms, sh, raw_opac, depth_keys, isect_counts, max_radius, projected_by_gid, c64, prep, zcut, near_counts);
    if (pinhole) return launch_pf_deg<false, true>(ctx, u, n, sh_degree, transforms, sh, raw_opac, depth_keys, isect_counts, max_radius, projected_by_gid, c64, prep, zcut, near_counts);
    if (mip) return launch_pf_deg<true, false>(ctx, u, n, sh_degree, transforms, sh, raw_opac, depth_keys, isect_counts, max_radius, projected_by_gid, c64, prep, zcut, near_counts);
    return launch_pf_deg<false, false>(ctx, u, n, sh_degree, transforms, sh, raw_opac, depth_keys, isect_counts, max_radius, projected_by_gid, c64, prep, zcut, near_counts);
}

// ---------------------------------------------------------------------------
// K4: project_visible  (kernels/project_visible.rs:23-88) — here: the records K1 stored by splat id,
// permuted into depth order.  One thread per output float: coalesced 4-byte stores, the nine reads of
// a row fall in one or two 64-byte sectors.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(PROJ_WG) void project_visible_kernel(uint32_t nv, const float* __restrict__ projected_by_gid,
                                                                 const uint32_t* __restrict__ global_from_compact_gid,
                                                                 float* __restrict__ projected) {
    const uint32_t e = blockIdx.x * PROJ_WG + threadIdx.x;
    if (e >= nv * 9u) return;
    const uint32_t cg = e / 9u, j = e - cg * 9u;
    projected[e] = projected_by_gid[(size_t)global_from_compact_gid[cg] * 9 + j];
}

int launch_project_visible(bh_ctx* ctx, uint32_t nv, const float* projected_by_gid, const uint32_t* gid, float* projected) {
    if (nv == 0) return 0;
    const uint64_t total = (uint64_t)nv * 9u;
    if (total > 0xFFFFFFFFull) return set_error(ctx, BH_ERR_UNSUPPORTED, "more than 2^32/9 visible splats");
    hipLaunchKernelGGL(project_visible_kernel, dim3((unsigned)((total + PROJ_WG - 1) / PROJ_WG)), dim3(PROJ_WG), 0, ctx->stream, nv,
                       projected_by_gid, gid, projected);
    BH_LAUNCH_CHECK(ctx, "project_visible_kernel");
    return 0;
}

// ---------------------------------------------------------------------------
// K5: map_gaussians_to_intersect  (kernels/map_gaussians.rs:15-80)
// ---------------------------------------------------------------------------
// The kernel also IS K4 on its way: lane cg fetches its record from the by-splat-id table K1 filled (one 36-byte
// gather through the depth permutation), uses it for the walk and stores it at row cg of `projected` — the compact,
// depth-ordered table the blend kernels and the backward read (coalesced 36-byte rows).  A separate gather launch
// (project_visible_kernel, still used when a frame has no intersections at all) cost 27 us.
// Depth-sliced lists (BH_FLAG_SLICED_LISTS, api.hip): the splats are in depth order and `cum_tiles_hit` is monotone, so
// "cum_tiles_hit[cg] <= budget" cuts the depth order into a NEAR slice — whose pairs are exactly the first I0 <= budget slots of
// the exact list — and a FAR rest.  FAR = false emits the near slice (budget = 0xFFFFFFFF: everything, the exact path) and
// reports where it ended (slice_info[0] = n0 splats, [1] = I0 pairs); FAR = true emits the rest, only into tiles whose pixels
// are not final yet (done_bits), at slots given by the scan of slice_count_kernel's counts.
// SPW = splats per wave (lanes SPW.. hold no splat but walk candidates like every lane).  The walk is a serial chain of LDS round
// trips per 64 candidates, and the near slice of a sliced frame is the few nearest = LARGEST splats: 64 of them in one wave are
// ~16 k candidates = 250 trips (38 us for the slowest wave, the kernel's whole duration, with most of the chip idle).  16 per
// wave: four times the waves, a quarter of the chain.  The exact path (every splat, the chip is full anyway) keeps 64.
template <bool FAR, int SPW>
__global__ __launch_bounds__(PROJ_WG) void map_gaussians_kernel(
    uint32_t nv, uint32_t tile_bw, uint32_t tile_bh, uint32_t tile_y0, uint32_t tile_y1, const float* __restrict__ projected_by_gid,
    const uint32_t* __restrict__ global_from_compact_gid, float* __restrict__ projected,
    const uint32_t* __restrict__ cum_tiles_hit, uint32_t* __restrict__ tile_id_from_isect,
    uint32_t* __restrict__ compact_gid_from_isect, float4* __restrict__ zero_span, uint32_t zero_f4, uint32_t budget,
    uint32_t* __restrict__ slice_info, const uint32_t* __restrict__ far_counts, const uint32_t* __restrict__ far_block_totals,
    const uint32_t* __restrict__ far_group_totals, const uint32_t* __restrict__ done_bits, const uint32_t* __restrict__ gate,
    const uint32_t* __restrict__ zcut, const uint32_t* __restrict__ depth_keys_sorted, const uint32_t* __restrict__ nv_dev, uint32_t pair_cap) {
    __shared__ WalkLds s_walk[PROJ_WAVES];
    __shared__ uint32_t s_far[2 * PROJ_WAVES];
    if (FAR && *gate == 0u) return;   // every tile is final: nothing left to list
    // nv_dev (api.hip "speculative K5"): queued before the host has read the frame's counts — `nv` is only a bound, the number of
    // listed splats sits on the device (the depth sort's first kernel left it there), and the pair buffers hold pair_cap entries: a
    // wave whose slot range does not fit emits nothing (the host sees the overflow in the counts and runs the kernel again).
    if (nv_dev) {
        const uint32_t live = *nv_dev;
        nv = live < nv ? live : nv;
        const size_t zf = ((size_t)nv * 10 + 3) / 4;
        zero_f4 = zf < (size_t)zero_f4 ? (uint32_t)zf : zero_f4;
    }
    // zcut != NULL (wave-uniform; FAR = false only): per-tile depth cuts instead of one slot budget — the compact splats are the
    // ones that own a pair in front of some cut, cum_tiles_hit is the scan of K1's near counts, and exactly those pairs are emitted
    const uint32_t tid_lin = blockIdx.x * PROJ_WG + threadIdx.x;
    // housekeeping for the backward: clear its v_combined accumulator on the way (coalesced, fire-and-forget)
    for (size_t i = tid_lin; i < zero_f4; i += (size_t)gridDim.x * PROJ_WG) zero_span[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cg0 = (blockIdx.x * PROJ_WAVES + (uint32_t)wave) * (uint32_t)SPW;   // the wave's first splat
    const uint32_t cg = lane < SPW ? cg0 + (uint32_t)lane : 0xFFFFFFFFu;
    // FAR: where this splat's pairs go = the far pairs of every block in front (slice_count_kernel's group and block totals, summed
    // here: a few coalesced loads per thread instead of two scan launches) + the block-local exclusive scan of the per-splat counts
    uint32_t far_base = 0, far_cnt = 0;
    if (FAR) {
        far_cnt = cg < nv ? far_counts[cg] : 0u;
        uint32_t ahead = 0;
        const uint32_t group = blockIdx.x / FAR_GROUP_BLOCKS;
        for (uint32_t g = threadIdx.x; g < group; g += PROJ_WG) ahead += far_group_totals[g];
        { const uint32_t b = group * FAR_GROUP_BLOCKS + threadIdx.x; if (threadIdx.x < FAR_GROUP_BLOCKS && b < blockIdx.x) ahead += far_block_totals[b]; }
        uint32_t incl = far_cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ahead += __shfl_down(ahead, off);
        if (lane == 63) s_far[wave] = incl;
        if (lane == 0) s_far[PROJ_WAVES + wave] = ahead;
        __syncthreads();
        uint32_t before = 0, block_ahead = 0, block_total = 0;
#pragma unroll
        for (int w = 0; w < PROJ_WAVES; ++w) {
            before += w < wave ? s_far[w] : 0u;
            block_total += s_far[w];
            block_ahead += s_far[PROJ_WAVES + w];
        }
        far_base = block_ahead + before + incl - far_cnt;
        // the last block knows the far slice's size: the sort and the offsets kernel take it from here
        if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 0) slice_info[3] = block_ahead + block_total;
    }
    float xy_x = 0.0f, xy_y = 0.0f, pt = 0.0f;
    Sym2 conic = Sym2{0.0f, 0.0f, 0.0f};
    TileBbox bb = TileBbox{0, 0, 0, 0};
    uint32_t base = 0, end = 0, nb = 0, zkey = 0;
    bool mine = false;   // this lane's splat belongs to the slice being emitted (and, FAR, still reaches a live tile)
    if (cg < nv) {
        // (the lane's independent loads side by side — its slot range, its depth key, its splat id — not one wait each)
        uint32_t cum_end = cum_tiles_hit[cg];
        uint32_t cum_prev = cum_tiles_hit[cg == 0u ? 0u : cg - 1u];
        uint32_t my_gid = global_from_compact_gid[cg];
        if (zcut) zkey = depth_keys_sorted[cg];
        asm volatile("" : "+v"(cum_end), "+v"(cum_prev), "+v"(my_gid), "+v"(zkey));
        if (cg == 0u) cum_prev = 0u;
        if (FAR) {
            base = far_base;
            end = far_base + far_cnt;
            mine = cum_end > budget && end > base;
        } else if (zcut) {
            base = cum_prev;
            end = cum_end;
            mine = end > base;
            if (slice_info && cg + 1u == nv) { slice_info[0] = nv; slice_info[1] = cum_end; }   // the far pass sorts behind the near list
        } else {
            base = cum_prev;
            end = cum_end;
            mine = cum_end <= budget;
            if (slice_info) {   // where the near slice ends (the next splat's range does not fit, or there is none)
                const uint32_t next_end = cg + 1u < nv ? cum_tiles_hit[cg + 1u] : 0xFFFFFFFFu;
                if (mine && (cg + 1u == nv || next_end > budget)) { slice_info[0] = cg + 1u; slice_info[1] = cum_end; }
                if (!mine && cg == 0u) { slice_info[0] = 0u; slice_info[1] = 0u; }
            }
        }
        if (mine) {
            const float* src = projected_by_gid + (size_t)my_gid * 9;
            float p[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) p[k] = src[k];
            float* dst = projected + (size_t)cg * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) dst[k] = p[k];
            xy_x = p[0]; xy_y = p[1];
            conic = Sym2{p[2], p[3], p[4]};
            pt = bh_logf(p[5] * 255.0f);
            float ex, ey;
            compute_bbox_extent(conic, pt, ex, ey);
            bb = get_tile_bbox(xy_x, xy_y, ex, ey, tile_bw, tile_y0, tile_y1);
            nb = (bb.max_y - bb.min_y) * (bb.max_x - bb.min_x);
        }
    }
    const unsigned long long mine_mask = __ballot(mine);
    if (mine_mask == 0ull) return;   // (wave-uniform; the kernel has no block barrier)
    WalkLds& w = s_walk[wave];
    // The wave's splats own ONE contiguous slot range: from the range start of its first emitting splat to the range end of
    // its last one (lanes in between that emit nothing have empty ranges: base == end, or — near slice — do not exist, the
    // slice being a prefix of the depth order).
    const int first_lane = __builtin_ctzll(mine_mask), last_lane = 63 - __builtin_clzll(mine_mask);
    const uint32_t wave_base = __shfl(base, first_lane);
    const uint32_t wave_total = __shfl(end, last_lane) - wave_base;
    if (!FAR && (wave_base > pair_cap || wave_total > pair_cap - wave_base)) return;   // (speculative launch only: pair_cap = 0xFFFFFFFF otherwise)
    // Emit order inside one splat is irrelevant: its tile ids are distinct, so after the
    // stable tile sort only the order ACROSS splats (depth order = slot ranges) survives.
    if (FAR) {
        (void)flat_tile_walk_emit(w, lane, nb, xy_x, xy_y, conic, pt, bb, wave_base, wave_total, tile_bw, cg0, tile_id_from_isect, compact_gid_from_isect, KeepLiveTiles{done_bits, tile_bw});
    } else {
        if (zcut) (void)flat_tile_walk_emit(w, lane, nb, xy_x, xy_y, conic, pt, bb, wave_base, wave_total, tile_bw, cg0, tile_id_from_isect, compact_gid_from_isect, KeepNearOfCut{zcut, tile_bw}, zkey);
        else (void)flat_tile_walk_emit(w, lane, nb, xy_x, xy_y, conic, pt, bb, wave_base, wave_total, tile_bw, cg0, tile_id_from_isect, compact_gid_from_isect);
    }
    (void)tile_bh;
}

// Second slice, pass 1: how many LIVE tiles (done bit clear) does each far splat reach?  counts[cg] = 0 for the near slice.
// Same walk, same test as K1 / K5 (only the tile filter in front of it), so the emit pass cannot disagree with the count.
// bit b set = band b of 32 equal bands of the `extent` tiles along one axis overlaps [lo, hi)
BH_DEV uint32_t band_of(uint32_t t, uint32_t extent) { return (t * 32u) / extent; }   // t < extent <= 4095: no overflow
BH_DEV uint32_t band_mask(uint32_t lo, uint32_t hi, uint32_t extent) {
    const uint32_t b0 = band_of(lo, extent), b1 = band_of(hi - 1u, extent);
    return (0xFFFFFFFFu >> (31u - (b1 - b0))) << b0;
}

__global__ __launch_bounds__(PROJ_WG) void slice_count_kernel(uint32_t nv, uint32_t tile_bw, uint32_t tile_bh, uint32_t tile_y0, uint32_t tile_y1,
                                                             const float* __restrict__ projected_by_gid,
                                                             const uint32_t* __restrict__ global_from_compact_gid,
                                                             const uint32_t* __restrict__ cum_tiles_hit, uint32_t budget,
                                                             const uint32_t* __restrict__ done_bits, const uint32_t* __restrict__ gate,
                                                             uint32_t* __restrict__ counts, uint32_t* __restrict__ block_totals,
                                                             uint32_t* __restrict__ group_totals, const uint32_t* __restrict__ live_bands) {
    __shared__ WalkLds s_walk[PROJ_WAVES];
    __shared__ uint32_t s_tot[PROJ_WAVES];
    if (*gate == 0u) return;
    // which 1/32 bands of tile columns / rows hold a live tile (the blend kernel ORs them together as it parks tiles): a splat whose
    // box misses them all reaches no live tile and is not walked — with few live tiles that is nearly every splat
    const uint32_t live_cols = live_bands[0], live_rows = live_bands[1];
    const uint32_t cg = blockIdx.x * PROJ_WG + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float xy_x = 0.0f, xy_y = 0.0f, pt = 0.0f;
    Sym2 conic = Sym2{0.0f, 0.0f, 0.0f};
    TileBbox bb = TileBbox{0, 0, 0, 0};
    uint32_t nb = 0;
    if (cg < nv && cum_tiles_hit[cg] > budget) {
        const float* src = projected_by_gid + (size_t)global_from_compact_gid[cg] * 9;
        xy_x = src[0]; xy_y = src[1];
        conic = Sym2{src[2], src[3], src[4]};
        pt = bh_logf(src[5] * 255.0f);
        float ex, ey;
        compute_bbox_extent(conic, pt, ex, ey);
        bb = get_tile_bbox(xy_x, xy_y, ex, ey, tile_bw, tile_y0, tile_y1);
        nb = (bb.max_y - bb.min_y) * (bb.max_x - bb.min_x);
        if (nb) {
            if ((band_mask(bb.min_x, bb.max_x, tile_bw) & live_cols) == 0u || (band_mask(bb.min_y, bb.max_y, tile_bh) & live_rows) == 0u) nb = 0u;
        }
    }
    WalkLds& w = s_walk[wave];
    uint32_t hits = 0;
    if (__ballot(nb > 0u) != 0ull) {
        const uint32_t wrank = flat_tile_walk(w, lane, nb, xy_x, xy_y, conic, pt, bb, [&](uint32_t r, uint32_t, uint32_t, uint32_t) { atomicAdd(&w.count[r], 1u); },
                                              KeepLiveTiles{done_bits, tile_bw});
        hits = nb ? w.count[wrank] : 0u;
    }
    if (cg < nv) counts[cg] = hits;
    uint32_t wsum = hits;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wsum += __shfl_down(wsum, off);
    if (lane == 0) s_tot[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < PROJ_WAVES; ++w) t += s_tot[w];
        block_totals[blockIdx.x] = t;
        if (t) atomicAdd(group_totals + blockIdx.x / FAR_GROUP_BLOCKS, t);
    }
}

int launch_map_gaussians(bh_ctx* ctx, uint32_t nv, const ViewUniforms& u, const float* projected_by_gid, const uint32_t* gid,
                         float* projected, const uint32_t* cum_tiles_hit, uint32_t* tile_ids, uint32_t* isect_gids,
                         float4* zero_span, uint32_t zero_f4, uint32_t budget, uint32_t* slice_info, const uint32_t* zcut,
                         const uint32_t* depth_keys_sorted, const uint32_t* nv_dev, uint32_t pair_cap) {
    if (nv == 0) return 0;
    if (zcut && (!slice_info || !depth_keys_sorted)) return set_error(ctx, BH_ERR_INVALID_ARG, "map_gaussians: a depth-cut table needs the slice words and the sorted keys");
    const dim3 grid((nv + PROJ_WG - 1) / PROJ_WG), block(PROJ_WG);
    const uint32_t* nul = nullptr;
    if (slice_info) {   // the near slice of a sliced frame: few, large splats (see SPW)
#define BH_K5_SPW 16
        constexpr int SPW = BH_K5_SPW;
        const dim3 grid16((nv + PROJ_WAVES * SPW - 1) / (PROJ_WAVES * SPW));
        hipLaunchKernelGGL((map_gaussians_kernel<false, SPW>), grid16, block, 0, ctx->stream, nv, u.tile_bw, u.tile_bh, u.tile_y0, u.tile_y1, projected_by_gid, gid,
                           projected, cum_tiles_hit, tile_ids, isect_gids, zero_span, zero_f4, budget, slice_info, nul, nul, nul, nul, nul, zcut, depth_keys_sorted, nv_dev, pair_cap);
    } else if (ctx->knob_k5_exact_spw != 64u) {
        // complete lists: 32 splats per wave (the wave's flat candidate list is half as long: 50.9 -> 46.9 us at 1 M splats / 9.8 M pairs;
        // 16: 49.5).  BH_K5_EXACT_SPW = 16 | 64 selects the others (A/B).
        const uint32_t spw = ctx->knob_k5_exact_spw == 16u ? 16u : 32u;
        const dim3 gridn((nv + PROJ_WAVES * spw - 1) / (PROJ_WAVES * spw));
        if (spw == 16u)
            hipLaunchKernelGGL((map_gaussians_kernel<false, 16>), gridn, block, 0, ctx->stream, nv, u.tile_bw, u.tile_bh, u.tile_y0, u.tile_y1, projected_by_gid, gid,
                               projected, cum_tiles_hit, tile_ids, isect_gids, zero_span, zero_f4, budget, slice_info, nul, nul, nul, nul, nul, nul, nul, nv_dev, pair_cap);
        else
            hipLaunchKernelGGL((map_gaussians_kernel<false, 32>), gridn, block, 0, ctx->stream, nv, u.tile_bw, u.tile_bh, u.tile_y0, u.tile_y1, projected_by_gid, gid,
                               projected, cum_tiles_hit, tile_ids, isect_gids, zero_span, zero_f4, budget, slice_info, nul, nul, nul, nul, nul, nul, nul, nv_dev, pair_cap);
    } else {
        hipLaunchKernelGGL((map_gaussians_kernel<false, 64>), grid, block, 0, ctx->stream, nv, u.tile_bw, u.tile_bh, u.tile_y0, u.tile_y1, projected_by_gid, gid,
                           projected, cum_tiles_hit, tile_ids, isect_gids, zero_span, zero_f4, budget, slice_info, nul, nul, nul, nul, nul, nul, nul, nv_dev, pair_cap);
    }
    BH_LAUNCH_CHECK(ctx, "map_gaussians_kernel");
    return 0;
}

// The far slice of a depth-sliced forward: count -> emit (the scan between them is folded into the emit kernel), both no-ops when
// *gate (the number of tiles the near slice left unsaturated) is zero.  counts: [nv], block_totals: [ceil(nv / 256)] scratch;
// group_totals: [ceil(blocks / FAR_GROUP_BLOCKS)], zero on entry; slice_info[3] receives the number of far pairs, slice_info[4..5]
// hold the live column / row bands.
int launch_map_gaussians_far(bh_ctx* ctx, uint32_t nv, const ViewUniforms& u, const float* projected_by_gid, const uint32_t* gid,
                             float* projected, const uint32_t* cum_tiles_hit, uint32_t budget, const uint32_t* done_bits, const uint32_t* gate,
                             uint32_t* counts, uint32_t* block_totals, uint32_t* group_totals, uint32_t* slice_info, uint32_t* tile_ids,
                             uint32_t* isect_gids) {
    if (nv == 0) return 0;
    const dim3 grid((nv + PROJ_WG - 1) / PROJ_WG), block(PROJ_WG);
    hipLaunchKernelGGL(slice_count_kernel, grid, block, 0, ctx->stream, nv, u.tile_bw, u.tile_bh, u.tile_y0, u.tile_y1, projected_by_gid, gid, cum_tiles_hit,
                       budget, done_bits, gate, counts, block_totals, group_totals, (const uint32_t*)(slice_info + 4));
    BH_LAUNCH_CHECK(ctx, "slice_count_kernel");
    hipLaunchKernelGGL((map_gaussians_kernel<true, 64>), grid, block, 0, ctx->stream, nv, u.tile_bw, u.tile_bh, u.tile_y0, u.tile_y1, projected_by_gid, gid,
                       projected, cum_tiles_hit, tile_ids, isect_gids, (float4*)nullptr, 0u, budget, slice_info, (const uint32_t*)counts,
                       (const uint32_t*)block_totals, (const uint32_t*)group_totals, done_bits, gate, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0xFFFFFFFFu);
    BH_LAUNCH_CHECK(ctx, "map_gaussians_kernel<far>");
    return 0;
}

// ---------------------------------------------------------------------------
// K18: project_backward  (bwd/kernels/project_backwards.rs:101-254)
// ---------------------------------------------------------------------------
BH_DEV Quat apply_normalize_vjp(Quat q, Quat g) {
    const float lsq = qdot(q, q);
    const float l = __builtin_sqrtf(lsq);
    const float inv = 1.0f / (l * lsq);
    const float qw = q.w, qx = q.x, qy = q.y, qz = q.z;
    const float gw = g.w, gx = g.x, gy = g.y, gz = g.z;
    const float cc0 = -qw * qx, cc1 = -qx * qy, cc2 = -qy * qw;
    const float cs0 = -qw * qz, cs1 = -qx * qz, cs2 = -qy * qz;
    const float sw = qw * qw, sx = qx * qx, sy = qy * qy, sz = qz * qz;
    return Quat{((lsq - sw) * gw + cc0 * gx + cc2 * gy + cs0 * gz) * inv,
                (cc0 * gw + (lsq - sx) * gx + cc1 * gy + cs1 * gz) * inv,
                (cc2 * gw + cc1 * gx + (lsq - sy) * gy + cs2 * gz) * inv,
                (cs0 * gw + cs1 * gx + cs2 * gy + (lsq - sz) * gz) * inv};
}
BH_DEV Quat quat_to_mat_vjp(Quat q, const Mat3& v) {
    const float qw = q.w, qx = q.x, qy = q.y, qz = q.z;
    const float w_grad = qx * (v.c1z - v.c2y) + qy * (v.c2x - v.c0z) + qz * (v.c0y - v.c1x);
    const float x_grad = -2.0f * qx * (v.c1y + v.c2z) + qy * (v.c0y + v.c1x) + qz * (v.c0z + v.c2x) + qw * (v.c1z - v.c2y);
    const float y_grad = qx * (v.c0y + v.c1x) - 2.0f * qy * (v.c0x + v.c2z) + qz * (v.c1z + v.c2y) + qw * (v.c2x - v.c0z);
    const float z_grad = qx * (v.c0z + v.c2x) + qy * (v.c1z + v.c2y) - 2.0f * qz * (v.c0x + v.c1y) + qw * (v.c0y - v.c1x);
    return Quat{2.0f * w_grad, 2.0f * x_grad, 2.0f * y_grad, 2.0f * z_grad};
}
BH_DEV Sym2 inverse2x2_vjp(Sym2 minv, Sym2 v) {
    const float tmp00 = -minv.c00 * v.c00 + -minv.c01 * v.c01;
    const float tmp01 = -minv.c01 * v.c00 + -minv.c11 * v.c01;
    const float tmp10 = -minv.c00 * v.c01 + -minv.c01 * v.c11;
    const float tmp11 = -minv.c01 * v.c01 + -minv.c11 * v.c11;
    return Sym2{tmp00 * minv.c00 + tmp10 * minv.c01, tmp01 * minv.c00 + tmp11 * minv.c01, tmp01 * minv.c01 + tmp11 * minv.c11};
}
// camera_model/pinhole.rs:59-123
BH_DEV Vec3A projection_vjp_pinhole(const Mat2x3& jac, Vec3A mean_c, Sym3 cov_c, const ViewUniforms& u, Sym2 v_cov2d, Vec2 v_mean2d) {
    const float fx = u.fx, fy = u.fy;
    const float mx = mean_c.x, my = mean_c.y, mz = mean_c.z;
    const float inv_z = 1.0f / mz;
    const float mx_rz_raw = mx * inv_z, my_rz_raw = my * inv_z;
    const float mx_rz = clampf(mx_rz_raw, u.lim_neg_x, u.lim_pos_x);
    const float my_rz = clampf(my_rz_raw, u.lim_neg_y, u.lim_pos_y);
    const bool in_x = mx_rz_raw <= u.lim_pos_x && mx_rz_raw >= u.lim_neg_x;
    const bool in_y = my_rz_raw <= u.lim_pos_y && my_rz_raw >= u.lim_neg_y;
    const float inv_z2 = inv_z * inv_z;
    const float inv_z3 = inv_z2 * inv_z;
    float v_mx = fx * inv_z * v_mean2d.x;
    float v_my = fy * inv_z * v_mean2d.y;
    float v_mz = -(fx * mx * v_mean2d.x + fy * my * v_mean2d.y) * inv_z2;
    const Mat2x3 tmp = sym2_mul_mat2x3(v_cov2d, jac);
    const float vj00 = 2.0f * dot(row0(tmp), s3row0(cov_c));
    const float vj11 = 2.0f * dot(row1(tmp), s3row1(cov_c));
    const float vj20 = 2.0f * dot(row0(tmp), s3row2(cov_c));
    const float vj21 = 2.0f * dot(row1(tmp), s3row2(cov_c));
    const float tx = mz * mx_rz;
    const float ty = mz * my_rz;
    if (in_x) v_mx += -fx * inv_z2 * vj20; else v_mz += -fx * inv_z3 * vj20 * tx;
    if (in_y) v_my += -fy * inv_z2 * vj21; else v_mz += -fy * inv_z3 * vj21 * ty;
    v_mz += -fx * inv_z2 * vj00 - fy * inv_z2 * vj11 + 2.0f * fx * tx * inv_z3 * vj20 + 2.0f * fy * ty * inv_z3 * vj21;
    return Vec3A{v_mx, v_my, v_mz};
}

template <bool MIP, int DEG, bool PINHOLE>
__global__ __launch_bounds__(PROJ_WG) void project_backward_kernel(
    ViewUniforms u, uint32_t nv, const float* __restrict__ transforms, const float* __restrict__ sh_coeffs,
    const float* __restrict__ raw_opac, const uint32_t* __restrict__ global_from_compact_gid,
    float* __restrict__ v_combined, float* __restrict__ v_transforms, float* __restrict__ v_coeffs,
    float* __restrict__ v_raw_opac, float* __restrict__ v_refine_weight, const bool mark_written, const float* __restrict__ projected) {
    const uint32_t cg = blockIdx.x * PROJ_WG + threadIdx.x;
    if (cg >= nv) return;
    uint32_t gid = global_from_compact_gid[cg];
    float* rg = v_combined + (size_t)cg * 10;
    float g[10];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 10; ++k) g[k] = rg[k];
    asm volatile("" : "+v"(gid));   // (the splat id travels with the accumulator row, not behind the test on it)
    // The blend backward left RAW sums (rasterize.hip): P Q = sums of v_sigma (pixel - mean), R2 R3 R4 = its second moments, the
    // three colour sums, Vs = sum of v_sigma, the refine weight.  Their per-splat linear maps (rasterize_backwards.rs:318-381:
    // v_xy = -conic (P, Q), v_conic = (R2/2, R3, R4/2), the colour clamp's gates, v_alpha0 = -Vs / alpha0) commute with the sum
    // over tiles and are applied here, once per splat; the row is stored back, so v_combined IS RasterizeGrads afterwards.
    {
        bool raw_any = false;
#pragma unroll
        for (int k = 0; k < 10; ++k) raw_any = raw_any || (g[k] != 0.0f);
        if (!raw_any) return;   // (nothing reached this splat: the row is zero and stays zero)
        const float* pr = projected + (size_t)cg * 9;
        const float c00 = pr[2], c01 = pr[3], c11 = pr[4], a0 = pr[5], cr = pr[6], cg_ = pr[7], cb = pr[8];
        const float P = g[0], Q = g[1];
        g[0] = -__builtin_fmaf(c00, P, c01 * Q);
        g[1] = -__builtin_fmaf(c11, Q, c01 * P);
        g[2] = 0.5f * g[2];
        g[4] = 0.5f * g[4];
        g[5] = cr >= 0.0f ? g[5] : 0.0f;    // rasterize.rs:147-149: a colour channel clamped at 0 passes no gradient
        g[6] = cg_ >= 0.0f ? g[6] : 0.0f;
        g[7] = cb >= 0.0f ? g[7] : 0.0f;
        g[8] = -g[8] / a0;
#pragma unroll
        for (int k = 0; k < 10; ++k) rg[k] = g[k];
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) any = any || (g[k] != 0.0f);
    constexpr int C = (DEG + 1) * (DEG + 1);
    if (!any) return;   // the row stays what it is: zero (the caller cleared the dense outputs) or, with mark_written, unmarked
    // the splat's inputs side by side (they were six dependent round trips: mean, scale x, y, z, quaternion, opacity)
    float tr[10];
    {
        const float* trp = transforms + (size_t)gid * 10;
#pragma unroll
        for (int k = 0; k < 10; ++k) tr[k] = trp[k];
    }
    float raw_o = raw_opac[gid];
    asm volatile("" : "+v"(tr[0]), "+v"(tr[1]), "+v"(tr[2]), "+v"(tr[3]), "+v"(tr[4]), "+v"(tr[5]), "+v"(tr[6]), "+v"(tr[7]), "+v"(tr[8]), "+v"(tr[9]), "+v"(raw_o));
    const Vec3A mean = v3(tr[0], tr[1], tr[2]);
    const Vec3A scl = v3(bh_expf(tr[7]), bh_expf(tr[8]), bh_expf(tr[9]));
    const Quat qu = Quat{tr[3], tr[4], tr[5], tr[6]};
    const Quat q = qnormalize(qu);
    const Vec3A u_world = sub(mean, camera_pos(u));
    const float u_len = length(u_world);
    const Vec3A v = scale(u_world, 1.0f / u_len);
    const Vec3A v_color = v3(g[5], g[6], g[7]);
    sh_coeffs_to_color_vjp<DEG>(v_coeffs + (size_t)gid * C * 3, v, v_color);
    const Vec3A v_v_sh = sh_color_viewdir_vjp<DEG>(sh_coeffs + (size_t)gid * C * 3, v, v_color);
    const float v_dot_vv = dot(v, v_v_sh);
    const Vec3A v_mean_from_sh = scale(sub(v_v_sh, scale(v, v_dot_vv)), 1.0f / u_len);
    const Vec3A mean_c = world_to_cam(mean, u);
    const Mat3 r = quat_to_mat3(q);
    const Mat3 m = mul_diag(r, scl);
    const Sym2 raw_cov = calc_cov2d<PINHOLE>(scl, q, mean_c, u);
    float filter_comp;
    const Sym2 cov = compensate_cov2d<MIP>(raw_cov, filter_comp);
    const float os = sigmoid(raw_o);
    v_raw_opac[gid] = filter_comp * g[8] * os * (1.0f - os);
    const float refine_clean = is_finite_f32(g[9]) ? g[9] : 0.0f;
    // mark_written (the single-GPU train step, which clears ONLY the refine-weight vector): the sign bit of the (non-negative)
    // refine weight says "this splat's gradient rows were written this step" — the update kernel reads no other row
    const float refine_w = clampf(refine_clean, 0.0f, 1.0e32f);
    v_refine_weight[gid] = mark_written ? u2f(f2u(refine_w) | 0x80000000u) : refine_w;
    const Sym2 conic_inv = sym2_inverse(cov);
    const Sym2 v_inv = Sym2{g[2], g[3] * 0.5f, g[4]};
    const Sym2 v_cov2d = inverse2x2_vjp(conic_inv, v_inv);
    const Sym3 covar = outer_product_self(m);
    const Mat3 view_rot = view_rotation(u);
    const Sym3 cov_c = congruence(covar, view_rot);
    const Mat2x3 jac = project_jacobian<PINHOLE>(mean_c, u);
    const Vec2 v_xy = Vec2{g[0], g[1]};
    Vec3A v_mean_c;  // camera_model/mod.rs:86-125
    if (PINHOLE) v_mean_c = projection_vjp_pinhole(jac, mean_c, cov_c, u, v_cov2d, v_xy);
    else if (u.model == CAM_KB4) v_mean_c = projection_vjp_kb4(jac, mean_c, cov_c, u, v_cov2d, v_xy, u.dist);
    else if (u.model == CAM_RT8) v_mean_c = projection_vjp_rt8(mean_c, cov_c, u, v_cov2d, v_xy, u.dist);
    else v_mean_c = projection_vjp_tpf(jac, mean_c, cov_c, u, v_cov2d, v_xy, u.dist);
    const Sym3 vcc = transpose_congruence_sym2(jac, v_cov2d);
    const Vec3A v_mean = add(transpose_mul_vec3(view_rot, v_mean_c), v_mean_from_sh);
    const Mat3 v_m = sym3_mul_mat3(sym3_scale(transpose_congruence(vcc, view_rot), 2.0f), m);
    const Vec3A v_scale = v3(dot(col0(r), col0(v_m)) * scl.x, dot(col1(r), col1(v_m)) * scl.y, dot(col2(r), col2(v_m)) * scl.z);
    const Quat q_grad = quat_to_mat_vjp(q, mul_diag(v_m, scl));
    const Quat v_q = apply_normalize_vjp(qu, q_grad);
    float* vt = v_transforms + (size_t)gid * 10;
    vt[0] = v_mean.x; vt[1] = v_mean.y; vt[2] = v_mean.z;
    vt[3] = v_q.w; vt[4] = v_q.x; vt[5] = v_q.y; vt[6] = v_q.z;
    vt[7] = v_scale.x; vt[8] = v_scale.y; vt[9] = v_scale.z;
}

template <bool MIP, bool PINHOLE>
static int launch_pb_deg(bh_ctx* ctx, const ViewUniforms& u, uint32_t nv, uint32_t deg, const float* t, const float* sh,
                         const float* ro, const uint32_t* gid, float* vc, float* vt, float* vsh, float* vro, float* vr, bool rm, const float* pj) {
    const dim3 grid((nv + PROJ_WG - 1) / PROJ_WG), block(PROJ_WG);
    switch (deg) {
        case 0: hipLaunchKernelGGL((project_backward_kernel<MIP, 0, PINHOLE>), grid, block, 0, ctx->stream, u, nv, t, sh, ro, gid, vc, vt, vsh, vro, vr, rm, pj); break;
        case 1: hipLaunchKernelGGL((project_backward_kernel<MIP, 1, PINHOLE>), grid, block, 0, ctx->stream, u, nv, t, sh, ro, gid, vc, vt, vsh, vro, vr, rm, pj); break;
        case 2: hipLaunchKernelGGL((project_backward_kernel<MIP, 2, PINHOLE>), grid, block, 0, ctx->stream, u, nv, t, sh, ro, gid, vc, vt, vsh, vro, vr, rm, pj); break;
        case 3: hipLaunchKernelGGL((project_backward_kernel<MIP, 3, PINHOLE>), grid, block, 0, ctx->stream, u, nv, t, sh, ro, gid, vc, vt, vsh, vro, vr, rm, pj); break;
        case 4: hipLaunchKernelGGL((project_backward_kernel<MIP, 4, PINHOLE>), grid, block, 0, ctx->stream, u, nv, t, sh, ro, gid, vc, vt, vsh, vro, vr, rm, pj); break;
        default: return set_error(ctx, BH_ERR_INVALID_ARG, "sh_degree must be 0..4");
    }
    BH_LAUNCH_CHECK(ctx, "project_backward_kernel");
    return 0;
}

int launch_project_backward(bh_ctx* ctx, const ViewUniforms& u, uint32_t nv, bool mip, uint32_t sh_degree,
                            const float* transforms, const float* sh, const float* raw_opac, const uint32_t* gid,
                            float* v_combined, float* v_transforms, float* v_sh, float* v_raw_opac,
                            float* v_refine, bool row_mask, const float* projected) {
    if (nv == 0) return 0;
    if (!projected) return set_error(ctx, BH_ERR_INVALID_ARG, "project_backward: the projected rows are needed to map the raw gradient sums");
    if (u.model == CAM_PINHOLE)
        return mip ? launch_pb_deg<true, true>(ctx, u, nv, sh_degree, transforms, sh, raw_opac, gid, v_combined, v_transforms, v_sh, v_raw_opac, v_refine, row_mask, projected)
                   : launch_pb_deg<false, true>(ctx, u, nv, sh_degree, transforms, sh, raw_opac, gid, v_combined, v_transforms, v_sh, v_raw_opac, v_refine, row_mask, projected);
    return mip ? launch_pb_deg<true, false>(ctx, u, nv, sh_degree, transforms, sh, raw_opac, gid, v_combined, v_transforms, v_sh, v_raw_opac, v_refine, row_mask, projected)
               : launch_pb_deg<false, false>(ctx, u, nv, sh_degree, transforms, sh, raw_opac, gid, v_combined, v_transforms, v_sh, v_raw_opac, v_refine, row_mask, projected);
}

}  // namespace bh
