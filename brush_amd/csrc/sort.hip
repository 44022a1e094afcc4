// sort.hip — stable LSD radix argsort of (u32 key, u32 value) pairs.
//
// Contract = brush_sort::radix_argsort (brush-sort/src/lib.rs:16-125): sort by the
// low `bits` bits of the key, stable.  Only the result is contractual; the
// reference's 4-bit FidelityFX-style passes (5 kernels per pass, kernels.rs:29-401)
// are replaced by a wave64 design:
//   * 8-bit digits: half the passes (4 for depth keys, 2 for <=16-bit tile ids);
//   * per pass: histogram (+ digit totals) -> one-launch row scan of the [digit][block]
//     table -> scatter  (3 launches; a chained-scan "onesweep" was measured SLOWER on
//     MI355X: every look-back hop is a cross-XCD fabric round trip);
//   * ranking inside a block uses wave-wide digit matching (8 ballots) so each wave
//     ranks 64 keys per step without LDS atomics; waves own contiguous chunks so
//     the block order is the input order (stability);
//   * keys are re-ordered through LDS before the global write so every digit run
//     leaves the block as one contiguous, coalesced burst.
// HBM traffic per pass: 4 (hist read) + 8 (scatter read) + 8 (write) = 20 B/pair —
// the algorithmic figure of SURVEY.md §8d.
#include "context.h"

namespace bh {

constexpr int SORT_WG = 256;
constexpr int SORT_WAVES = SORT_WG / 64;
// keys per thread: 16 (4096 keys per block) for large sorts; 8 for sorts of up to SMALL_SORT_MAX keys (the depth sort:
// 1 M keys are only 245 blocks of 4096 on 256 CUs — each block is a serial latency chain with nothing to overlap;
// 2048-key blocks halve the chain and put two of them on a CU)
constexpr uint32_t SMALL_SORT_MAX = 2u << 20;
constexpr int RADIX = 256;

// `mask` is 0xFF except in the last pass, where it keeps only the bits still
// inside `bits` (the contract is "sort on the low `bits` bits").
BH_DEV uint32_t digit_of(uint32_t key, uint32_t shift, uint32_t mask) { return (key >> shift) & mask; }

// lanes of this wave whose digit equals mine
BH_DEV unsigned long long match_digit(uint32_t d) {
    unsigned long long m = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned long long bal = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    return m;
}

// Device-side length (the depth-sliced forward, api.hip): the host knows only an upper bound `n` of the number of pairs; the
// exact count sits in device memory (`n_dev`), `gate` == 0 switches the whole sort off (every block leaves at once), and the
// last pass writes its output `*out_base` elements into the destination.  The grid and the [digit][block] table are sized for
// the bound; blocks past the live count return before touching anything.  All three NULL: the plain sort.
struct SortDyn {
    const uint32_t* n_dev = nullptr;
    const uint32_t* gate = nullptr;
    const uint32_t* out_base = nullptr;
};
BH_DEV uint32_t sort_live_n(uint32_t n, const SortDyn& dyn) {
    if (dyn.gate && *dyn.gate == 0u) return 0u;
    if (dyn.n_dev) {
        const uint32_t v = *dyn.n_dev;
        return v < n ? v : n;
    }
    return n;
}

// hist[digit * nblocks + block]
// (electing one leader per digit group with 8 ballots instead of the LDS atomic was measured: tile sort 132 -> 155 us)
template <int SORT_KPT>
__global__ __launch_bounds__(SORT_WG) void radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift,
                                                            uint32_t mask, uint32_t nblocks, uint32_t* __restrict__ hist, SortDyn dyn) {
    __shared__ uint32_t s_hist[SORT_WAVES][RADIX];
    n = sort_live_n(n, dyn);
    if (blockIdx.x * (uint32_t)(SORT_WG * SORT_KPT) >= n) return;   // (only with a device-side length)
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < SORT_WAVES * RADIX; i += SORT_WG) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    constexpr int SORT_TILE = SORT_WG * SORT_KPT;
    const uint32_t base = blockIdx.x * SORT_TILE;
    if (base + (uint32_t)SORT_TILE <= n && (reinterpret_cast<uintptr_t>(keys) & 15u) == 0) {
        // every block but the last: 16-byte loads (which thread counts which key is irrelevant to a histogram)
        const uint4* k4 = reinterpret_cast<const uint4*>(keys + base);   // base is a multiple of 1024 keys
        uint4 v[SORT_KPT / 4];
#pragma unroll
        for (int k = 0; k < SORT_KPT / 4; ++k) v[k] = k4[k * SORT_WG + tid];
#pragma unroll
        for (int k = 0; k < SORT_KPT / 4; ++k) {
            atomicAdd(&s_hist[wave][digit_of(v[k].x, shift, mask)], 1u);
            atomicAdd(&s_hist[wave][digit_of(v[k].y, shift, mask)], 1u);
            atomicAdd(&s_hist[wave][digit_of(v[k].z, shift, mask)], 1u);
            atomicAdd(&s_hist[wave][digit_of(v[k].w, shift, mask)], 1u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < SORT_KPT; ++k) {
            const uint32_t idx = base + k * SORT_WG + tid;
            if (idx < n) atomicAdd(&s_hist[wave][digit_of(keys[idx], shift, mask)], 1u);
        }
    }
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) total += s_hist[w][tid];
    hist[(size_t)tid * nblocks + blockIdx.x] = total;
}

// Row-wise exclusive scan of the [digit][block] table in ONE launch (block d owns digit
// row d and also emits the row total); the scatter kernel adds the 256-entry prefix over
// the digit totals itself.  3 launches per pass instead of the 5 of a generic
// reduce/spine/apply scan over the whole table.
constexpr int ROWSCAN_MAX_EPT = 16;  // rows of up to 4096 blocks (16.7 M keys) in one trip; longer rows loop with a carry
__global__ __launch_bounds__(SORT_WG) void radix_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t nblocks, uint32_t n, uint32_t tile,
                                                               uint32_t* __restrict__ digit_totals, SortDyn dyn) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t d = blockIdx.x;
    const uint32_t live_n = sort_live_n(n, dyn);
    if (live_n == 0u) return;
    const uint32_t live = (live_n + tile - 1u) / tile;   // blocks that wrote their column (== nblocks for the plain sort)
    uint32_t* row = hist + (size_t)d * nblocks;
    __shared__ uint32_t s_chunk[ROWSCAN_MAX_EPT][SORT_WAVES];
    uint32_t run = 0;
    for (uint32_t c0 = 0; c0 < live; c0 += (uint32_t)ROWSCAN_MAX_EPT * SORT_WG) {
        // the row segment stays in registers: entry c0 + k*256 + tid (coalesced), ROWSCAN_MAX_EPT chunks
        uint32_t v[ROWSCAN_MAX_EPT], incl[ROWSCAN_MAX_EPT];
#pragma unroll
        for (int k = 0; k < ROWSCAN_MAX_EPT; ++k) {
            const uint32_t i = c0 + (uint32_t)k * SORT_WG + tid;
            v[k] = i < live ? row[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < ROWSCAN_MAX_EPT; ++k) {
            uint32_t x = v[k];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(x, off);
                if (lane >= off) x += t;
            }
            incl[k] = x;
            if (lane == 63) s_chunk[k][wave] = x;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ROWSCAN_MAX_EPT; ++k) {
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) {
                const uint32_t c = s_chunk[k][w];
                before += w < wave ? c : 0u;
                total += c;
            }
            const uint32_t i = c0 + (uint32_t)k * SORT_WG + tid;
            if (i < live) row[i] = run + before + incl[k] - v[k];
            run += total;
        }
        __syncthreads();   // s_chunk is rewritten by the next trip
    }
    if (tid == 0) digit_totals[d] = run;
}

template <bool HAS_VALS, int SORT_KPT>
__global__ __launch_bounds__(SORT_WG) void radix_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                               uint32_t n, uint32_t shift, uint32_t mask, uint32_t nblocks,
                                                               const uint32_t* __restrict__ offsets,  // exclusive scan of hist
                                                               const uint32_t* __restrict__ digit_totals,  // row-scan mode: offsets are per-row, add the digit prefix
                                                               uint32_t* __restrict__ out_keys, uint32_t* __restrict__ out_vals, SortDyn dyn) {
    constexpr int SORT_TILE = SORT_WG * SORT_KPT;
    n = sort_live_n(n, dyn);
    if (blockIdx.x * (uint32_t)SORT_TILE >= n) return;   // (only with a device-side length)
    if (dyn.out_base) {   // the last pass of a sort that appends to an existing list
        const uint32_t ob = *dyn.out_base;
        out_keys += ob;
        out_vals += ob;
    }
    __shared__ uint32_t s_cnt[SORT_WAVES][RADIX];   // per-wave running digit counts -> wave bases
    __shared__ uint32_t s_dbase[RADIX];             // exclusive scan of block digit totals
    __shared__ uint32_t s_gofs[RADIX];              // global offset of (digit, block) minus s_dbase
    __shared__ uint32_t s_keys[SORT_TILE];
    __shared__ uint32_t s_vals[SORT_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < SORT_WAVES * RADIX; i += SORT_WG) (&s_cnt[0][0])[i] = 0;
    __syncthreads();

    const uint32_t block_base = blockIdx.x * SORT_TILE;
    const uint32_t wave_base = block_base + wave * (64 * SORT_KPT);
    uint32_t key[SORT_KPT], val[SORT_KPT], rank[SORT_KPT];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // this thread's two table entries are needed only after the ranking: issue their loads first so the (strided, L2)
    // round trip runs under it
    const uint32_t my_offset = offsets[(size_t)tid * nblocks + blockIdx.x];
    const uint32_t my_gtotal = digit_totals ? digit_totals[tid] : 0u;
    if (block_base + (uint32_t)SORT_TILE <= n) {   // every block but the last: no bounds tests around the loads
#pragma unroll
        for (int k = 0; k < SORT_KPT; ++k) {
            const uint32_t idx = wave_base + k * 64 + lane;
            key[k] = keys[idx];
            val[k] = HAS_VALS ? vals[idx] : idx;
        }
    } else {
#pragma unroll
        for (int k = 0; k < SORT_KPT; ++k) {
            const uint32_t idx = wave_base + k * 64 + lane;
            const bool valid = idx < n;
            key[k] = valid ? keys[idx] : 0xFFFFFFFFu;
            val[k] = valid ? (HAS_VALS ? vals[idx] : idx) : 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < SORT_KPT; ++k) {
        // invalid tail elements take digit 255 and sit at the highest in-block
        // positions, so they never disturb the rank of a valid element.
        const uint32_t idx = wave_base + k * 64 + lane;
        const uint32_t d = idx < n ? digit_of(key[k], shift, mask) : 0xFFu;
        const unsigned long long peers = match_digit(d);
        const uint32_t prior = s_cnt[wave][d];
        rank[k] = prior + (uint32_t)__popcll(peers & lt_mask);
        // all lanes have read `prior` (one wave executes in lock-step and LDS ops
        // retire in order) before the leader of each digit group bumps the count
        if ((peers & lt_mask) == 0ull) s_cnt[wave][d] = prior + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    // per digit: exclusive scan over waves -> wave bases; block totals
    uint32_t total;
    {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            const uint32_t c = s_cnt[w][tid];
            s_cnt[w][tid] = run;
            run += c;
        }
        total = run;
    }
    // exclusive scan of the 256 digit totals across the block
    {
        const uint32_t gt = my_gtotal;
        uint32_t incl = total, gincl = gt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off), g = __shfl_up(gincl, off);
            if (lane >= off) { incl += t; gincl += g; }
        }
        __shared__ uint32_t s_wsum[2][SORT_WAVES];
        if (lane == 63) { s_wsum[0][wave] = incl; s_wsum[1][wave] = gincl; }
        __syncthreads();
        uint32_t wofs = 0, gofs = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) { wofs += (w < wave) ? s_wsum[0][w] : 0u; gofs += (w < wave) ? s_wsum[1][w] : 0u; }
        const uint32_t excl = incl - total + wofs;
        const uint32_t below = gincl - gt + gofs;  // keys with a smaller digit (0 when offsets already hold the full scan)
        s_dbase[tid] = excl;
        s_gofs[tid] = below + my_offset - excl;
    }
    __syncthreads();
    // local reorder: position inside the block's digit-sorted tile
#pragma unroll
    for (int k = 0; k < SORT_KPT; ++k) {
        const uint32_t idx = wave_base + k * 64 + lane;
        const uint32_t d = idx < n ? digit_of(key[k], shift, mask) : 0xFFu;
        const uint32_t lpos = s_dbase[d] + s_cnt[wave][d] + rank[k];
        s_keys[lpos] = key[k];
        s_vals[lpos] = val[k];
    }
    __syncthreads();
    const uint32_t valid_in_block = n - block_base < (uint32_t)SORT_TILE ? n - block_base : (uint32_t)SORT_TILE;
#pragma unroll
    for (int k = 0; k < SORT_KPT; ++k) {
        const uint32_t e = k * SORT_WG + tid;
        if (e < valid_in_block) {
            const uint32_t kk = s_keys[e];
            const uint32_t pos = s_gofs[digit_of(kk, shift, mask)] + e;
            out_keys[pos] = kk;
            out_vals[pos] = s_vals[e];
        }
    }
}

// alloc_n >= n: what the scratch slots are sized for (a caller whose bound n moves from frame to frame passes its largest: the
// arena only grows, and growing waits for the stream)
static int radix_argsort_impl(bh_ctx* ctx, const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t bits,
                              uint32_t* out_keys, uint32_t* out_vals, const SortDyn& dyn, uint32_t alloc_n = 0) {
    if (bits > 32) return set_error(ctx, BH_ERR_INVALID_ARG, "radix_argsort: bits must be <= 32");
    if (n == 0) return 0;
    const bool dynamic = dyn.n_dev || dyn.gate || dyn.out_base;
    // digits of equal width: 13-bit tile ids sort as 7 + 6 bits, not 8 + 5 — the wider a pass, the shorter the digit runs
    // a block writes (4096 keys over 256 digits = 64-byte bursts; over 128 digits = 128-byte bursts)
    const uint32_t passes = bits == 0 ? 1 : (bits + 7) / 8;
    const uint32_t base_w = bits / passes, wide = bits % passes;   // the first `wide` passes take one extra bit
    const bool small = n <= SMALL_SORT_MAX;
    uint32_t kpt = small ? 8u : 16u;
    if (const uint32_t k = ctx->knob_sort_kpt) {   // developer knob BH_SORT_KPT (read once at bh_create): 4 | 8 | 16
        if ((k == 4 && n <= (1u << 22)) || k == 8 || k == 16) kpt = k;
    }
    const uint32_t tile = SORT_WG * kpt;
    const uint32_t nblocks = (n + tile - 1) / tile;
    const size_t bytes = (size_t)n * 4;
    if (alloc_n < n) alloc_n = n;
    const size_t alloc_bytes = (size_t)alloc_n * 4;
    const uint32_t alloc_blocks = (alloc_n + tile - 1) / tile;
    // [256] digit totals followed by the [256][nblocks] table
    uint32_t* totals = (uint32_t*)ensure(ctx, SLOT_SORT_HIST, ((size_t)RADIX * alloc_blocks + RADIX) * 4);
    if (!totals) return BH_ERR_OOM;
    uint32_t* hist = totals + RADIX;
    // (a device-side length always takes the row scan: it walks only the live part of each row, however long the table)
    const bool rowscan = dynamic || nblocks <= (uint32_t)ROWSCAN_MAX_EPT * SORT_WG;
    // Ping-pong through two scratch pairs; pass 0 reads the caller's input (never
    // written), the last pass lands in out_* unless that would alias its source
    // (single-pass in-place call), in which case it is staged and copied.
    uint32_t* sk[2] = {nullptr, nullptr};
    uint32_t* sv[2] = {nullptr, nullptr};
    const bool in_place = keys == out_keys || (vals && vals == out_vals);
    if (dynamic && in_place) return set_error(ctx, BH_ERR_INVALID_ARG, "radix_argsort: a device-length sort needs distinct input and output buffers");
    if (passes > 1 || in_place) {
        sk[0] = (uint32_t*)ensure(ctx, SLOT_SORT_KEYS_A, alloc_bytes);
        sv[0] = (uint32_t*)ensure(ctx, SLOT_SORT_VALS_A, alloc_bytes);
        if (!sk[0] || !sv[0]) return BH_ERR_OOM;
    }
    if (passes > 2) {
        sk[1] = (uint32_t*)ensure(ctx, SLOT_SORT_KEYS_B, alloc_bytes);
        sv[1] = (uint32_t*)ensure(ctx, SLOT_SORT_VALS_B, alloc_bytes);
        if (!sk[1] || !sv[1]) return BH_ERR_OOM;
    }
    const uint32_t* src_k = keys;
    const uint32_t* src_v = vals;
    for (uint32_t p = 0; p < passes; ++p) {
        const bool last = p + 1 == passes;
        uint32_t* dst_k = sk[p & 1];
        uint32_t* dst_v = sv[p & 1];
        if (last && src_k != out_keys && src_v != out_vals) {
            dst_k = out_keys;
            dst_v = out_vals;
        }
        SortDyn pd = dyn;
        if (!last) pd.out_base = nullptr;
        const uint32_t width = base_w + (p < wide ? 1u : 0u);
        const uint32_t shift = p * base_w + (p < wide ? p : wide);
        const uint32_t mask = (1u << width) - 1u;
        if (kpt == 4u) hipLaunchKernelGGL(radix_hist_kernel<4>, dim3(nblocks), dim3(SORT_WG), 0, ctx->stream, src_k, n, shift, mask, nblocks, hist, pd);
        else if (kpt == 8u) hipLaunchKernelGGL(radix_hist_kernel<8>, dim3(nblocks), dim3(SORT_WG), 0, ctx->stream, src_k, n, shift, mask, nblocks, hist, pd);
        else hipLaunchKernelGGL(radix_hist_kernel<16>, dim3(nblocks), dim3(SORT_WG), 0, ctx->stream, src_k, n, shift, mask, nblocks, hist, pd);
        BH_LAUNCH_CHECK(ctx, "radix_hist_kernel");
        if (rowscan) {
            hipLaunchKernelGGL(radix_rowscan_kernel, dim3(RADIX), dim3(SORT_WG), 0, ctx->stream, hist, nblocks, n, tile, totals, pd);
            BH_LAUNCH_CHECK(ctx, "radix_rowscan_kernel");
        } else {
            BH_TRY(prefix_sum(ctx, hist, nullptr, RADIX * nblocks, hist, /*exclusive=*/true));
        }
        const uint32_t* tot = rowscan ? totals : nullptr;
        const dim3 grid(nblocks), block(SORT_WG);
        if (kpt == 4u) {
            if (src_v) hipLaunchKernelGGL((radix_scatter_kernel<true, 4>), grid, block, 0, ctx->stream, src_k, src_v, n, shift, mask, nblocks, hist, tot, dst_k, dst_v, pd);
            else hipLaunchKernelGGL((radix_scatter_kernel<false, 4>), grid, block, 0, ctx->stream, src_k, src_v, n, shift, mask, nblocks, hist, tot, dst_k, dst_v, pd);
        } else if (kpt == 8u) {
            if (src_v) hipLaunchKernelGGL((radix_scatter_kernel<true, 8>), grid, block, 0, ctx->stream, src_k, src_v, n, shift, mask, nblocks, hist, tot, dst_k, dst_v, pd);
            else hipLaunchKernelGGL((radix_scatter_kernel<false, 8>), grid, block, 0, ctx->stream, src_k, src_v, n, shift, mask, nblocks, hist, tot, dst_k, dst_v, pd);
        } else {
            if (src_v) hipLaunchKernelGGL((radix_scatter_kernel<true, 16>), grid, block, 0, ctx->stream, src_k, src_v, n, shift, mask, nblocks, hist, tot, dst_k, dst_v, pd);
            else hipLaunchKernelGGL((radix_scatter_kernel<false, 16>), grid, block, 0, ctx->stream, src_k, src_v, n, shift, mask, nblocks, hist, tot, dst_k, dst_v, pd);
        }
        BH_LAUNCH_CHECK(ctx, "radix_scatter_kernel");
        src_k = dst_k;
        src_v = dst_v;
    }
    if (src_k != out_keys) BH_HIP(ctx, hipMemcpyAsync(out_keys, src_k, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (src_v != out_vals) BH_HIP(ctx, hipMemcpyAsync(out_vals, src_v, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------
// The forward's tile sort + get_tile_offsets (render.rs:228-243, get_tile_offset.rs) in FIVE launches instead of seven.
//
// The (tile id, compact splat id) pairs arrive in depth order and leave grouped by tile, depth order kept.  As two LSD passes
// plus the offsets kernel that is hist / row scan / scatter twice and one more launch: seven dependent launches of 5-18 us
// for 20 MB.  Here the FIRST pass takes the HIGH digit (the three kernels above, stable): the pairs of one digit — a bucket of
// 2^low_bits consecutive tiles — then sit together, still in depth order, and ONE block per bucket finishes the job without any
// global table: it counts the bucket's pairs per tile and wave (a wave owns a contiguous part of the bucket), scans, writes the
// tiles' [begin, end) rows of the offsets table — it knows them — and places every pair at
//     bucket start + pairs of lower tiles + pairs of the same tile owned by earlier waves + rank inside the wave's part,
// which is the stable order.  Ranking is the scatter kernel's (wave-wide digit matching, no LDS atomics).
// ---------------------------------------------------------------------------
// Work is dealt in PARTS, not in buckets: a bucket of `size` pairs is ceil(size / TP_CHUNK) parts and ONE block handles one part,
// whichever bucket it belongs to — a frame whose pairs sit in a few consecutive tiles (a zoomed-in view: through round 5 one
// 1024-thread block per bucket made that a cliff, 4 M pairs in one bucket = ~10 ms) sorts as fast as an even one, and complete
// lists (7.8 M pairs: 30 k per bucket) no longer run 30 ranking steps per wave in a row.  Two launches:
//   count: every part counts its pairs per (wave, tile of the bucket) and per tile                          -> part tables
//   place: every part adds up, per tile, the parts of its bucket (all of them: the tile's row of the offsets table and the pairs
//          of the bucket's lower tiles; the earlier ones: where its own pairs of that tile start) and places its pairs.
constexpr uint32_t TP_CHUNK = 4096;   // pairs per part
constexpr int TP_WG = 512;            // 8 waves, 512 contiguous pairs each (8 ranking steps)
constexpr int TP_WAVES = TP_WG / 64;
constexpr int TP_STEPS = TP_CHUNK / TP_WG;   // 64-pair steps per wave
constexpr int TB_MAXBINS = 256;       // low_bits <= 8
static_assert(TP_CHUNK == (uint32_t)TP_WAVES * 64u * (uint32_t)TP_STEPS, "a part is waves x steps x 64 pairs");

struct PartInfo { uint32_t bucket, part, parts, first_block, start, size; };
// Which part block `b` is: parts are numbered bucket by bucket.  Block-uniform; false = no such part (the grid is sized for the
// worst case n / TP_CHUNK + 256).  s_scan: [2][TP_WAVES] scratch.
BH_DEV bool find_part(const uint32_t* __restrict__ digit_totals, uint32_t b, uint32_t* s_scan, uint32_t* s_found, PartInfo& out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t size = tid < RADIX ? digit_totals[tid] : 0u;
    const uint32_t parts = (size + TP_CHUNK - 1u) / TP_CHUNK;
    uint32_t ip = parts, is = size;   // inclusive scans over the 256 buckets (the first four waves)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t tp = __shfl_up(ip, off), ts = __shfl_up(is, off);
        if (lane >= off) { ip += tp; is += ts; }
    }
    if (lane == 63) { s_scan[wave] = ip; s_scan[TP_WAVES + wave] = is; }
    if (tid == 0) s_found[0] = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t wp = 0, ws = 0;
#pragma unroll
    for (int w = 0; w < RADIX / 64; ++w) if (w < wave) { wp += s_scan[w]; ws += s_scan[TP_WAVES + w]; }
    const uint32_t first = ip - parts + wp;   // exclusive
    if (tid < RADIX && b >= first && b < first + parts) {
        s_found[0] = (uint32_t)tid; s_found[1] = first; s_found[2] = parts; s_found[3] = is - size + ws; s_found[4] = size;
    }
    __syncthreads();
    if (s_found[0] == 0xFFFFFFFFu) return false;
    out.bucket = s_found[0]; out.first_block = s_found[1]; out.parts = s_found[2]; out.start = s_found[3]; out.size = s_found[4];
    out.part = b - out.first_block;
    return true;
}

// lanes of the wave that hold the same tile of the bucket (low_bits ballots)
BH_DEV unsigned long long tile_peers(uint32_t b, bool valid, uint32_t low_bits) {
    unsigned long long m = __ballot(valid);
    for (uint32_t k = 0; k < low_bits; ++k) {
        const unsigned long long bal = __ballot((b >> k) & 1u);
        m &= ((b >> k) & 1u) ? bal : ~bal;
    }
    return m;
}

// part_tab: per block [TP_WAVES + 1][bins]: rows 0..7 = pairs per (wave, tile), row 8 = pairs per tile of the whole part
__global__ __launch_bounds__(TP_WG) void tile_parts_count_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ digit_totals,
                                                                uint32_t low_bits, uint32_t* __restrict__ part_tab) {
    __shared__ uint32_t s_cnt[TP_WAVES][TB_MAXBINS];
    __shared__ uint32_t s_scan[2 * TP_WAVES];
    __shared__ uint32_t s_found[5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    PartInfo pi;
    if (!find_part(digit_totals, blockIdx.x, s_scan, s_found, pi)) return;
    const uint32_t bins = 1u << low_bits, low_mask = bins - 1u;
    for (uint32_t i = (uint32_t)tid; i < (uint32_t)TP_WAVES * bins; i += TP_WG) s_cnt[i / bins][i % bins] = 0u;
    __syncthreads();
    const uint32_t part_lo = pi.part * TP_CHUNK, part_hi = min(pi.size, part_lo + TP_CHUNK);
    const uint32_t lo = min(part_hi, part_lo + (uint32_t)wave * (64u * TP_STEPS)), hi = min(part_hi, lo + 64u * TP_STEPS);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t kk[TP_STEPS];
#pragma unroll
    for (int k = 0; k < TP_STEPS; ++k) {   // all loads first: the ranking steps are LDS round trips that cannot start before their key
        const uint32_t i = lo + (uint32_t)k * 64u + (uint32_t)lane;
        kk[k] = keys[pi.start + (i < hi ? i : (hi > lo ? lo : 0u))];
    }
#pragma unroll
    for (int k = 0; k < TP_STEPS; ++k) {
        const uint32_t i0 = lo + (uint32_t)k * 64u;
        if (i0 >= hi) break;   // wave-uniform
        const bool valid = i0 + (uint32_t)lane < hi;
        const uint32_t bn = valid ? (kk[k] & low_mask) : 0u;
        const unsigned long long peers = tile_peers(bn, valid, low_bits);
        if (valid && (peers & lt_mask) == 0ull) atomicAdd(&s_cnt[wave][bn], (uint32_t)__popcll(peers));   // (no return value: nothing waits for it)
    }
    __syncthreads();
    uint32_t* tab = part_tab + (size_t)blockIdx.x * (TP_WAVES + 1) * bins;
    for (uint32_t i = (uint32_t)tid; i < bins; i += TP_WG) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < TP_WAVES; ++w) {
            const uint32_t c = s_cnt[w][i];
            tab[(uint32_t)w * bins + i] = run;   // exclusive over the part's waves
            run += c;
        }
        tab[(uint32_t)TP_WAVES * bins + i] = run;
    }
}

__global__ __launch_bounds__(TP_WG) void tile_parts_place_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                const uint32_t* __restrict__ digit_totals, uint32_t low_bits, uint32_t num_tiles,
                                                                const uint32_t* __restrict__ part_tab, uint32_t* __restrict__ out_keys,
                                                                uint32_t* __restrict__ out_vals, uint32_t* __restrict__ tile_offsets) {
    __shared__ uint32_t s_cnt[TP_WAVES][TB_MAXBINS];   // running position of (wave, tile) inside the tile's run
    __shared__ uint32_t s_base[TB_MAXBINS];            // pairs of the bucket's lower tiles
    __shared__ uint32_t s_acc[2][TP_WG];               // column sums in flight: [0] all parts, [1] the earlier parts
    __shared__ uint32_t s_scan[2 * TP_WAVES];
    __shared__ uint32_t s_found[5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    PartInfo pi;
    if (!find_part(digit_totals, blockIdx.x, s_scan, s_found, pi)) return;
    const uint32_t bins = 1u << low_bits, low_mask = bins - 1u;
    const uint32_t part_lo = pi.part * TP_CHUNK, part_hi = min(pi.size, part_lo + TP_CHUNK);
    const uint32_t lo = min(part_hi, part_lo + (uint32_t)wave * (64u * TP_STEPS)), hi = min(part_hi, lo + 64u * TP_STEPS);
    // the part's pairs are on their way while the tables are added up
    uint32_t kk[TP_STEPS], vv[TP_STEPS];
#pragma unroll
    for (int k = 0; k < TP_STEPS; ++k) {
        const uint32_t i = lo + (uint32_t)k * 64u + (uint32_t)lane;
        const uint32_t ic = pi.start + (i < hi ? i : (hi > lo ? lo : 0u));
        kk[k] = keys[ic];
        vv[k] = vals[ic];
    }
    // per tile of the bucket: pairs in all parts / in the parts before this one.  rows = TP_WG / bins parts are read per trip
    const uint32_t rows = (uint32_t)TP_WG / bins, col = (uint32_t)tid % bins, row = (uint32_t)tid / bins;
    uint32_t all = 0u, before = 0u;
    for (uint32_t q = row; q < pi.parts; q += rows) {
        const uint32_t c = part_tab[((size_t)(pi.first_block + q) * (TP_WAVES + 1) + TP_WAVES) * bins + col];
        all += c;
        before += q < pi.part ? c : 0u;
    }
    s_acc[0][tid] = all;
    s_acc[1][tid] = before;
    __syncthreads();
    uint32_t total = 0u, mine_before = 0u;
    if ((uint32_t)tid < bins) {
        for (uint32_t r = 0; r < rows; ++r) { total += s_acc[0][r * bins + (uint32_t)tid]; mine_before += s_acc[1][r * bins + (uint32_t)tid]; }
    }
    {
        uint32_t incl = total;   // exclusive scan of the tile totals over the bucket (bins <= 256: the first four waves)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        __syncthreads();   // (s_scan was read by find_part)
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        uint32_t wofs = 0;
#pragma unroll
        for (int w = 0; w < RADIX / 64; ++w) wofs += (w < wave) ? s_scan[w] : 0u;
        if ((uint32_t)tid < bins) {
            const uint32_t excl = incl - total + wofs;
            s_base[tid] = excl + mine_before;   // where THIS part's pairs of the tile start inside the bucket
            const uint32_t tile = (pi.bucket << low_bits) | (uint32_t)tid;
            if (pi.part == 0u && total != 0u && tile < num_tiles) {   // (an absent tile keeps the zeros the table was cleared to; sentinel ids have no row)
                tile_offsets[tile * 2] = pi.start + excl;
                tile_offsets[tile * 2 + 1] = pi.start + excl + total;
            }
        }
    }
    const uint32_t* tab = part_tab + (size_t)blockIdx.x * (TP_WAVES + 1) * bins;
    for (uint32_t i = (uint32_t)tid; i < (uint32_t)TP_WAVES * bins; i += TP_WG) s_cnt[i / bins][i % bins] = tab[i];
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < TP_STEPS; ++k) {
        const uint32_t i0 = lo + (uint32_t)k * 64u;
        if (i0 >= hi) break;   // wave-uniform
        const bool valid = i0 + (uint32_t)lane < hi;
        const uint32_t bn = valid ? (kk[k] & low_mask) : 0u;
        const unsigned long long peers = tile_peers(bn, valid, low_bits);
        // the group's leader takes the running count with a RETURNING LDS atomic and hands it to its peers: a wave's LDS operations
        // execute in program order, so the steps stay in depth order, and no step's registers depend on the previous step's
        uint32_t prior = 0u;
        if (valid && (peers & lt_mask) == 0ull) prior = atomicAdd(&s_cnt[wave][bn], (uint32_t)__popcll(peers));
        prior = (uint32_t)__shfl((int)prior, peers ? __ffsll((long long)peers) - 1 : 0);
        if (valid) {
            const uint32_t pos = pi.start + s_base[bn] + prior + (uint32_t)__popcll(peers & lt_mask);
            out_keys[pos] = kk[k];
            out_vals[pos] = vv[k];
        }
    }
}

bool tile_sort_supported(uint32_t bits, uint32_t n) { return bits > 8u && bits <= 16u && n <= (16u << 20); }

// keys = tile ids (< 2^bits, or the sentinel 0xFFFFFFFF), vals = compact splat ids, n pairs in depth order.  -> out_keys / out_vals
// sorted by tile (stable) and tile_offsets[tile] = [begin, end) for every tile that has pairs (the table must be zero already).
int tile_sort_offsets(bh_ctx* ctx, const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t bits, uint32_t num_tiles,
                      uint32_t* out_keys, uint32_t* out_vals, uint32_t* tile_offsets, uint32_t alloc_n) {
    if (!tile_sort_supported(bits, n)) return set_error(ctx, BH_ERR_INVALID_ARG, "tile_sort_offsets: 9..16 key bits, at most 16 M pairs");
    if (n == 0) return 0;
    const uint32_t low_bits = bits - 8u, shift = low_bits, mask = 0xFFu;
    uint32_t kpt = n <= SMALL_SORT_MAX ? 8u : 16u;
    if (const uint32_t k = ctx->knob_sort_kpt) { if (k == 8 || k == 16) kpt = k; }
    const uint32_t tile = SORT_WG * kpt;
    const uint32_t nblocks = (n + tile - 1) / tile;
    if (alloc_n < n) alloc_n = n;
    const uint32_t alloc_blocks = (alloc_n + tile - 1) / tile;
    uint32_t* totals = (uint32_t*)ensure(ctx, SLOT_SORT_HIST, ((size_t)RADIX * alloc_blocks + RADIX) * 4);
    uint32_t* mid_k = (uint32_t*)ensure(ctx, SLOT_SORT_KEYS_A, (size_t)alloc_n * 4);
    uint32_t* mid_v = (uint32_t*)ensure(ctx, SLOT_SORT_VALS_A, (size_t)alloc_n * 4);
    if (!totals || !mid_k || !mid_v) return BH_ERR_OOM;
    uint32_t* hist = totals + RADIX;
    const SortDyn none{};
    const dim3 grid(nblocks), block(SORT_WG);
    if (kpt == 8u) hipLaunchKernelGGL(radix_hist_kernel<8>, grid, block, 0, ctx->stream, keys, n, shift, mask, nblocks, hist, none);
    else hipLaunchKernelGGL(radix_hist_kernel<16>, grid, block, 0, ctx->stream, keys, n, shift, mask, nblocks, hist, none);
    BH_LAUNCH_CHECK(ctx, "radix_hist_kernel");
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(RADIX), dim3(SORT_WG), 0, ctx->stream, hist, nblocks, n, tile, totals, none);
    BH_LAUNCH_CHECK(ctx, "radix_rowscan_kernel");
    if (kpt == 8u) hipLaunchKernelGGL((radix_scatter_kernel<true, 8>), grid, block, 0, ctx->stream, keys, vals, n, shift, mask, nblocks, hist, totals, mid_k, mid_v, none);
    else hipLaunchKernelGGL((radix_scatter_kernel<true, 16>), grid, block, 0, ctx->stream, keys, vals, n, shift, mask, nblocks, hist, totals, mid_k, mid_v, none);
    BH_LAUNCH_CHECK(ctx, "radix_scatter_kernel");
    // parts of at most TP_CHUNK pairs, numbered bucket by bucket: at most n / TP_CHUNK + one partial part per bucket
    const uint32_t max_parts = n / TP_CHUNK + RADIX, alloc_parts = alloc_n / TP_CHUNK + RADIX;
    uint32_t* part_tab = (uint32_t*)ensure(ctx, SLOT_SORT_PARTS, (size_t)alloc_parts * (TP_WAVES + 1) * ((size_t)1 << low_bits) * 4);
    if (!part_tab) return BH_ERR_OOM;
    hipLaunchKernelGGL(tile_parts_count_kernel, dim3(max_parts), dim3(TP_WG), 0, ctx->stream, mid_k, totals, low_bits, part_tab);
    BH_LAUNCH_CHECK(ctx, "tile_parts_count_kernel");
    hipLaunchKernelGGL(tile_parts_place_kernel, dim3(max_parts), dim3(TP_WG), 0, ctx->stream, mid_k, mid_v, totals, low_bits, num_tiles, part_tab, out_keys, out_vals,
                       tile_offsets);
    BH_LAUNCH_CHECK(ctx, "tile_parts_place_kernel");
    return 0;
}

int radix_argsort(bh_ctx* ctx, const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t bits,
                  uint32_t* out_keys, uint32_t* out_vals) {
    return radix_argsort_impl(ctx, keys, vals, n, bits, out_keys, out_vals, SortDyn{});
}

int radix_argsort_dev(bh_ctx* ctx, const uint32_t* keys, const uint32_t* vals, uint32_t n_max, const uint32_t* n_dev, const uint32_t* gate,
                      const uint32_t* out_base, uint32_t bits, uint32_t* out_keys, uint32_t* out_vals, uint32_t alloc_n) {
    SortDyn dyn;
    dyn.n_dev = n_dev;
    dyn.gate = gate;
    dyn.out_base = out_base;
    return radix_argsort_impl(ctx, keys, vals, n_max, bits, out_keys, out_vals, dyn, alloc_n);
}

}  // namespace bh
