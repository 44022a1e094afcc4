// depth_sort.hip — the forward's depth ordering: stable argsort of the per-splat depth keys AND the prefix sum of the
// per-splat tile counts in that order, in FOUR launches.
//
// Reference: render.rs:177-187 — radix_argsort(depths, 32 bits) (brush-sort: 8 four-bit passes, 40 launches), then
// int_gather + prefix_sum (brush-prefix-sum: 5 launches).  Round 1 ran this as a generic 8-bit LSD sort (4 passes x 3
// launches) + a two-launch scan: 14 launches, ~7.5 us each, for 8 MB of keys — launch latency, not bandwidth (0.12 ms).
//
// Here the structure of the keys is used.  A depth key is the bit pattern of a positive float (culled splats carry
// 0xFFFFFFFF), and the visible keys of a frame span a narrow range of it [kmin, kmax], which K1 leaves behind (per-block
// maxima of key and ~key in 128 slots).  So:
//   1. ONE most-significant-digit split: digit = (key - kmin) >> shift with shift the smallest that keeps the visible range
//      inside digits 0..254 (culled -> 255): histogram, row scan of the [digit][block] table, stable scatter — the three
//      kernels of a radix pass.  The histogram kernel also adds up the tile counts per digit.
//   2. ONE kernel with a block per digit bucket finishes the job: the keys of a bucket agree in everything above `shift`,
//      so the block sorts them on the low `shift` bits (stable LSD passes through LDS, 4096 keys at a time, running digit
//      bases carried from chunk to chunk: a bucket of any size works, typical ones are one or two chunks) and, knowing
//      the tile counts of the buckets in front of it from step 1, writes the inclusive scan of the tile counts in sorted
//      order on its way out.
// The result is the same permutation the stable 32-bit sort produces (ties keep splat-id order), so everything downstream
// stays bit-identical.  bh_radix_argsort / bh_prefix_sum (the reference's generic operators) keep the generic kernels.
#include <cstddef>

#include "context.h"

namespace bh {

constexpr int DS_WG = 256;
constexpr int DS_WAVES = DS_WG / 64;
#define BH_DS_KPT 8   /* 16 until late in round 4: 2048-key chunks put twice the blocks on the chip for the histogram and the split
                         (split 12.6 -> 9.8 us at 1 M splats, histogram + row scan +0.6); the fused path then reaches 8.4 M splats */
constexpr int DS_KPT = BH_DS_KPT;
constexpr int DS_TILE = DS_WG * DS_KPT;   // 2048 keys per chunk
constexpr int DS_RADIX = 256;
constexpr uint32_t CULLED_KEY = 0xFFFFFFFFu;

struct DepthSplit {
    uint32_t kmin, kmax;
    uint32_t scale;      // digit = ((key - kmin) * scale) >> 32: the visible range [kmin, kmax] spread over ALL of the digits 0..254
    bool any_visible;
    const uint32_t* spl; // LDS: [255] splitters of the frame before (SPLITTERS below) or NULL: the linear split
};
// SPLITTERS (round 6).  A linear split of the key range is balanced only while the depths are spread evenly over it: a scene whose
// splats crowd into a thin shell of depths (or one with a few far outliers) puts most keys into a handful of the 255 buckets, and the
// bucket kernel lasts as long as its largest bucket (beyond FAST_CAP keys: the chunked many-pass path — 2 ms instead of 20 us for half
// a million keys in one bucket).  The bucket kernel already holds every bucket SORTED — so it leaves the 254 exact quantiles of the
// frame's visible keys behind (spl[j] = the key of rank (j + 1) Nv / 255), and the next frame OF THE SAME VIEW takes the number of
// splitters <= key as its digit: an eight-step search in LDS.  A view's depth distribution changes slowly from one visit to the next;
// a frame whose key range [kmin, kmax] has moved by more than a sixteenth of the table's (another list mode, a scene that has been
// replaced) ignores the table — keys beyond the last splitter would all land in one bucket — and splits linearly, as does a
// view's first frame.  The permutation is the stable sort's whatever the digits are.  One table per view and list mode (api.hip keeps
// them behind the view's tile table; frames without a view share one per ctx); it is read by the histogram and split kernels and
// rewritten IN PLACE by the bucket kernel behind them.
//   table [DSORT_SPL_STRIDE]: [0..253] splitters, ascending | [254] 0xFFFFFFFF (the search's sentinel) | [255] != 0: valid |
//                             [256] kmin, [257] kmax of the frame that wrote it
constexpr uint32_t SPL_WORDS = 258;
static_assert(SPL_WORDS <= DSORT_SPL_STRIDE, "context.h reserves the table");
constexpr uint32_t SPL_MIN_KEYS = 16384;   // frames with fewer visible keys leave no table (the linear split serves them)
// A frame that finds NO table (a view's first frame in this list mode; the host knows) does not fall back to the linear split blindly:
// every SPL_SAMPLE_STRIDE-th key is sorted first — the same four kernels on 1/64 of the keys, linear split (a crowded sample is a few
// thousand keys in one bucket: nothing) — and the bucket kernel of THAT run leaves the sample's quantiles as the frame's table.  ~30 us,
// once per view and list mode; without it the first frame of a scene whose depths crowd into a thin shell took 2.4 ms instead of 0.44.
constexpr uint32_t SPL_SAMPLE_STRIDE = 64;
constexpr uint32_t SPL_SAMPLE_MIN_N = 1u << 17;     // smaller frames are sorted as they come
constexpr uint32_t SPL_SAMPLE_MIN_KEYS = 1024;      // visible keys of the sample below which it leaves no table
BH_DEV DepthSplit make_split(uint32_t kmax, uint32_t nmin) {
    DepthSplit sp;
    sp.any_visible = nmin != 0u;      // ~key of a visible key is never 0 (the key would be 0xFFFFFFFF)
    sp.kmin = ~nmin;
    sp.kmax = kmax;
    const uint32_t range = sp.any_visible ? kmax - sp.kmin : 0u;
    // largest scale with (range * scale) >> 32 <= 254
    sp.scale = (uint32_t)min((unsigned long long)0xFFFFFFFFull, (255ull << 32) / ((unsigned long long)range + 1ull));
    if ((((unsigned long long)range * sp.scale) >> 32) > 254ull) sp.scale -= 1u;
    sp.spl = nullptr;
    return sp;
}

// Every block derives the split from K1's 128 (max key, max ~key) pairs itself: 1 KB of L2 hits and one block reduction —
// cheaper than a launch or a grid-wide hand-over.
BH_DEV DepthSplit depth_split(const uint32_t* __restrict__ minmax, uint32_t* s_red /*[2 * DS_WAVES]*/, const uint32_t* __restrict__ spl_in /*NULL: linear*/,
                              uint32_t* s_spl /*[SPL_WORDS]*/) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t a = 0, b = 0;
    if (tid < (int)COUNTER_SLOTS) {
        a = minmax[2 * tid];
        b = minmax[2 * tid + 1];
    }
    if (spl_in != nullptr) {   // (the barriers below publish it)
        s_spl[tid] = spl_in[tid];
        if (tid < (int)SPL_WORDS - DS_WG) s_spl[DS_WG + tid] = spl_in[DS_WG + tid];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a = max(a, (uint32_t)__shfl_xor((int)a, off));
        b = max(b, (uint32_t)__shfl_xor((int)b, off));
    }
    if (lane == 0) { s_red[wave] = a; s_red[DS_WAVES + wave] = b; }
    __syncthreads();
    uint32_t kmax = 0, nmin = 0;
#pragma unroll
    for (int w = 0; w < DS_WAVES; ++w) { kmax = max(kmax, s_red[w]); nmin = max(nmin, s_red[DS_WAVES + w]); }
    __syncthreads();
    DepthSplit sp = make_split(kmax, nmin);
    if (spl_in != nullptr && sp.any_visible && s_spl[255] != 0u) {   // a valid table: does this frame's key range still match its frame's?
        const uint32_t tmin = s_spl[256], tmax = s_spl[257], tol = (tmax - tmin) >> 4;
        const uint32_t dmin = sp.kmin > tmin ? sp.kmin - tmin : tmin - sp.kmin, dmax = sp.kmax > tmax ? sp.kmax - tmax : tmax - sp.kmax;
        if (dmin <= tol && dmax <= tol) sp.spl = s_spl;
    }
    return sp;
}
BH_DEV uint32_t depth_digit(uint32_t key, const DepthSplit& sp) {
    if (key == CULLED_KEY) return 255u;
    if (sp.spl != nullptr) {   // (block-uniform) number of splitters <= key: 0..254 (spl[254] = 0xFFFFFFFF > every visible key)
        uint32_t d = 0;
#pragma unroll
        for (uint32_t step = 128u; step != 0u; step >>= 1) d += sp.spl[d + step - 1u] <= key ? step : 0u;
        return d;
    }
    return min(254u, (uint32_t)(((unsigned long long)(key - sp.kmin) * sp.scale) >> 32));
}

// lanes of this wave whose digit equals mine
BH_DEV unsigned long long ds_match(uint32_t d) {
    uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const uint32_t bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        const uint32_t flip = bit - 1u;   // bit set: 0, clear: ~0
        lo &= (uint32_t)bal ^ flip;
        hi &= (uint32_t)(bal >> 32) ^ flip;
    }
    return ((unsigned long long)hi << 32) | lo;
}

// The mid-pipeline readback (render.rs:146-168) without a copy launch: K1's counter set (context.h: [COUNTER_SLOTS] block totals +
// the previous frame's slicing feedback) is added up by one wave of the first kernel queued behind K1 and stored straight into
// the pinned host block — [4] u64 totals | [3] u32 feedback (max need, unsaturated pairs, unsaturated tiles).  A blit kernel
// between K1 and the sort cost 4.5 us of device time plus its two launch gaps in every frame.
BH_DEV void counter_sums_to_host(const uint32_t* __restrict__ set, uint32_t* __restrict__ host_sums, int lane, uint32_t tag, uint32_t* __restrict__ dev_sums) {
    const unsigned long long* c64 = reinterpret_cast<const unsigned long long*>(set);
    unsigned long long tot[COUNTER_K1_U64];
#pragma unroll
    for (uint32_t c = 0; c < COUNTER_K1_U64; ++c) tot[c] = 0ull;
    uint32_t need = 0, pairs = 0, tiles = 0;
    for (uint32_t k = (uint32_t)lane; k < COUNTER_SLOTS; k += 64u) {
#pragma unroll
        for (uint32_t c = 0; c < COUNTER_K1_U64; ++c) tot[c] += c64[COUNTER_K1_U64 * k + c];
        const uint32_t* fb = set + COUNTER_FB_WORD + 3u * k;
        need = max(need, fb[0]);
        pairs += fb[1];
        tiles += fb[2];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (uint32_t c = 0; c < COUNTER_K1_U64; ++c) tot[c] += __shfl_down(tot[c], off);
        need = max(need, (uint32_t)__shfl_down((int)need, off));
        pairs += __shfl_down(pairs, off);
        tiles += __shfl_down(tiles, off);
    }
    if (lane == 0) {
        // (a copy for the kernels queued behind the sort before the host has read anything: the list builder takes the number of
        //  listed splats from here, api.hip "speculative K5")
        if (dev_sums) {
#pragma unroll
            for (uint32_t c = 0; c < COUNTER_K1_U64; ++c) dev_sums[c] = tot[c] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot[c];
        }
        volatile unsigned long long* h64 = reinterpret_cast<volatile unsigned long long*>(host_sums);
#pragma unroll
        for (uint32_t c = 0; c < COUNTER_K1_U64; ++c) h64[c] = tot[c];
        volatile uint32_t* h32 = host_sums + 2u * COUNTER_K1_U64;
        h32[0] = need;
        h32[1] = pairs;
        h32[2] = tiles;
        // tag != 0: the host does not wait for an event behind this kernel (a barrier packet: ~6 us of bubble in front of the next
        // kernel) but polls this word — stored last, behind a system-scope fence, so the sums are there when it is
        if (tag) {
            __threadfence_system();
            h32[3] = tag;
        }
    }
}

// ---- 1a: histogram of the split digit per block + tile-count sums per (digit, block) ------------------------------------
// (block `nblocks`, present when rb_set is given, carries the counter readback instead of a chunk of keys)
__global__ __launch_bounds__(DS_WG) void dsort_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ counts,
                                                          uint32_t n, uint32_t nblocks, const uint32_t* __restrict__ minmax,
                                                          uint32_t* __restrict__ hist, uint32_t* __restrict__ csum,
                                                          const uint32_t* __restrict__ rb_set, uint32_t* __restrict__ rb_host, uint32_t rb_tag, uint32_t* __restrict__ rb_dev,
                                                          const uint32_t* __restrict__ spl_in, uint8_t* __restrict__ digits /*[n]: every key's split digit, for the split kernel*/) {
    __shared__ uint32_t s_hist[DS_WAVES][DS_RADIX];
    __shared__ uint32_t s_csum[DS_WAVES][DS_RADIX];
    __shared__ uint32_t s_red[2 * DS_WAVES];
    __shared__ uint32_t s_spl[SPL_WORDS];
    const int tid = threadIdx.x, wave = tid >> 6;
    if (blockIdx.x == nblocks) {   // (block-uniform: taken before the first barrier)
        if (wave == 0) counter_sums_to_host(rb_set, rb_host, tid, rb_tag, rb_dev);
        return;
    }
    for (int i = tid; i < DS_WAVES * DS_RADIX; i += DS_WG) { (&s_hist[0][0])[i] = 0; (&s_csum[0][0])[i] = 0; }
    const DepthSplit sp = depth_split(minmax, s_red, spl_in, s_spl);   // (contains the barrier behind the clears)
    const uint32_t base = blockIdx.x * DS_TILE;
    // culled keys (digit 255) are neither counted nor moved: nothing reads the order's tail behind the visible splats, and with
    // per-tile cuts three quarters of the keys are culled — their LDS atomics would all land on one counter
    auto count_one = [&](uint32_t key, uint32_t cnt) {
        const uint32_t d = depth_digit(key, sp);
        if (d != 255u) {
            atomicAdd(&s_hist[wave][d], 1u);
            atomicAdd(&s_csum[wave][d], cnt);
        }
        return d;
    };
    if (base + (uint32_t)DS_TILE <= n && ((reinterpret_cast<uintptr_t>(keys) | reinterpret_cast<uintptr_t>(counts)) & 15u) == 0) {
        // every chunk but the last: 16-byte loads (which thread counts which key is irrelevant to a histogram)
        const uint4* k4 = reinterpret_cast<const uint4*>(keys + base);
        const uint4* c4 = reinterpret_cast<const uint4*>(counts + base);
        uint4 kv[DS_KPT / 4], cv[DS_KPT / 4];
#pragma unroll
        for (int k = 0; k < DS_KPT / 4; ++k) { kv[k] = k4[k * DS_WG + tid]; cv[k] = c4[k * DS_WG + tid]; }
        // (the digits leave as bytes, four per store: the split kernel reads them back instead of searching the splitters twice per key)
        uint32_t* d4 = reinterpret_cast<uint32_t*>(digits + base);
#pragma unroll
        for (int k = 0; k < DS_KPT / 4; ++k) {
            const uint32_t d0 = count_one(kv[k].x, cv[k].x);
            const uint32_t d1 = count_one(kv[k].y, cv[k].y);
            const uint32_t d2 = count_one(kv[k].z, cv[k].z);
            const uint32_t d3 = count_one(kv[k].w, cv[k].w);
            d4[k * DS_WG + tid] = d0 | (d1 << 8) | (d2 << 16) | (d3 << 24);
        }
    } else {
#pragma unroll
        for (int k = 0; k < DS_KPT; ++k) {
            const uint32_t idx = base + k * DS_WG + tid;
            if (idx < n) digits[idx] = (uint8_t)count_one(keys[idx], counts[idx]);
        }
    }
    __syncthreads();
    uint32_t t = 0, c = 0;
#pragma unroll
    for (int w = 0; w < DS_WAVES; ++w) { t += s_hist[w][tid]; c += s_csum[w][tid]; }
    hist[(size_t)tid * nblocks + blockIdx.x] = t;
    csum[(size_t)tid * nblocks + blockIdx.x] = c;
}

// ---- 1b: per digit row: exclusive scan of the block histogram (-> scatter offsets), row totals of both tables ----------
constexpr int DS_ROW_EPT = 16;   // rows of up to 4096 blocks (16.7 M splats); dsort_supported() guards it
__global__ __launch_bounds__(DS_WG) void dsort_rowscan_kernel(uint32_t* __restrict__ hist, const uint32_t* __restrict__ csum, uint32_t nblocks,
                                                             uint32_t* __restrict__ digit_totals /*[256] keys | [256] tile counts*/) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t d = blockIdx.x;
    uint32_t* row = hist + (size_t)d * nblocks;
    const uint32_t* crow = csum + (size_t)d * nblocks;
    uint32_t v[DS_ROW_EPT], incl[DS_ROW_EPT];
    __shared__ uint32_t s_chunk[DS_ROW_EPT][DS_WAVES];
    __shared__ uint32_t s_c[DS_WAVES];
    uint32_t cacc = 0;
#pragma unroll
    for (int k = 0; k < DS_ROW_EPT; ++k) {
        const uint32_t i = (uint32_t)k * DS_WG + tid;
        v[k] = i < nblocks ? row[i] : 0u;
        cacc += i < nblocks ? crow[i] : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cacc += __shfl_down(cacc, off);
    if (lane == 0) s_c[wave] = cacc;
#pragma unroll
    for (int k = 0; k < DS_ROW_EPT; ++k) {
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
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < DS_ROW_EPT; ++k) {
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < DS_WAVES; ++w) {
            const uint32_t c = s_chunk[k][w];
            before += w < wave ? c : 0u;
            total += c;
        }
        const uint32_t i = (uint32_t)k * DS_WG + tid;
        if (i < nblocks) row[i] = run + before + incl[k] - v[k];
        run += total;
    }
    if (tid == 0) {
        digit_totals[d] = run;
        digit_totals[DS_RADIX + d] = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]);
    }
}

// One chunk (<= 4096 elements, in index order) of a stable scatter by digit: element e goes to dst[s_base[digit] + (number of
// elements with the same digit in front of it in this chunk)], and s_base advances by the chunk's digit counts.  The chunk is
// re-ordered through LDS first, so every digit run leaves as one contiguous, coalesced burst.
template <int WG>
struct ChunkLds {
    uint32_t cnt[WG / 64][DS_RADIX];    // per-wave digit counts -> wave bases
    uint32_t dbase[DS_RADIX];           // exclusive scan of the chunk's digit counts
    uint32_t gofs[DS_RADIX];            // destination of the chunk's first element of a digit, minus dbase
    uint32_t wsum[WG / 64];
    uint32_t keys[DS_TILE];
    uint32_t vals[DS_TILE];
    uint8_t digs[DS_TILE];              // (callers with precomputed digits: the re-ordered elements' digits)
};
// WG threads (a multiple of 256): thread (wave w, lane l) ranks elements w * 64 * KPT + k * 64 + l; digit `tid` of the 256 is
// looked after by thread `tid` (threads >= 256 only rank and move).
template <int WG, class DigitFn>
BH_DEV void scatter_chunk(ChunkLds<WG>& L, uint32_t* __restrict__ s_base /*[256] running destinations (LDS)*/, const uint32_t* __restrict__ src_k,
                          const uint32_t* __restrict__ src_v /*NULL: the element index*/, uint32_t first, uint32_t count, uint32_t* __restrict__ dst_k,
                          uint32_t* __restrict__ dst_v, DigitFn digit, uint32_t* __restrict__ alt_k = nullptr, uint32_t* __restrict__ alt_v = nullptr,
                          const uint8_t* __restrict__ pre_dig = nullptr /*[.. first + count): the elements' digits, already computed (block-uniform)*/) {
    constexpr int WAVES = WG / 64, KPT = DS_TILE / WG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < WAVES * DS_RADIX; i += WG) (&L.cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t wave_first = wave * (64 * KPT);
    uint32_t key[KPT], val[KPT], rank[KPT], dig[KPT];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    if (count == (uint32_t)DS_TILE) {   // a full chunk: no bounds tests around the loads
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t e = wave_first + k * 64 + lane;
            key[k] = src_k[first + e];
            val[k] = src_v ? src_v[first + e] : first + e;
        }
        if (pre_dig) {
#pragma unroll
            for (int k = 0; k < KPT; ++k) dig[k] = pre_dig[first + wave_first + k * 64 + lane];
        } else {
#pragma unroll
            for (int k = 0; k < KPT; ++k) dig[k] = digit(key[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t e = wave_first + k * 64 + lane;
            const bool valid = e < count;
            key[k] = valid ? src_k[first + e] : 0u;
            val[k] = valid ? (src_v ? src_v[first + e] : first + e) : 0u;
            // invalid tail elements take digit 255 and sit at the highest in-chunk positions: they never disturb a valid rank
            dig[k] = valid ? (pre_dig ? (uint32_t)pre_dig[first + e] : digit(key[k])) : 255u;
        }
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const unsigned long long peers = ds_match(dig[k]);
        const uint32_t prior = L.cnt[wave][dig[k]];
        rank[k] = prior + (uint32_t)__popcll(peers & lt_mask);
        // all lanes have read `prior` (a wave executes in lock-step and its LDS operations retire in order) before the
        // leader of each digit group bumps the count
        if ((peers & lt_mask) == 0ull) L.cnt[wave][dig[k]] = prior + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    uint32_t total = 0;
    if (tid < DS_RADIX) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = L.cnt[w][tid];
            L.cnt[w][tid] = run;
            run += c;
        }
        total = run;   // elements of digit `tid` in this chunk (incl. the invalid tail under 255)
    }
    {
        uint32_t incl = total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) L.wsum[wave] = incl;
        __syncthreads();
        if (tid < DS_RADIX) {
            uint32_t wofs = 0;
#pragma unroll
            for (int w = 0; w < DS_RADIX / 64; ++w) wofs += (w < wave) ? L.wsum[w] : 0u;
            const uint32_t excl = incl - total + wofs;
            L.dbase[tid] = excl;
            const uint32_t dst0 = s_base[tid];
            L.gofs[tid] = dst0 - excl;
            // the invalid tail is counted under digit 255 but never written: only valid elements advance the base
            const uint32_t tail = (uint32_t)DS_TILE - count;
            s_base[tid] = dst0 + total - (tid == 255 ? tail : 0u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const uint32_t lpos = L.dbase[dig[k]] + L.cnt[wave][dig[k]] + rank[k];
        L.keys[lpos] = key[k];
        L.vals[lpos] = val[k];
        if (pre_dig) L.digs[lpos] = (uint8_t)dig[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const uint32_t e = k * WG + tid;
        if (e < count) {   // valid elements occupy the first `count` re-ordered positions (the tail sorts last)
            const uint32_t kk = L.keys[e];
            const uint32_t d = pre_dig ? (uint32_t)L.digs[e] : digit(kk);
            const uint32_t pos = L.gofs[d] + e;
            // alt_* != NULL (the depth split): digit 255 = the culled splats, whose place in the order nobody reads — not written at all
            if (alt_k != nullptr && d == 255u) continue;
            dst_k[pos] = kk;
            dst_v[pos] = L.vals[e];
        }
    }
    __syncthreads();
}

// ---- 1c: the split itself: stable scatter of (key, splat id) by split digit --------------------------------------------------
__global__ __launch_bounds__(DS_WG) void dsort_split_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t nblocks,
                                                           const uint32_t* __restrict__ offsets /*row-scanned hist*/,
                                                           const uint32_t* __restrict__ digit_totals, uint32_t* __restrict__ out_keys,
                                                           uint32_t* __restrict__ out_vals, uint32_t* __restrict__ fin_keys, uint32_t* __restrict__ fin_vals,
                                                           const uint8_t* __restrict__ digits /*[n] the histogram kernel's*/) {
    __shared__ ChunkLds<DS_WG> L;
    __shared__ uint32_t s_base[DS_RADIX];
    __shared__ uint32_t s_red[2 * DS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // destination of this block's first element of digit `tid`: keys with a smaller digit + this digit's keys in earlier blocks
    {
        const uint32_t gt = digit_totals[tid];
        uint32_t gincl = gt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t g = __shfl_up(gincl, off);
            if (lane >= off) gincl += g;
        }
        if (lane == 63) s_red[wave] = gincl;
        __syncthreads();
        uint32_t gofs = 0;
#pragma unroll
        for (int w = 0; w < DS_WAVES; ++w) gofs += (w < wave) ? s_red[w] : 0u;
        s_base[tid] = gincl - gt + gofs + offsets[(size_t)tid * nblocks + blockIdx.x];
    }
    __syncthreads();
    const uint32_t first = blockIdx.x * DS_TILE;
    const uint32_t count = n - first < (uint32_t)DS_TILE ? n - first : (uint32_t)DS_TILE;
    scatter_chunk<DS_WG>(L, s_base, keys, nullptr, first, count, out_keys, out_vals, [](uint32_t) { return 0u; }, fin_keys, fin_vals, digits);
}

// ---- 2: one block per bucket: sort on the low bits, then the scan of the tile counts --------------------------------------------
#define BH_BK_WG 1024   /* 16 waves: four per SIMD, the only block on its CU (LDS).  512 (rounds 2-3): 39.5 us, 1024: 34.0 us — the
                           kernel is one latency chain per bucket, the largest bucket sets its length, more waves shorten every link */
constexpr int BK_WG = BH_BK_WG;
constexpr int BK_WAVES = BK_WG / 64;
constexpr int BK_KPT = DS_TILE / BK_WG;        // scan / chunk granularity: 4096 elements, 8 per thread
constexpr int FAST_CAP = BK_WG <= 512 ? 8192 : 7168;   // keys of a bucket that is sorted resident in LDS (two (key, id) buffers: ping-pong)
constexpr int FAST_NDIG = 512;                 // digits of up to 9 bits
constexpr size_t FAST_LDS_WORDS = 4 * FAST_CAP + (BK_WAVES + 1) * FAST_NDIG;
static_assert(FAST_LDS_WORDS * 4 >= sizeof(ChunkLds<BK_WG>), "the dynamic LDS block holds either path's arrays");
static_assert(offsetof(ChunkLds<BK_WG>, vals) == offsetof(ChunkLds<BK_WG>, keys) + sizeof(uint32_t) * DS_TILE, "scan staging spans keys[] and vals[]");

// block-wide exclusive offset of `v` (one value per thread) and the block total; s_w: [BK_WAVES] scratch
BH_DEV uint32_t bk_excl_scan(uint32_t v, uint32_t& total, uint32_t* s_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = __shfl_up(incl, off);
        if (lane >= off) incl += x;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t wofs = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BK_WAVES; ++w) { wofs += (w < wave) ? s_w[w] : 0u; tot += s_w[w]; }
    total = tot;
    __syncthreads();
    return incl - v + wofs;
}

// inclusive scan of counts[vals[i]] over one bucket (render.rs:185-187), 4096 elements at a time with a running carry; `stage`:
// 4096 + 512 words of LDS; vals: the bucket's sorted splat ids (LDS or global)
BH_DEV void bk_scan_counts(const uint32_t* vals, uint32_t size, const uint32_t* __restrict__ counts, uint32_t carry, uint32_t* __restrict__ cum_out,
                           uint32_t* stage, uint32_t* s_w) {
    const int tid = threadIdx.x;
    const uint32_t nch = (size + DS_TILE - 1) / DS_TILE;
    for (uint32_t c = 0; c < nch; ++c) {
        const uint32_t first = c * DS_TILE;
        const uint32_t count = size - first < (uint32_t)DS_TILE ? size - first : (uint32_t)DS_TILE;
        // thread t owns elements 8t .. 8t+7 of the chunk, staged through LDS (+1 word per thread run: conflict-free) so the global side stays coalesced
#pragma unroll
        for (int k = 0; k < BK_KPT; ++k) {
            const uint32_t e = k * BK_WG + tid;
            stage[e + e / BK_KPT] = e < count ? counts[vals[first + e]] : 0u;
        }
        __syncthreads();
        uint32_t v[BK_KPT], tsum = 0;
#pragma unroll
        for (int k = 0; k < BK_KPT; ++k) { v[k] = stage[tid * (BK_KPT + 1) + k]; tsum += v[k]; }
        uint32_t total;
        uint32_t run = carry + bk_excl_scan(tsum, total, s_w);
#pragma unroll
        for (int k = 0; k < BK_KPT; ++k) { run += v[k]; stage[tid * (BK_KPT + 1) + k] = run; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BK_KPT; ++k) {
            const uint32_t e = k * BK_WG + tid;
            if (e < count) cum_out[first + e] = stage[e + e / BK_KPT];
        }
        carry += total;
        __syncthreads();
    }
}

// Block-wide smallest key (-> lo) and the width in bits of the bucket's key span, from every thread's (min, max); s_w: [2 * BK_WAVES].
// (Taken from the keys themselves, not from the split: a bucket of a crowded depth range is narrower than its share of the frame's
// range and needs fewer counting passes.)
BH_DEV uint32_t bk_key_bits(uint32_t& lo, uint32_t hi, uint32_t* s_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
    }
    if (lane == 0) { s_w[wave] = lo; s_w[BK_WAVES + wave] = hi; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < BK_WAVES; ++w) { lo = min(lo, s_w[w]); hi = max(hi, s_w[BK_WAVES + w]); }
    __syncthreads();
    return hi > lo ? 32u - (uint32_t)__clz(hi - lo) : 0u;
}

// spl_out (optional): the next frame's splitter table (SPLITTERS above) — the block that holds rank (j + 1) Nv / 255 writes splitter j.
BH_DEV void bk_write_splitters(const uint32_t* sorted /*the bucket's keys, ascending (LDS or global)*/, uint32_t start, uint32_t size, uint32_t nv,
                               uint32_t* __restrict__ spl_out, uint32_t spl_min_keys) {
    const uint32_t j = threadIdx.x;
    if (spl_out == nullptr || nv < spl_min_keys || j >= 254u) return;
    const uint32_t r = (uint32_t)(((unsigned long long)(j + 1u) * nv) / 255ull);
    if (r >= start && r - start < size) spl_out[j] = sorted[r - start];
}

__global__ __launch_bounds__(BK_WG) void dsort_bucket_kernel(uint32_t* __restrict__ a_keys, uint32_t* __restrict__ a_vals /*the split's output (scratch)*/,
                                                            uint32_t* __restrict__ out_keys, uint32_t* __restrict__ out_vals,
                                                            const uint32_t* __restrict__ minmax, const uint32_t* __restrict__ digit_totals,
                                                            const uint32_t* __restrict__ counts, uint32_t* __restrict__ cum /*NULL: no scan*/,
                                                            uint32_t* __restrict__ spl_out /*NULL: no table for the next frame*/, uint32_t spl_min_keys) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];   // FAST_LDS_WORDS: the resident path's arrays | the chunked path's ChunkLds
    __shared__ uint32_t s_base[DS_RADIX];
    __shared__ uint32_t s_hist[DS_RADIX];
    __shared__ uint32_t s_red[2 * BK_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = blockIdx.x;
    // where this bucket starts, and how many tiles the buckets in front of it hit
    uint32_t start, size, tiles_before, nv;
    {
        const uint32_t gt = tid < DS_RADIX ? digit_totals[tid] : 0u;
        const uint32_t ct = tid < 255 ? digit_totals[DS_RADIX + tid] : 0u;
        uint32_t tot;
        const uint32_t gex = bk_excl_scan(gt, nv, s_red);   // (digit 255 is never counted: the total is the number of visible keys)
        const uint32_t cex = bk_excl_scan(ct, tot, s_red);
        if (tid < DS_RADIX) { s_base[tid] = gex; s_hist[tid] = cex; }
        __syncthreads();
        start = s_base[b];
        size = s_base[b + 1u] - start;     // (b <= 254)
        tiles_before = s_hist[b];
        __syncthreads();
    }
    if (b == 0u && spl_out != nullptr) {   // the table's sentinel, its validity word and the frame's key range (block-uniform branch)
        uint32_t x = 0, y = 0;
        if (tid < (int)COUNTER_SLOTS) { x = minmax[2 * tid]; y = minmax[2 * tid + 1]; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            x = max(x, (uint32_t)__shfl_xor((int)x, off));
            y = max(y, (uint32_t)__shfl_xor((int)y, off));
        }
        if (lane == 0) { s_red[wave] = x; s_red[BK_WAVES + wave] = y; }   // (waves >= 2 hold zeros: neutral)
        __syncthreads();
        if (tid == 0) {
            uint32_t kmax = 0, nmin = 0;
            for (int w = 0; w < BK_WAVES; ++w) { kmax = max(kmax, s_red[w]); nmin = max(nmin, s_red[BK_WAVES + w]); }
            spl_out[254] = 0xFFFFFFFFu;
            spl_out[255] = nv >= spl_min_keys ? 1u : 0u;
            spl_out[256] = ~nmin;
            spl_out[257] = kmax;
        }
        __syncthreads();
    }
    if (size == 0u) return;
    if (size <= (uint32_t)FAST_CAP) {
        // ---- resident path (the usual case): the whole bucket lives in LDS and is sorted there by stable counting passes on
        // (key - the bucket's smallest key), digits of up to 9 bits, ping-pong between two (key, splat id) buffers; global
        // memory is touched once on the way in and once on the way out.  Rolled loops over 64-element rows (a wave owns a
        // contiguous run of rows: stable): a pass is (a) digit counts per wave by LDS atomics, (b) scan over waves and digits,
        // (c) placement: the rank inside a row from a wave-wide digit match (ballots), the row's base from a running per-wave
        // cursor per digit.
        // buffer c: keys at s_dyn + 2 c FAST_CAP, splat ids FAST_CAP behind them
        uint32_t* fcnt = s_dyn + 4 * FAST_CAP;                 // [BK_WAVES][FAST_NDIG]
        uint32_t* fdb = fcnt + BK_WAVES * FAST_NDIG;           // [FAST_NDIG]
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (uint32_t i = tid; i < size; i += BK_WG) {
            const uint32_t k = a_keys[start + i];
            s_dyn[i] = k;
            s_dyn[FAST_CAP + i] = a_vals[start + i];
            lo = min(lo, k);
            hi = max(hi, k);
        }
        const uint32_t bits = bk_key_bits(lo, hi, s_red);      // the bucket's keys span less than 2^bits (0: one distinct key — nothing to sort)
        const uint32_t passes = (bits + 8u) / 9u;              // 0 for bits == 0
        const uint32_t base_w = passes ? bits / passes : 0u, wide = passes ? bits % passes : 0u;
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        const uint32_t steps = (size + (uint32_t)BK_WG - 1u) / (uint32_t)BK_WG;   // rows per wave
        const uint32_t wave_first = wave * steps * 64u;
        uint32_t cur = 0;
        for (uint32_t p = 0; p < passes; ++p) {
            const uint32_t width = base_w + (p < wide ? 1u : 0u);
            const uint32_t pshift = p * base_w + (p < wide ? p : wide);
            const uint32_t mask = (1u << width) - 1u;
            const uint32_t* sk = s_dyn + cur * (2 * FAST_CAP); const uint32_t* sv = sk + FAST_CAP;
            uint32_t* dk = s_dyn + (cur ^ 1u) * (2 * FAST_CAP); uint32_t* dv = dk + FAST_CAP;
            for (int i = tid; i < BK_WAVES * FAST_NDIG; i += BK_WG) fcnt[i] = 0;
            __syncthreads();
            // (a) counts per (wave, digit)
            for (uint32_t r = 0; r < steps; ++r) {
                const uint32_t e = wave_first + r * 64u + lane;
                if (e < size) atomicAdd(&fcnt[wave * FAST_NDIG + (((sk[e] - lo) >> pshift) & mask)], 1u);
            }
            __syncthreads();
            // (b) per digit (one per thread): wave cursors; exclusive scan of the digit totals
            {
                uint32_t run = 0;
                if (tid < FAST_NDIG) {
#pragma unroll
                    for (int w = 0; w < BK_WAVES; ++w) {
                        const uint32_t c = fcnt[w * FAST_NDIG + tid];
                        fcnt[w * FAST_NDIG + tid] = run;
                        run += c;
                    }
                }
                uint32_t tot;
                const uint32_t ex = bk_excl_scan(run, tot, s_red);
                if (tid < FAST_NDIG) fdb[tid] = ex;
            }
            __syncthreads();
            // (c) placement
            for (uint32_t r = 0; r < steps; ++r) {
                const uint32_t e = wave_first + r * 64u + lane;
                const bool valid = e < size;
                const uint32_t key = valid ? sk[e] : 0u, val = valid ? sv[e] : 0u;
                // (the row's tail takes digit `mask`: it ranks behind every valid element of that digit and is never written)
                const uint32_t d = valid ? ((key - lo) >> pshift) & mask : mask;
                // lanes of the row with my digit: AND over the bits of (ballot of the bit) XNOR (my bit)
                uint32_t plo = 0xFFFFFFFFu, phi = 0xFFFFFFFFu;
                for (uint32_t bb = 0; bb < width; ++bb) {
                    const uint32_t bit = (d >> bb) & 1u;
                    const unsigned long long bal = __ballot(bit);
                    const uint32_t flip = bit - 1u;          // bit set: 0, clear: ~0
                    plo &= (uint32_t)bal ^ flip;
                    phi &= (uint32_t)(bal >> 32) ^ flip;
                }
                const unsigned long long peers = ((unsigned long long)phi << 32) | plo;
                const uint32_t cursor = fcnt[wave * FAST_NDIG + d];
                const uint32_t pos = fdb[d] + cursor + (uint32_t)__popcll(peers & lt_mask);
                // every lane has read the cursor (a wave's LDS operations retire in order) before the group's first lane advances it
                if ((peers & lt_mask) == 0ull) fcnt[wave * FAST_NDIG + d] = cursor + (uint32_t)__popcll(peers);
                if (valid) { dk[pos] = key; dv[pos] = val; }
            }
            __syncthreads();
            cur ^= 1u;
        }
        const uint32_t* fk = s_dyn + cur * (2 * FAST_CAP); const uint32_t* fv = fk + FAST_CAP;
        uint32_t* spare = s_dyn + (cur ^ 1u) * (2 * FAST_CAP);   // the other buffer: 2 * FAST_CAP words
        bk_write_splitters(fk, start, size, nv, spl_out, spl_min_keys);
        // on the way out: the tile count of every splat, all gathers in flight at once
        for (uint32_t i = tid; i < size; i += BK_WG) {
            const uint32_t v = fv[i];
            out_keys[start + i] = fk[i];
            out_vals[start + i] = v;
            if (cum != nullptr) { const uint32_t e = i; spare[e + e / BK_KPT] = counts[v]; }
        }
        if (cum == nullptr) return;
        __syncthreads();
        // inclusive scan of the tile counts in sorted order (render.rs:185-187): thread t owns 8 consecutive elements of each 4096-run
        {
            uint32_t carry = tiles_before;
            const uint32_t nch = (size + DS_TILE - 1) / DS_TILE;
            for (uint32_t c = 0; c < nch; ++c) {
                const uint32_t first = c * DS_TILE;
                uint32_t v[BK_KPT], tsum = 0;
#pragma unroll
                for (int k = 0; k < BK_KPT; ++k) {
                    const uint32_t e = first + tid * BK_KPT + k;
                    v[k] = e < size ? spare[e + e / BK_KPT] : 0u;
                    tsum += v[k];
                }
                uint32_t total;
                uint32_t run = carry + bk_excl_scan(tsum, total, s_red);
#pragma unroll
                for (int k = 0; k < BK_KPT; ++k) {
                    const uint32_t e = first + tid * BK_KPT + k;
                    run += v[k];
                    if (e < size) spare[e + e / BK_KPT] = run;
                }
                carry += total;
            }
            __syncthreads();
            for (uint32_t i = tid; i < size; i += BK_WG) cum[start + i] = spare[i + i / BK_KPT];
        }
        return;
    }
    // ---- chunked path: a bucket of any size (depth distributions that defeat the range split), through global scratch:
    // stable LSD passes scratch -> out -> scratch -> out, 4096 keys at a time, running digit bases carried from chunk to chunk.
    // An ODD number of passes lands in `out` (0 bits: one pass of width 0 = a stable copy).
    ChunkLds<BK_WG>& L = *reinterpret_cast<ChunkLds<BK_WG>*>(s_dyn);
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t i = tid; i < size; i += BK_WG) {
        const uint32_t k = a_keys[start + i];
        lo = min(lo, k);
        hi = max(hi, k);
    }
    const uint32_t bits = bk_key_bits(lo, hi, s_red);
    // (digits of at most 8 bits — the 256-entry tables — and an odd count: 1 pass up to 8 bits, 3 up to 24, 5 beyond.  K1 rejects
    //  |z| > 1e10, which keeps a bucket's span below 2^24 today; the 5-pass case is there so that nothing depends on it.)
    const uint32_t passes = bits <= 8u ? 1u : (bits <= 24u ? 3u : 5u);
    const uint32_t base_w = bits / passes, wide = bits % passes;
    uint32_t* src_k = a_keys; uint32_t* src_v = a_vals;
    uint32_t* dst_k = out_keys; uint32_t* dst_v = out_vals;
    const uint32_t nchunks = (size + DS_TILE - 1) / DS_TILE;
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t width = base_w + (p < wide ? 1u : 0u);
        const uint32_t pshift = p * base_w + (p < wide ? p : wide);
        const uint32_t mask = (1u << width) - 1u;
        auto digit = [=](uint32_t k) { return ((k - lo) >> pshift) & mask; };
        // digit counts of the whole bucket -> running destinations
        if (tid < DS_RADIX) s_hist[tid] = 0;
        __syncthreads();
        if (width != 0u) {
            for (uint32_t i = tid; i < size; i += BK_WG) atomicAdd(&s_hist[digit(src_k[start + i])], 1u);
        } else if (tid == 0) {
            s_hist[0] = size;
        }
        __syncthreads();
        {
            uint32_t tot;
            const uint32_t ex = bk_excl_scan(tid < DS_RADIX ? s_hist[tid] : 0u, tot, s_red);
            if (tid < DS_RADIX) s_base[tid] = start + ex;
        }
        __syncthreads();
        for (uint32_t c = 0; c < nchunks; ++c) {
            const uint32_t first = start + c * DS_TILE;
            const uint32_t count = size - c * DS_TILE < (uint32_t)DS_TILE ? size - c * DS_TILE : (uint32_t)DS_TILE;
            scatter_chunk<BK_WG>(L, s_base, src_k, src_v, first, count, dst_k, dst_v, digit);
        }
        // the next pass reads what this one wrote (other threads' global stores): make them visible to the block
        __threadfence_block();
        __syncthreads();
        uint32_t* tk = src_k; src_k = dst_k; dst_k = tk;
        uint32_t* tv = src_v; src_v = dst_v; dst_v = tv;
    }
    bk_write_splitters(out_keys + start, start, size, nv, spl_out, spl_min_keys);
    if (cum != nullptr) bk_scan_counts(out_vals + start, size, counts, tiles_before, cum + start, &L.keys[0], s_red);
}

// Up to the row scan's reach (4096 chunks of DS_TILE keys: 8.4 M splats); beyond it the generic sort + scan run.  (Until round 4 the limit was 4 M:
// "the fused path pays off while launches, not bytes, are the cost".  With per-tile cuts only the listed sixth of the 6 M / 4K
// scene's splats is moved at all: depth order + scan 255 -> 97 us there; with complete lists 306 -> 302.)
#define BH_DSORT_MAX_N (16u << 20)
constexpr uint32_t DSORT_MAX_N = BH_DSORT_MAX_N;
bool depth_sort_supported(uint32_t n) { return n > 0 && n <= DSORT_MAX_N && (n + DS_TILE - 1) / DS_TILE <= (uint32_t)DS_ROW_EPT * DS_WG; }

// keys: [n] depth keys (culled = 0xFFFFFFFF); minmax: K1's [COUNTER_SLOTS][2] (max key, max ~key over visible splats);
// counts: [n] tiles hit per splat.  -> out_keys / out_vals: the stable argsort; cum: inclusive scan of counts[out_vals[i]]
// over the visible prefix (entries behind it are not written).
// rb_set / rb_host / rb_done (optional, all or none): the counter set whose sums the first kernel stores into the pinned host block
// rb_host, and the event recorded right behind that kernel (counter_sums_to_host above) — or, rb_tag != 0, no event: the kernel
// stores the tag behind the sums and the host polls for it.
// spl (optional): the view's splitter table [DSORT_SPL_STRIDE] for this list mode (SPLITTERS above) and the host's note whether a frame has
// written it yet; NULL: the ctx's.
__global__ __launch_bounds__(256) void dsort_sample_kernel(const uint32_t* __restrict__ keys, uint32_t stride, uint32_t m, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < m) out[i] = keys[(size_t)i * stride];
}

// the four launches; scratch sized by the caller
static int dsort_launches(bh_ctx* ctx, const uint32_t* keys, const uint32_t* minmax, const uint32_t* counts, uint32_t n, uint32_t* out_keys, uint32_t* out_vals,
                          uint32_t* cum, const uint32_t* rb_set, uint32_t* rb_host, hipEvent_t rb_done, uint32_t rb_tag, uint32_t* rb_dev, uint32_t* totals,
                          uint32_t* a_keys, uint32_t* a_vals, const uint32_t* spl_in, uint32_t* spl_out, uint32_t spl_min_keys) {
    const uint32_t nblocks = (n + DS_TILE - 1) / DS_TILE;
    uint32_t* hist = totals + 2 * DS_RADIX;
    uint32_t* csum = hist + (size_t)DS_RADIX * nblocks;
    uint8_t* digits = reinterpret_cast<uint8_t*>(csum + (size_t)DS_RADIX * nblocks);   // [nblocks * DS_TILE] every key's split digit (hist -> split)
    hipLaunchKernelGGL(dsort_hist_kernel, dim3(nblocks + (rb_set ? 1u : 0u)), dim3(DS_WG), 0, ctx->stream, keys, counts, n, nblocks, minmax, hist, csum,
                       rb_set, rb_host, rb_tag, rb_dev, spl_in, digits);
    BH_LAUNCH_CHECK(ctx, "dsort_hist_kernel");
    if (rb_set && !rb_tag) BH_HIP(ctx, hipEventRecord(rb_done, ctx->stream));   // (rb_tag: the host polls the tag word instead)
    hipLaunchKernelGGL(dsort_rowscan_kernel, dim3(DS_RADIX), dim3(DS_WG), 0, ctx->stream, hist, csum, nblocks, totals);
    BH_LAUNCH_CHECK(ctx, "dsort_rowscan_kernel");
    hipLaunchKernelGGL(dsort_split_kernel, dim3(nblocks), dim3(DS_WG), 0, ctx->stream, keys, n, nblocks, hist, totals, a_keys, a_vals, out_keys, out_vals, digits);
    BH_LAUNCH_CHECK(ctx, "dsort_split_kernel");
    // buckets 0..254; the culled splats (digit 255: all keys 0xFFFFFFFF, already in splat-id order) went straight to the output
    if (!ctx->dsort_lds_raised) {   // 146 KB of dynamic LDS: above the 64 KB default, opt in once per ctx (the attribute is per device)
        BH_HIP(ctx, hipFuncSetAttribute((const void*)dsort_bucket_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FAST_LDS_WORDS * 4)));
        ctx->dsort_lds_raised = true;
    }
    hipLaunchKernelGGL(dsort_bucket_kernel, dim3(DS_RADIX - 1), dim3(BK_WG), FAST_LDS_WORDS * 4, ctx->stream, a_keys, a_vals, out_keys, out_vals, minmax, totals, counts, cum, spl_out,
                       spl_min_keys);
    BH_LAUNCH_CHECK(ctx, "dsort_bucket_kernel");
    return 0;
}

int depth_sort_scan(bh_ctx* ctx, const uint32_t* keys, const uint32_t* minmax, const uint32_t* counts, uint32_t n, uint32_t* out_keys,
                    uint32_t* out_vals, uint32_t* cum, const uint32_t* rb_set, uint32_t* rb_host, hipEvent_t rb_done, uint32_t rb_tag, uint32_t* rb_dev, uint32_t* spl,
                    bool* spl_written) {
    if (n == 0) return 0;
    const uint32_t nblocks = (n + DS_TILE - 1) / DS_TILE;
    // the splitter table (SPLITTERS above): the caller's (a view's) or, for a frame without a view, the ctx's own
    if (!ctx->knob_dsort_splitters) spl = nullptr;
    else if (spl == nullptr) {
        if (ctx->dsort_spl == nullptr) {
            if (hipMalloc((void**)&ctx->dsort_spl, DSORT_SPL_STRIDE * 4) != hipSuccess) { (void)hipGetLastError(); ctx->dsort_spl = nullptr; return BH_ERR_OOM; }
            BH_HIP(ctx, hipMemsetAsync(ctx->dsort_spl, 0, DSORT_SPL_STRIDE * 4, ctx->stream));
            ctx->dsort_spl_written = false;
        }
        spl = ctx->dsort_spl;
        spl_written = &ctx->dsort_spl_written;
    }
    // [512] digit totals (keys | tile counts), then the two [256][nblocks] tables, then the keys' digits (bytes)
    uint32_t* totals = (uint32_t*)ensure(ctx, SLOT_SORT_HIST, ((size_t)2 * DS_RADIX * nblocks + 2 * DS_RADIX) * 4 + (size_t)nblocks * DS_TILE);
    uint32_t* a_keys = (uint32_t*)ensure(ctx, SLOT_SORT_KEYS_A, (size_t)n * 4);
    uint32_t* a_vals = (uint32_t*)ensure(ctx, SLOT_SORT_VALS_A, (size_t)n * 4);
    if (!totals || !a_keys || !a_vals) return BH_ERR_OOM;
    if (spl && spl_written && !*spl_written && n >= SPL_SAMPLE_MIN_N) {
        // no table yet: the sample's quantiles first (see SPL_SAMPLE_STRIDE).  Its sorted output lands in the real run's output buffers,
        // which the real run overwrites; its "tile counts" are the sample keys themselves (no scan is asked for).
        const uint32_t m = n / SPL_SAMPLE_STRIDE;
        uint32_t* sample = (uint32_t*)ensure(ctx, SLOT_SORT_KEYS_B, (size_t)m * 4);
        if (!sample) return BH_ERR_OOM;
        hipLaunchKernelGGL(dsort_sample_kernel, dim3((m + 255u) / 256u), dim3(256), 0, ctx->stream, keys, SPL_SAMPLE_STRIDE, m, sample);
        BH_LAUNCH_CHECK(ctx, "dsort_sample_kernel");
        BH_TRY(dsort_launches(ctx, sample, minmax, sample, m, out_keys, out_vals, nullptr, nullptr, nullptr, nullptr, 0u, nullptr, totals, a_keys, a_vals, nullptr, spl,
                              SPL_SAMPLE_MIN_KEYS));
    }
    BH_TRY(dsort_launches(ctx, keys, minmax, counts, n, out_keys, out_vals, cum, rb_set, rb_host, rb_done, rb_tag, rb_dev, totals, a_keys, a_vals, spl, spl, SPL_MIN_KEYS));
    if (spl && spl_written) *spl_written = true;
    return 0;
}

}  // namespace bh
