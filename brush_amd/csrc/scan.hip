// scan.hip — u32 prefix sum (wrapping), inclusive or exclusive, with an optional
// gather on the input.
//
// Contract = brush_prefix_sum::prefix_sum (brush-prefix-sum/src/lib.rs:11-93,
// inclusive).  The reference runs a 512-wide Hillis-Steele per block with 18
// barriers and a recursive block-sum pyramid (kernels.rs:20-73).  Here: 4096
// elements per 256-thread block, 16 per thread in registers, wave64 shuffle scans,
// one spine pass over the block sums: reduce -> spine -> apply (12 B/element).  Up to SELF_SPINE_MAX blocks
// (4 M elements) the apply kernel sums the block totals in front of it itself — one coalesced load per
// thread — and the one-block spine launch (5 us of pure latency on this chip) disappears: reduce -> apply.
// The gather variant fuses `int_gather(intersect_counts, gid)` (render.rs:185).
#include "context.h"

namespace bh {

constexpr int SCAN_WG = 256;
constexpr int SCAN_EPT = 16;
constexpr int SCAN_TILE = SCAN_WG * SCAN_EPT;

BH_DEV uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// exclusive offset of this thread's value within the block, and the block total
BH_DEV uint32_t block_excl_scan(uint32_t v, uint32_t& block_total, uint32_t* s_wave /*[4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t ofs = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_WG / 64; ++w) {
        const uint32_t s = s_wave[w];
        ofs += w < wave ? s : 0u;
        tot += s;
    }
    block_total = tot;
    __syncthreads();
    return ofs + incl - v;
}

template <bool GATHER>
BH_DEV uint32_t load_elem(const uint32_t* __restrict__ in, const uint32_t* __restrict__ gather, uint32_t i) {
    return GATHER ? in[gather[i]] : in[i];
}

// GATHER: the gathered values are also left in `staged` (= the scan's output array) in index order, so that the apply pass reads
// them back coalesced instead of gathering a second time (6 M splats: the scan of the tile counts 170 -> ~90 us)
template <bool GATHER>
__global__ __launch_bounds__(SCAN_WG) void scan_reduce_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ gather,
                                                             uint32_t n, uint32_t* __restrict__ sums, const uint32_t* __restrict__ gate,
                                                             uint32_t* __restrict__ staged) {
    __shared__ uint32_t s_wave[SCAN_WG / 64];
    if (gate && *gate == 0u) return;   // (depth-sliced forward: nothing left to list, the result is never read)
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < SCAN_EPT; ++j) {
        const uint32_t i = base + j * SCAN_WG + threadIdx.x;  // coalesced; order is irrelevant for a sum
        if (i < n) {
            const uint32_t v = load_elem<GATHER>(in, gather, i);
            if (GATHER) staged[i] = v;
            acc += v;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < SCAN_WG / 64; ++w) t += s_wave[w];
        sums[blockIdx.x] = t;
    }
}

// single block: sums[i] <- exclusive prefix of sums
__global__ __launch_bounds__(SCAN_WG) void scan_spine_kernel(uint32_t* __restrict__ sums, uint32_t nb, const uint32_t* __restrict__ gate) {
    __shared__ uint32_t s_wave[SCAN_WG / 64];
    if (gate && *gate == 0u) return;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += SCAN_TILE) {
        uint32_t v[SCAN_EPT];
        uint32_t tsum = 0;
#pragma unroll
        for (int j = 0; j < SCAN_EPT; ++j) {
            const uint32_t i = base + threadIdx.x * SCAN_EPT + j;
            v[j] = i < nb ? sums[i] : 0u;
            tsum += v[j];
        }
        uint32_t total;
        uint32_t run = carry + block_excl_scan(tsum, total, s_wave);
#pragma unroll
        for (int j = 0; j < SCAN_EPT; ++j) {
            const uint32_t i = base + threadIdx.x * SCAN_EPT + j;
            if (i < nb) sums[i] = run;
            run += v[j];
        }
        carry += total;
    }
}

// +1 dword of padding per 16 so that "thread t owns elements 16t..16t+15" is a
// stride-17 (conflict-free) LDS access while the global side stays coalesced.
BH_DEV uint32_t scan_pad(uint32_t e) { return e + (e >> 4); }

constexpr uint32_t SELF_SPINE_MAX = 4 * SCAN_WG;

// SELF_SPINE: `sums` holds the raw block totals (gridDim.x <= SELF_SPINE_MAX); otherwise their exclusive prefix
template <bool GATHER, bool EXCLUSIVE, bool SELF_SPINE>
// (in may be out — the staged gather above: a block reads its whole tile before the barrier and writes it behind it; hence no __restrict__ on the two)
__global__ __launch_bounds__(SCAN_WG) void scan_apply_kernel(const uint32_t* in, const uint32_t* __restrict__ gather,
                                                            uint32_t n, const uint32_t* __restrict__ sums,
                                                            uint32_t* out, const uint32_t* __restrict__ gate) {
    __shared__ uint32_t s_wave[SCAN_WG / 64];
    __shared__ uint32_t s_tile[SCAN_TILE + SCAN_TILE / 16];
    if (gate && *gate == 0u) return;
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint32_t ahead = 0;   // this thread's share of the totals of the blocks in front
    if (SELF_SPINE) {
#pragma unroll
        for (uint32_t k = 0; k < SELF_SPINE_MAX / SCAN_WG; ++k) {
            const uint32_t i = k * SCAN_WG + threadIdx.x;
            if (i < blockIdx.x) ahead += sums[i];
        }
    }
#pragma unroll
    for (int j = 0; j < SCAN_EPT; ++j) {
        const uint32_t e = j * SCAN_WG + threadIdx.x;
        const uint32_t i = base + e;
        s_tile[scan_pad(e)] = i < n ? load_elem<GATHER>(in, gather, i) : 0u;
    }
    __syncthreads();
    uint32_t v[SCAN_EPT];
    uint32_t tsum = 0;
    const uint32_t own = threadIdx.x * (SCAN_EPT + 1);
#pragma unroll
    for (int j = 0; j < SCAN_EPT; ++j) {
        v[j] = s_tile[own + j];
        tsum += v[j];
    }
    uint32_t total;
    uint32_t carry;
    if (SELF_SPINE) {
        block_excl_scan(ahead, carry, s_wave);   // carry <- block-wide sum of `ahead`
    } else {
        carry = sums ? sums[blockIdx.x] : 0u;
    }
    uint32_t run = carry + block_excl_scan(tsum, total, s_wave);
#pragma unroll
    for (int j = 0; j < SCAN_EPT; ++j) {
        const uint32_t o = EXCLUSIVE ? run : run + v[j];
        run += v[j];
        s_tile[own + j] = o;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SCAN_EPT; ++j) {
        const uint32_t e = j * SCAN_WG + threadIdx.x;
        const uint32_t i = base + e;
        if (i < n) out[i] = s_tile[scan_pad(e)];
    }
}

template <bool GATHER>
static int scan_impl(bh_ctx* ctx, const uint32_t* in, const uint32_t* gather, uint32_t n, uint32_t* out, bool exclusive, const uint32_t* gate) {
    const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t* sums = nullptr;
    if (nb > 1) {
        sums = (uint32_t*)ensure(ctx, SLOT_SCAN_SUMS, (size_t)nb * 4);
        if (!sums) return BH_ERR_OOM;
        hipLaunchKernelGGL(scan_reduce_kernel<GATHER>, dim3(nb), dim3(SCAN_WG), 0, ctx->stream, in, gather, n, sums, gate, out);
        BH_LAUNCH_CHECK(ctx, "scan_reduce_kernel");
        if (nb > SELF_SPINE_MAX) {
            hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(SCAN_WG), 0, ctx->stream, sums, nb, gate);
            BH_LAUNCH_CHECK(ctx, "scan_spine_kernel");
        }
    }
    const bool self_spine = nb > 1 && nb <= SELF_SPINE_MAX;
    const dim3 grid(nb), block(SCAN_WG);
    if (GATHER && nb > 1) {   // the reduce pass staged the gathered values in `out`: scan them in place
        if (self_spine) {
            if (exclusive) hipLaunchKernelGGL((scan_apply_kernel<false, true, true>), grid, block, 0, ctx->stream, out, nullptr, n, sums, out, gate);
            else hipLaunchKernelGGL((scan_apply_kernel<false, false, true>), grid, block, 0, ctx->stream, out, nullptr, n, sums, out, gate);
        } else {
            if (exclusive) hipLaunchKernelGGL((scan_apply_kernel<false, true, false>), grid, block, 0, ctx->stream, out, nullptr, n, sums, out, gate);
            else hipLaunchKernelGGL((scan_apply_kernel<false, false, false>), grid, block, 0, ctx->stream, out, nullptr, n, sums, out, gate);
        }
        BH_LAUNCH_CHECK(ctx, "scan_apply_kernel");
        return 0;
    }
    if (self_spine) {
        if (exclusive) hipLaunchKernelGGL((scan_apply_kernel<GATHER, true, true>), grid, block, 0, ctx->stream, in, gather, n, sums, out, gate);
        else hipLaunchKernelGGL((scan_apply_kernel<GATHER, false, true>), grid, block, 0, ctx->stream, in, gather, n, sums, out, gate);
    } else {
        if (exclusive) hipLaunchKernelGGL((scan_apply_kernel<GATHER, true, false>), grid, block, 0, ctx->stream, in, gather, n, sums, out, gate);
        else hipLaunchKernelGGL((scan_apply_kernel<GATHER, false, false>), grid, block, 0, ctx->stream, in, gather, n, sums, out, gate);
    }
    BH_LAUNCH_CHECK(ctx, "scan_apply_kernel");
    return 0;
}

int prefix_sum(bh_ctx* ctx, const uint32_t* in, const uint32_t* gather, uint32_t n, uint32_t* out, bool exclusive, const uint32_t* gate) {
    if (n == 0) return 0;
    return gather ? scan_impl<true>(ctx, in, gather, n, out, exclusive, gate) : scan_impl<false>(ctx, in, gather, n, out, exclusive, gate);
}

}  // namespace bh
