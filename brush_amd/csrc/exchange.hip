// exchange.hip — the compaction side of the mask-keyed gradient exchange (data parallel over cameras).
//
// Not in the reference (single-GPU).  A view's gradient rows are non-zero only for the splats that reached a pixel of
// that view (visible[] = 1; ~10 % of the rows at the 1 M / 1080p bench workload), so instead of all-reducing the dense
// [N, 11 + 3C] gradient block every step, bh_train_step (BhTrainBatch.exchange_mode = 1)
//   1. sums the visible flags over the ranks (needed anyway: vis_weight counts views)      — 4 B per splat
//   2. takes U = { i : summed visible > 0 } — identical on every rank — and lists it in ascending order (count / place),
//   3. gathers the rows of U into a compact [|U|, 11 + 3C] block, sums THAT, scatters it back.
// Rows outside U are zero on every rank, so the dense buffer ends up exactly as a dense all-reduce would leave it
// (up to the summation order inside the collective).  When U is more than half of the scene the dense block is summed
// instead (the compaction would not pay).  The kernels below are steps 2 and 3.
#include "context.h"

namespace bh {

constexpr int EX_WG = 256;

// ---- rank -> splat id of the union, in three launches (count per 4096-splat block, spine over the block counts, place)
constexpr int EX_EPT = 16;
constexpr int EX_TILE = EX_WG * EX_EPT;

BH_DEV uint32_t ex_wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

__global__ __launch_bounds__(EX_WG) void union_count_kernel(const float* __restrict__ visible_sum, uint32_t n, uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t s_w[EX_WG / 64];
    const uint32_t base = blockIdx.x * EX_TILE;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < EX_EPT; ++j) {
        const uint32_t i = base + j * EX_WG + threadIdx.x;
        if (i < n) acc += visible_sum[i] > 0.0f ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// one block: block_counts <- exclusive prefix, total -> total_out[0]   (nb = N / 4096: a few hundred to a few thousand)
__global__ __launch_bounds__(EX_WG) void union_spine_kernel(uint32_t* __restrict__ block_counts, uint32_t nb, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t s_w[EX_WG / 64];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += EX_WG) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? block_counts[i] : 0u;
        const uint32_t incl = ex_wave_incl_scan(v, lane);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t ofs = s_carry, tot = 0;
#pragma unroll
        for (int w = 0; w < EX_WG / 64; ++w) { ofs += w < wave ? s_w[w] : 0u; tot += s_w[w]; }
        if (i < nb) block_counts[i] = ofs + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) total_out[0] = s_carry;
}

// idx[rank] = splat id: thread t owns the 16 consecutive splats base + 16 t .. (ascending ids -> ascending ranks).
// SELF_SPINE (up to 4 * 256 blocks = 4 M splats): `block_offsets` holds the RAW block counts and every block adds up the
// ones in front of it itself (one coalesced load per thread) — the one-block spine launch is 5 us of pure latency; the
// last block also delivers the total.
constexpr uint32_t EX_SELF_SPINE_MAX = 4 * EX_WG;
template <bool SELF_SPINE>
__global__ __launch_bounds__(EX_WG) void union_place_kernel(const float* __restrict__ visible_sum, uint32_t n, const uint32_t* __restrict__ block_offsets,
                                                           uint32_t* __restrict__ idx, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t s_w[EX_WG / 64];
    __shared__ uint32_t s_a[EX_WG / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t ahead = 0;
    if (SELF_SPINE) {
#pragma unroll
        for (uint32_t k = 0; k < EX_SELF_SPINE_MAX / EX_WG; ++k) {
            const uint32_t i = k * EX_WG + threadIdx.x;
            if (i < blockIdx.x) ahead += block_offsets[i];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ahead += __shfl_down(ahead, off);
        if (lane == 0) s_a[wave] = ahead;
    }
    const uint32_t first = blockIdx.x * EX_TILE + threadIdx.x * EX_EPT;
    uint32_t bits = 0, cnt = 0;
#pragma unroll
    for (int j = 0; j < EX_EPT; ++j) {
        const uint32_t i = first + j;
        const bool f = i < n && visible_sum[i] > 0.0f;
        bits |= f ? (1u << j) : 0u;
        cnt += f ? 1u : 0u;
    }
    const uint32_t incl = ex_wave_incl_scan(cnt, lane);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t ofs = SELF_SPINE ? (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]) : block_offsets[blockIdx.x];
    if (SELF_SPINE && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) total_out[0] = ofs + (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
#pragma unroll
    for (int w = 0; w < EX_WG / 64; ++w) ofs += w < wave ? s_w[w] : 0u;
    uint32_t r = ofs + incl - cnt;
#pragma unroll
    for (int j = 0; j < EX_EPT; ++j)
        if (bits & (1u << j)) idx[r++] = first + j;
}

// compact[r, :] <-> (v_transforms[i, 0:10] | v_sh[i, 0:3C] | v_raw_opac[i] [| refine_weight[i]]),  i = idx[r].  One thread
// per float of the compact block: coalesced on the compact side, row-contiguous on the dense side.  The refine-weight
// column exists only for the tile-partitioned frame (g_ref != NULL): its strips' partial sums add up, while
// data-parallel views keep per-view maxima that are MAX-reduced before refine.
template <bool GATHER>
__global__ __launch_bounds__(EX_WG) void exchange_rows_kernel(const uint32_t* __restrict__ idx, uint32_t count, uint32_t c3,
                                                             float* g_tr, float* g_sh, float* g_op, float* g_ref, float* compact) {
    const uint32_t k = 11u + c3 + (g_ref ? 1u : 0u);
    const uint64_t e = (uint64_t)blockIdx.x * EX_WG + threadIdx.x;
    if (e >= (uint64_t)count * k) return;
    const uint32_t r = (uint32_t)(e / k), j = (uint32_t)(e - (uint64_t)r * k);
    const uint32_t i = idx[r];
    float* dense = j < 10u ? &g_tr[(size_t)i * 10 + j] : (j < 10u + c3 ? &g_sh[(size_t)i * c3 + (j - 10u)] : (j == 10u + c3 ? &g_op[i] : &g_ref[i]));
    if (GATHER) compact[e] = *dense;
    else *dense = compact[e];
}

// idx[0..count) = ids of the splats whose summed visible flag is > 0, ascending; *count_dev = their number
int launch_union_index(bh_ctx* ctx, const float* visible_sum, uint32_t n, uint32_t* block_scratch, uint32_t* count_dev, uint32_t* idx) {
    if (n == 0) return 0;
    const uint32_t nb = (n + EX_TILE - 1) / EX_TILE;
    hipLaunchKernelGGL(union_count_kernel, dim3(nb), dim3(EX_WG), 0, ctx->stream, visible_sum, n, block_scratch);
    BH_LAUNCH_CHECK(ctx, "union_count_kernel");
    if (nb <= EX_SELF_SPINE_MAX) {
        hipLaunchKernelGGL(union_place_kernel<true>, dim3(nb), dim3(EX_WG), 0, ctx->stream, visible_sum, n, block_scratch, idx, count_dev);
    } else {
        hipLaunchKernelGGL(union_spine_kernel, dim3(1), dim3(EX_WG), 0, ctx->stream, block_scratch, nb, count_dev);
        BH_LAUNCH_CHECK(ctx, "union_spine_kernel");
        hipLaunchKernelGGL(union_place_kernel<false>, dim3(nb), dim3(EX_WG), 0, ctx->stream, visible_sum, n, block_scratch, idx, count_dev);
    }
    BH_LAUNCH_CHECK(ctx, "union_place_kernel");
    return 0;
}

int launch_exchange_rows(bh_ctx* ctx, bool gather, const uint32_t* idx, uint32_t count, uint32_t c3, float* g_tr, float* g_sh, float* g_op,
                         float* g_ref, float* compact) {
    if (count == 0) return 0;
    const uint64_t total = (uint64_t)count * (11u + c3 + (g_ref ? 1u : 0u));
    const dim3 grid((unsigned)((total + EX_WG - 1) / EX_WG)), block(EX_WG);
    if (gather) hipLaunchKernelGGL(exchange_rows_kernel<true>, grid, block, 0, ctx->stream, idx, count, c3, g_tr, g_sh, g_op, g_ref, compact);
    else hipLaunchKernelGGL(exchange_rows_kernel<false>, grid, block, 0, ctx->stream, idx, count, c3, g_tr, g_sh, g_op, g_ref, compact);
    BH_LAUNCH_CHECK(ctx, "exchange_rows_kernel");
    return 0;
}

}  // namespace bh
