// exchange.hip — the compaction side of the mask-keyed gradient exchange (data parallel over cameras).
//
// Not in the reference (single-GPU).  A view's gradient rows are non-zero only for the splats that reached a pixel of
// that view (visible[] = 1; ~10 % of the rows at the 1 M / 1080p bench workload), so instead of all-reducing the dense
// [N, 11 + 3C] gradient block every step, bh_train_step (BhTrainBatch.exchange_mode = 1)
//   1. sums the visible flags over the ranks (needed anyway: vis_weight counts views)      — 4 B per splat
//   2. takes U = { i : summed visible > 0 } — identical on every rank — ranks it with the scan,
//   3. gathers the rows of U into a compact [|U|, 11 + 3C] block, sums THAT, scatters it back.
// Rows outside U are zero on every rank, so the dense buffer ends up exactly as a dense all-reduce would leave it
// (up to the summation order inside the collective).  When U is more than half of the scene the dense block is summed
// instead (the compaction would not pay).  The kernels below are steps 2 and 3.
#include "context.h"

namespace bh {

constexpr int EX_WG = 256;

__global__ __launch_bounds__(EX_WG) void union_mask_kernel(const float* __restrict__ visible_sum, uint32_t n, uint32_t* __restrict__ mask) {
    const uint32_t i = blockIdx.x * EX_WG + threadIdx.x;
    if (i < n) mask[i] = visible_sum[i] > 0.0f ? 1u : 0u;
}

// idx[rank] = splat id, for the splats of U (rank = inclusive scan of the mask - 1)
__global__ __launch_bounds__(EX_WG) void union_index_kernel(const uint32_t* __restrict__ mask, const uint32_t* __restrict__ incl, uint32_t n,
                                                           uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * EX_WG + threadIdx.x;
    if (i < n && mask[i]) idx[incl[i] - 1u] = i;
}

// compact[r, :] <-> (v_transforms[i, 0:10] | v_sh[i, 0:3C] | v_raw_opac[i]),  i = idx[r].  One thread per float of the
// compact block: coalesced on the compact side, row-contiguous on the dense side.
template <bool GATHER>
__global__ __launch_bounds__(EX_WG) void exchange_rows_kernel(const uint32_t* __restrict__ idx, uint32_t count, uint32_t c3,
                                                             float* g_tr, float* g_sh, float* g_op, float* compact) {
    const uint32_t k = 11u + c3;
    const uint64_t e = (uint64_t)blockIdx.x * EX_WG + threadIdx.x;
    if (e >= (uint64_t)count * k) return;
    const uint32_t r = (uint32_t)(e / k), j = (uint32_t)(e - (uint64_t)r * k);
    const uint32_t i = idx[r];
    float* dense = j < 10u ? &g_tr[(size_t)i * 10 + j] : (j < 10u + c3 ? &g_sh[(size_t)i * c3 + (j - 10u)] : &g_op[i]);
    if (GATHER) compact[e] = *dense;
    else *dense = compact[e];
}

int launch_union_mask(bh_ctx* ctx, const float* visible_sum, uint32_t n, uint32_t* mask) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(union_mask_kernel, dim3((n + EX_WG - 1) / EX_WG), dim3(EX_WG), 0, ctx->stream, visible_sum, n, mask);
    BH_LAUNCH_CHECK(ctx, "union_mask_kernel");
    return 0;
}

int launch_union_index(bh_ctx* ctx, const uint32_t* mask, const uint32_t* incl, uint32_t n, uint32_t* idx) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(union_index_kernel, dim3((n + EX_WG - 1) / EX_WG), dim3(EX_WG), 0, ctx->stream, mask, incl, n, idx);
    BH_LAUNCH_CHECK(ctx, "union_index_kernel");
    return 0;
}

int launch_exchange_rows(bh_ctx* ctx, bool gather, const uint32_t* idx, uint32_t count, uint32_t c3, float* g_tr, float* g_sh, float* g_op,
                         float* compact) {
    if (count == 0) return 0;
    const uint64_t total = (uint64_t)count * (11u + c3);
    const dim3 grid((unsigned)((total + EX_WG - 1) / EX_WG)), block(EX_WG);
    if (gather) hipLaunchKernelGGL(exchange_rows_kernel<true>, grid, block, 0, ctx->stream, idx, count, c3, g_tr, g_sh, g_op, compact);
    else hipLaunchKernelGGL(exchange_rows_kernel<false>, grid, block, 0, ctx->stream, idx, count, c3, g_tr, g_sh, g_op, compact);
    BH_LAUNCH_CHECK(ctx, "exchange_rows_kernel");
    return 0;
}

}  // namespace bh
