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

#include "context.h"

namespace bh {

// ---------------------------------------------------------------------------
// K15: get_tile_offsets (get_tile_offset.rs:11-58)
// ---------------------------------------------------------------------------
// Four consecutive intersections per thread (one 16-byte load + the element in front of them).
__global__ __launch_bounds__(256) void tile_offsets_kernel(const uint32_t* __restrict__ tile_ids, uint32_t num_isect,
                                                          uint32_t num_tiles, uint32_t* __restrict__ tile_offsets) {
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 4u;
    if (i0 >= num_isect) return;
    uint32_t t[4];
    if (i0 + 4u <= num_isect) {
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
                if (i == num_isect - 1u) tile_offsets[tid * 2 + 1] = i + 1u;
                if (i == 0u) {
                    tile_offsets[tid * 2] = 0u;
                } else if (tid != prev) {
                    if (prev < num_tiles) tile_offsets[prev * 2 + 1] = i;
                    tile_offsets[tid * 2] = i;
                }
            } else if (i > 0u && prev < num_tiles) {
                // valid -> sentinel transition: close the last valid tile (the reference leaves its end at 0 here,
                // get_tile_offset.rs:28-57 / SURVEY App. B.2)
                tile_offsets[prev * 2 + 1] = i;
            }
            prev = tid;
        }
    }
}

int launch_tile_offsets(bh_ctx* ctx, const uint32_t* tile_ids_sorted, uint32_t num_isect, uint32_t num_tiles,
                        uint32_t* tile_offsets, bool pre_zeroed) {
    // the 8 x 16 work-class counters of the backward's tile order sit right behind the table: one fill clears both
    // (pre_zeroed: the forward's K1 already did, project.hip ForwardPrep)
    if (!pre_zeroed) BH_HIP(ctx, hipMemsetAsync(tile_offsets, 0, ((size_t)num_tiles * 2 + 8 * 16) * 4, ctx->stream));
    if (num_isect == 0) return 0;
    hipLaunchKernelGGL(tile_offsets_kernel, dim3((num_isect + 1023) / 1024), dim3(256), 0, ctx->stream, tile_ids_sorted, num_isect, num_tiles, tile_offsets);
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
};

// ---- longest-first tile order for the backward ------------------------------------------------------------
// A tile is one wave and the chip holds only ~8 waves per SIMD over the whole launch (8160 tiles at 1080p), so
// WHICH tiles share a SIMD decides the makespan: in index order the slowest SIMD carries ~10 % (up to ~40 % per
// wave slot) more blended splats than the mean at the bench workload (scripts/tile_work_hist.py).  The forward
// learns every tile's exact backward work (its list end is shrunk to the last useful splat, rasterize.rs:183-189),
// so it files the tile, per XCD band, into one of LPT_CLASSES work classes (an atomic append), and the backward
// maps block j of an XCD to that band's j-th tile in DESCENDING class order: heavy tiles start first, light ones
// fill the tail.  The band structure (each XCD keeps a contiguous range of tiles for its L2) is unchanged.
constexpr uint32_t LPT_CLASSES = 16;
// layout of the LPT scratch: [8 * LPT_CLASSES] counters (zeroed with tile_offsets), then [8][LPT_CLASSES][per] tile lists
BH_DEV uint32_t lpt_band_tiles(uint32_t num_tiles) { return (num_tiles + 7u) / 8u; }

// One staged splat = 12 floats (48 B, 16-B aligned rows):
//   [0..3] x y c00 c01   [4..7] c11 alpha max(r,0) max(g,0)   [8] max(b,0)
//   [9] sigma_cut  (conservative bound: alpha can only reach the cutoff where sigma <= sigma_cut)
//   [10] colour gate bits (raw r/g/b >= 0)   [11] compact gid
constexpr int SPLAT_STRIDE = 12;
constexpr int BATCH = 64;
constexpr float SIGMA_CUT_MARGIN = 0.01f;  // >> the error of bh_logf/exp_blend (~1e-7)

// exp(x) for the blend loops, x = -sigma <= 0 wherever the result is used (lanes that fail the
// sigma pre-test compute a value nobody reads).  Base-2 form chosen for gfx950 issue rates: ten
// full-rate VALU ops (mul, add, sub, sub, 5 fma, lshl_add) where the Cephes sequence of bh_expf
// takes 14 with three half-rate ones (rndne, cvt, ldexp): k = rint(x*log2e) through the 1.5*2^23
// magic add, 2^f from a degree-5 minimax polynomial on [-0.5, 0.5] (1.6e-7 max rel. error), and the
// exponent spliced in by adding k << 23 to the bit pattern.  The CPU checker used by the tests
// restates the same sequence, so images stay bit-identical to it.
BH_DEV float exp_blend(float x) {
#ifdef BH_HW_EXP  // measurement-only variant (not the shipped numerical spec)
    return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
#endif
    const float t = x * 1.44269504088896341f;
    const float s = t + 12582912.0f;
    const float kf = s - 12582912.0f;
    const float f = t - kf;
    float p = 1.3274633092805743e-3f;
    p = __builtin_fmaf(p, f, 9.671961888670921e-3f);
    p = __builtin_fmaf(p, f, 5.5506784468889236e-2f);
    p = __builtin_fmaf(p, f, 2.4022234976291656e-1f);
    p = __builtin_fmaf(p, f, 6.931470632553101e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return u2f(f2u(p) + (f2u(s) << 23));
}

// 8 XCDs take workgroups round-robin; give each XCD a contiguous band of tiles.
BH_DEV uint32_t tile_of_block(uint32_t b, uint32_t num_tiles) {
    const uint32_t per = (num_tiles + 7u) / 8u;
    return (b & 7u) * per + (b >> 3);
}

// Stage one batch of up to 64 splats of this tile into LDS (lane i stages splat i).
// Everything that is per-splat rather than per-pixel is done here once by the
// staging lane: colour clamp (rasterize.rs:147-149), the gate bits of the backward
// and the conservative sigma bound used for the wave-uniform quadrant skip.
template <bool SMOOTH>
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
        d[0] = make_float4(v[0], v[1], v[2], v[3]);
        d[1] = make_float4(v[4], v[5], __builtin_fmaxf(v[6], 0.0f), __builtin_fmaxf(v[7], 0.0f));
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
#ifdef BH_FWD_WAVES  // measurement-only: cap the forward's occupancy (waves per SIMD)
#define BH_FWD_ATTR __attribute__((amdgpu_waves_per_eu(BH_FWD_WAVES, BH_FWD_WAVES)))
#else
#define BH_FWD_ATTR
#endif
template <bool BWD_INFO, bool SMOOTH>
__global__ __launch_bounds__(64) BH_FWD_ATTR void rasterize_kernel(RasterUniforms u, const uint32_t* __restrict__ isect_gids,
                                                      uint32_t* __restrict__ tile_offsets, const float* __restrict__ projected,
                                                      const uint32_t* __restrict__ global_from_compact,
                                                      float* __restrict__ out_img, uint32_t* __restrict__ out_packed,
                                                      float* __restrict__ visible, uint32_t* __restrict__ lpt) {
    __shared__ __attribute__((aligned(16))) float s_splat[BATCH * SPLAT_STRIDE];
    const uint32_t local_tile = tile_of_block(blockIdx.x, u.num_tiles);
    if (local_tile >= u.num_tiles) return;
    const uint32_t tile = u.tile_begin + local_tile;
    const int lane = threadIdx.x;
    const uint32_t tx0 = (tile % u.tile_bw) * TILE_WIDTH, ty0 = (tile / u.tile_bw) * TILE_WIDTH;
    const uint32_t px0 = tx0 + (lane & 7), py0 = ty0 + (lane >> 3);
    const float pcx[2] = {(float)px0 + 0.5f, (float)(px0 + 8) + 0.5f};
    const float pcy[2] = {(float)py0 + 0.5f, (float)(py0 + 8) + 0.5f};
    // transmittance; a finished pixel keeps its final T with the sign flipped
    float tr[4], pr[4], pg[4], pb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool inside = (px0 + 8 * (q & 1)) < u.img_w && (py0 + 8 * (q >> 1)) < u.img_h;
        tr[q] = inside ? 1.0f : -1.0f;
        pr[q] = pg[q] = pb[q] = 0.0f;
    }
    const uint32_t range_lo = tile_offsets[tile * 2];
    const uint32_t range_hi = tile_offsets[tile * 2 + 1];
    uint32_t last_useful = range_lo;

    for (uint32_t batch_start = range_lo; batch_start < range_hi; batch_start += BATCH) {
        const bool live = tr[0] > 0.0f || tr[1] > 0.0f || tr[2] > 0.0f || tr[3] > 0.0f;
        if (__ballot(live) == 0ull) break;
        const uint32_t cnt = min((uint32_t)BATCH, range_hi - batch_start);
        __syncthreads();  // previous batch fully consumed (single wave: cheap)
        const uint32_t cg = stage_batch<SMOOTH>(isect_gids, projected, batch_start, cnt, lane, s_splat);
        __syncthreads();
        unsigned long long contrib_mask = 0ull;
        for (uint32_t t = 0; t < cnt; ++t) {
            const float4 s0 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE]);      // x y c00 c01
            const float4 s1 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE + 4]);  // c11 a r g
            const float2 s2 = *reinterpret_cast<const float2*>(&s_splat[t * SPLAT_STRIDE + 8]);  // b sigma_cut
            const uint32_t cut_bits = f2u(s2.y);
            float a_xx[2], b_x[2], c_y[2], dy[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float dx = pcx[k] - s0.x;
                a_xx[k] = (s0.z * dx) * dx;
                b_x[k] = s0.w * dx;
                dy[k] = pcy[k] - s0.y;
                c_y[k] = s1.x * dy[k];
            }
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = q & 1, m = q >> 1;
                const float qv = __builtin_fmaf(c_y[m], dy[m], a_xx[k]);
                const float sigma = __builtin_fmaf(b_x[k], dy[m], 0.5f * qv);
                // live pixel and 0 <= sigma <= sigma_cut (unsigned compare of the bit patterns)
                const bool pre = tr[q] > 0.0f && f2u(sigma) <= cut_bits;
                if (__ballot(pre) != 0ull) {
                    const float alpha = __builtin_fminf(0.999f, s1.y * exp_blend(-sigma));
                    const float w_cut = SMOOTH ? alpha_cutoff_weight(alpha) : (alpha >= ALPHA_CUTOFF_MID ? 1.0f : 0.0f);
                    const bool ok = pre && w_cut > 0.0f;  // pre already implies sigma >= 0
                    const float alpha_eff = SMOOTH ? alpha * w_cut : alpha;
                    const float next_t = tr[q] * (1.0f - alpha_eff);
                    const bool sat = next_t <= 1.0e-4f;
                    const bool contrib = ok && !sat;
                    const float vis = contrib ? alpha_eff * tr[q] : 0.0f;
                    pr[q] += s1.z * vis;
                    pg[q] += s1.w * vis;
                    pb[q] += s2.x * vis;
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
                const bool still = tr[0] > 0.0f || tr[1] > 0.0f || tr[2] > 0.0f || tr[3] > 0.0f;
                if (__ballot(still) == 0ull) break;
            }
        }
        if (BWD_INFO) {
            // rasterize.rs:143-145: mark splats that touched at least one pixel
            if ((contrib_mask >> lane) & 1ull) visible[global_from_compact[cg]] = 1.0f;
        }
    }

#pragma unroll
    for (int q = 0; q < 4; ++q) {
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
    // rasterize.rs:183-189: shrink the tile's end to one past the last useful splat
    if (BWD_INFO && lane == 0) {
        tile_offsets[tile * 2 + 1] = last_useful;
        if (lpt) {  // file the tile under its backward work class (longest-first order, see LPT above)
            const uint32_t work = last_useful - range_lo;
            const uint32_t cls = min(LPT_CLASSES - 1u, (uint32_t)((float)work * u.rcp_class_width));
            const uint32_t list = (blockIdx.x & 7u) * LPT_CLASSES + cls;
            const uint32_t pos = atomicAdd(&lpt[list], 1u);
            lpt[8u * LPT_CLASSES + list * lpt_band_tiles(u.num_tiles) + pos] = local_tile;
        }
    }
}

int launch_rasterize(bh_ctx* ctx, const ViewUniforms& vu, const float bg[3], bool bwd_info, bool smooth,
                     const uint32_t* isect_gids, uint32_t* tile_offsets, const float* projected,
                     const uint32_t* global_from_compact, float* out_img, uint32_t* out_packed, float* visible,
                     uint32_t* lpt, float class_width) {
    RasterUniforms u;
    u.rcp_class_width = 1.0f / (class_width > 1.0f ? class_width : 1.0f);
    u.tile_bw = vu.tile_bw;
    u.num_tiles = vu.tile_bw * (vu.tile_y1 - vu.tile_y0);
    u.tile_begin = vu.tile_bw * vu.tile_y0;
    u.img_w = vu.img_w;
    u.img_h = vu.img_h;
    u.bg_r = bg[0]; u.bg_g = bg[1]; u.bg_b = bg[2];
    const uint32_t nblocks = ((u.num_tiles + 7u) / 8u) * 8u;
    const dim3 grid(nblocks), block(64);
    if (bwd_info && smooth)
        hipLaunchKernelGGL((rasterize_kernel<true, true>), grid, block, 0, ctx->stream, u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible, lpt);
    else if (bwd_info)
        hipLaunchKernelGGL((rasterize_kernel<true, false>), grid, block, 0, ctx->stream, u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible, lpt);
    else
        hipLaunchKernelGGL((rasterize_kernel<false, false>), grid, block, 0, ctx->stream, u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible, lpt);
    BH_LAUNCH_CHECK(ctx, "rasterize_kernel");
    return 0;
}

// ---------------------------------------------------------------------------
// K17: rasterize_backwards (bwd/kernels/rasterize_backwards.rs:101-390)
// ---------------------------------------------------------------------------
// Wave-wide sum of ten per-lane values in 28 VALU ops (a ds_bpermute butterfly
// needs 60 LDS permutes + 60 adds): v_permlane32_swap / v_permlane16_swap fold two
// registers into one per step ("transpose-reduce"), then a DPP rotate-add finishes
// inside each 16-lane row.  Afterwards every lane of row r of k[i] holds component
// comp(i, r): k0 -> g0 g2 g1 g3, k1 -> g4 g6 g5 g7, k2 -> g8 - g9 -.
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

#ifndef BH_BWD_WAVES
#define BH_BWD_WAVES 4
#endif
template <bool SMOOTH>
__global__ __launch_bounds__(64, BH_BWD_WAVES) void rasterize_backward_kernel(RasterUniforms u, const uint32_t* __restrict__ isect_gids,
                                                               const uint32_t* __restrict__ tile_offsets,
                                                               const float* __restrict__ projected,
                                                               const float* __restrict__ out_img,
                                                               const float* __restrict__ v_output,
                                                               float* __restrict__ v_combined, const uint32_t* __restrict__ lpt) {
    __shared__ __attribute__((aligned(16))) float s_splat[BATCH * SPLAT_STRIDE];
    uint32_t local_tile;
    if (lpt) {
        // block j of XCD x takes the j-th tile of band x in descending work-class order (wave-uniform scalar code)
        const uint32_t xcd = blockIdx.x & 7u;
        uint32_t j = blockIdx.x >> 3;
        const uint32_t* cnt = lpt + xcd * LPT_CLASSES;
        uint32_t cls = LPT_CLASSES;
        bool found = false;
#pragma unroll
        for (uint32_t c = LPT_CLASSES; c-- > 0u;) {
            const uint32_t k = cnt[c];
            if (!found) {
                if (j < k) { cls = c; found = true; }
                else j -= k;
            }
        }
        if (!found) return;  // more blocks than tiles in this band
        local_tile = lpt[8u * LPT_CLASSES + (xcd * LPT_CLASSES + cls) * lpt_band_tiles(u.num_tiles) + j];
    } else {
        local_tile = tile_of_block(blockIdx.x, u.num_tiles);
        if (local_tile >= u.num_tiles) return;
    }
    const uint32_t tile = u.tile_begin + local_tile;
    const uint32_t range_lo = tile_offsets[tile * 2];
    const uint32_t range_hi = tile_offsets[tile * 2 + 1];
    if (range_hi <= range_lo) return;
    const int lane = threadIdx.x;
    const uint32_t tx0 = (tile % u.tile_bw) * TILE_WIDTH, ty0 = (tile / u.tile_bw) * TILE_WIDTH;
    const uint32_t px0 = tx0 + (lane & 7), py0 = ty0 + (lane >> 3);
    const float pcx[2] = {(float)px0 + 0.5f, (float)(px0 + 8) + 0.5f};
    const float pcy[2] = {(float)py0 + 0.5f, (float)(py0 + 8) + 0.5f};
    const float img_w_f = (float)u.img_w, img_h_f = (float)u.img_h;
    // which reduced component this lane adds to v_combined (see the reduction above)
    const int ri = lane & 15, row = lane >> 4;
    const int comp = ri * 4 + (((row & 1) << 1) | (row >> 1));
    const bool atom_lane = ri < 3 && comp < 10;
    // pixel replay state (rasterize_backwards.rs:186-228): remaining rgb and T (0 = finished)
    float sx[4], sy[4], sz[4], sw[4];
    float vox[4], voy[4], voz[4], v_o_w[4], inv_fa[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t px = px0 + 8 * (q & 1), py = py0 + 8 * (q >> 1);
        if (px < u.img_w && py < u.img_h) {
            const size_t pix = ((size_t)px + (size_t)py * u.img_w) * 4;
            const float4 o = *reinterpret_cast<const float4*>(&out_img[pix]);
            const float4 vo = *reinterpret_cast<const float4*>(&v_output[pix]);
            const float t_final = 1.0f - o.w;
            sx[q] = o.x - t_final * u.bg_r;
            sy[q] = o.y - t_final * u.bg_g;
            sz[q] = o.z - t_final * u.bg_b;
            sw[q] = 1.0f;
            vox[q] = vo.x; voy[q] = vo.y; voz[q] = vo.z;
            v_o_w[q] = (vo.w - (u.bg_r * vo.x + u.bg_g * vo.y + u.bg_b * vo.z)) * t_final;
            inv_fa[q] = 1.0f / __builtin_fmaxf(o.w, 1.0e-5f);
        } else {
            sx[q] = sy[q] = sz[q] = sw[q] = 0.0f;
            vox[q] = voy[q] = voz[q] = v_o_w[q] = 0.0f;
            inv_fa[q] = 1.0f;
        }
    }

    for (uint32_t batch_start = range_lo; batch_start < range_hi; batch_start += BATCH) {
        const uint32_t cnt = min((uint32_t)BATCH, range_hi - batch_start);
        __syncthreads();
        stage_batch<SMOOTH>(isect_gids, projected, batch_start, cnt, lane, s_splat);
        __syncthreads();
        for (uint32_t t = 0; t < cnt; ++t) {
            const float4 s0 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE]);      // x y c00 c01
            const float4 s1 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE + 4]);  // c11 a r g (clamped)
            const float4 s2 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE + 8]);  // b cut gate gid
            const float c00 = s0.z, c01 = s0.w, c11 = s1.x, color_a = s1.y;
            const float cr = s1.z, cgc = s1.w, cb = s2.x;
            const uint32_t cut_bits = f2u(s2.y);
            float dxp[2], dyp[2], a_xx[2], b_x[2], c_y[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                dxp[k] = pcx[k] - s0.x;  // pixel - mean (forward convention)
                a_xx[k] = (c00 * dxp[k]) * dxp[k];
                b_x[k] = c01 * dxp[k];
                dyp[k] = pcy[k] - s0.y;
                c_y[k] = c11 * dyp[k];
            }
            float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f, g5 = 0.f, g6 = 0.f, g7 = 0.f, g8 = 0.f, g9 = 0.f;
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = q & 1, m = q >> 1;
                // --- replay: identical arithmetic to the forward kernel -------------
                const float qv = __builtin_fmaf(c_y[m], dyp[m], a_xx[k]);
                const float sigma = __builtin_fmaf(b_x[k], dyp[m], 0.5f * qv);
                const bool pre = sw[q] > 0.0f && f2u(sigma) <= cut_bits;
                if (__ballot(pre) != 0ull) {
                    const float gaussian = exp_blend(-sigma);
                    const float alpha_raw = color_a * gaussian;
                    const float alpha = __builtin_fminf(0.999f, alpha_raw);
                    const float w_cut = SMOOTH ? alpha_cutoff_weight(alpha) : (alpha >= ALPHA_CUTOFF_MID ? 1.0f : 0.0f);
                    const bool ok = pre && w_cut > 0.0f;  // pre already implies sigma >= 0
                    const float alpha_eff = SMOOTH ? alpha * w_cut : alpha;
                    const float one_m = 1.0f - alpha_eff;
                    const float next_t = sw[q] * one_m;
                    const bool sat = next_t <= 1.0e-4f;
                    const bool contrib = ok && !sat;
                    const float T = sw[q];
                    const float vis = contrib ? alpha_eff * T : 0.0f;
                    // --- gradients (rasterize_backwards.rs:286-381); tolerance-checked, so
                    //     explicit fma / v_rcp / v_sqrt are used freely here ------------------
                    g5 = __builtin_fmaf(vis, vox[q], g5);
                    g6 = __builtin_fmaf(vis, voy[q], g6);
                    g7 = __builtin_fmaf(vis, voz[q], g7);
                    const float ra = __builtin_amdgcn_rcpf(one_m);
                    float dot_rgb = __builtin_fmaf(T, cr, -sx[q]) * vox[q];
                    dot_rgb = __builtin_fmaf(__builtin_fmaf(T, cgc, -sy[q]), voy[q], dot_rgb);
                    dot_rgb = __builtin_fmaf(__builtin_fmaf(T, cb, -sz[q]), voz[q], dot_rgb);
                    const float v_alpha_eff = (dot_rgb + v_o_w[q]) * ra;
                    const float dw = SMOOTH ? alpha_cutoff_weight_deriv(alpha) : 0.0f;
                    const float v_alpha = SMOOTH ? v_alpha_eff * (w_cut + alpha * dw) : v_alpha_eff;
                    // geometry / opacity / refine grads only below the alpha clamp (…:332)
                    const bool geo = contrib && alpha_raw <= 0.999f;
                    const float v_sigma = geo ? -alpha * v_alpha : 0.0f;
                    const float dx = -dxp[k], dy = -dyp[m];  // mean - pixel (…:300-301)
                    const float vxy_x = v_sigma * __builtin_fmaf(c01, dy, c00 * dx);
                    const float vxy_y = v_sigma * __builtin_fmaf(c11, dy, c01 * dx);
                    const float hs = 0.5f * v_sigma;
                    g2 = __builtin_fmaf(hs * dx, dx, g2);
                    g3 = __builtin_fmaf(v_sigma * dx, dy, g3);
                    g4 = __builtin_fmaf(hs * dy, dy, g4);
                    g0 += vxy_x;
                    g1 += vxy_y;
                    g8 += geo ? v_alpha * gaussian : 0.0f;
                    const float wx = vxy_x * img_w_f, wy = vxy_y * img_h_f;
                    g9 = __builtin_fmaf(__builtin_amdgcn_sqrtf(__builtin_fmaf(wx, wx, wy * wy)), inv_fa[q], g9);
                    // --- state update ---------------------------------------------------------
                    sx[q] = __builtin_fmaf(-vis, cr, sx[q]);
                    sy[q] = __builtin_fmaf(-vis, cgc, sy[q]);
                    sz[q] = __builtin_fmaf(-vis, cb, sz[q]);
                    sw[q] = ok ? (sat ? 0.0f : next_t) : T;
                    any = any || contrib;
                }
            }
            if (__ballot(any) != 0ull) {
                // colour gradients only where the raw colour was >= 0 (…:321-323); wave-uniform
                const uint32_t gate = f2u(s2.z);
                g5 = (gate & 1u) ? g5 : 0.0f;
                g6 = (gate & 2u) ? g6 : 0.0f;
                g7 = (gate & 4u) ? g7 : 0.0f;
                const float h0 = swap32_add(g0, g1), h1 = swap32_add(g2, g3), h2 = swap32_add(g4, g5);
                const float h3 = swap32_add(g6, g7), h4 = swap32_add(g8, g9);
                const float k0 = row_allreduce(swap16_add(h0, h1));
                const float k1 = row_allreduce(swap16_add(h2, h3));
                const float k2 = row_allreduce(swap16_add(h4, 0.0f));
                const float mine = ri == 0 ? k0 : (ri == 1 ? k1 : k2);
                const uint32_t cg = f2u(s2.w);
#ifdef BH_NO_ATOMIC  // measurement-only variant
                if (atom_lane && mine == 123.456f) v_combined[(size_t)cg * 10 + comp] = mine;
#else
                if (atom_lane) unsafeAtomicAdd(&v_combined[(size_t)cg * 10 + comp], mine);
#endif
            }
        }
    }
}

int launch_rasterize_backward(bh_ctx* ctx, const ViewUniforms& vu, const float bg[3], bool smooth,
                              const uint32_t* isect_gids, const uint32_t* tile_offsets, const float* projected,
                              const float* out_img, const float* v_output, float* v_combined, const uint32_t* lpt) {
    RasterUniforms u;
    u.rcp_class_width = 1.0f;
    u.tile_bw = vu.tile_bw;
    u.num_tiles = vu.tile_bw * (vu.tile_y1 - vu.tile_y0);
    u.tile_begin = vu.tile_bw * vu.tile_y0;
    u.img_w = vu.img_w;
    u.img_h = vu.img_h;
    u.bg_r = bg[0]; u.bg_g = bg[1]; u.bg_b = bg[2];
    const uint32_t nblocks = ((u.num_tiles + 7u) / 8u) * 8u;
    const dim3 grid(nblocks), block(64);
    // profiling level 2 (bench.py's timed region): the launch carries its own start / stop events (context.h)
    hipEvent_t ea = ctx->prof.ext_a, eb = ctx->prof.ext_b;
    ctx->prof.ext_a = ctx->prof.ext_b = nullptr;
    if (ea && smooth)
        hipExtLaunchKernelGGL(rasterize_backward_kernel<true>, grid, block, 0, ctx->stream, ea, eb, 0, u, isect_gids, tile_offsets, projected, out_img, v_output, v_combined, lpt);
    else if (ea)
        hipExtLaunchKernelGGL(rasterize_backward_kernel<false>, grid, block, 0, ctx->stream, ea, eb, 0, u, isect_gids, tile_offsets, projected, out_img, v_output, v_combined, lpt);
    else if (smooth)
        hipLaunchKernelGGL(rasterize_backward_kernel<true>, grid, block, 0, ctx->stream, u, isect_gids, tile_offsets, projected, out_img, v_output, v_combined, lpt);
    else
        hipLaunchKernelGGL(rasterize_backward_kernel<false>, grid, block, 0, ctx->stream, u, isect_gids, tile_offsets, projected, out_img, v_output, v_combined, lpt);
    BH_LAUNCH_CHECK(ctx, "rasterize_backward_kernel");
    return 0;
}

}  // namespace bh
