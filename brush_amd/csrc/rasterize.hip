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
//   * exp() is the same fixed polynomial as everywhere else (device_math.h) without
//     the range guards: bit-identical where alpha can reach 1/255.
#include "context.h"

namespace bh {

// ---------------------------------------------------------------------------
// K15: get_tile_offsets (get_tile_offset.rs:11-58)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_offsets_kernel(const uint32_t* __restrict__ tile_ids, uint32_t num_isect,
                                                          uint32_t num_tiles, uint32_t* __restrict__ tile_offsets) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= num_isect) return;
    const uint32_t tid = tile_ids[i];
    if (tid >= num_tiles) return;  // sentinel rows
    if (i == num_isect - 1) tile_offsets[tid * 2 + 1] = i + 1;
    if (i == 0) {
        tile_offsets[tid * 2] = 0;
    } else {
        const uint32_t prev = tile_ids[i - 1];
        if (tid != prev) {
            if (prev < num_tiles) tile_offsets[prev * 2 + 1] = i;
            tile_offsets[tid * 2] = i;
        }
    }
}

int launch_tile_offsets(bh_ctx* ctx, const uint32_t* tile_ids_sorted, uint32_t num_isect, uint32_t num_tiles,
                        uint32_t* tile_offsets) {
    BH_HIP(ctx, hipMemsetAsync(tile_offsets, 0, (size_t)num_tiles * 2 * 4, ctx->stream));
    if (num_isect == 0) return 0;
    hipLaunchKernelGGL(tile_offsets_kernel, dim3((num_isect + 255) / 256), dim3(256), 0, ctx->stream, tile_ids_sorted, num_isect, num_tiles, tile_offsets);
    BH_LAUNCH_CHECK(ctx, "tile_offsets_kernel");
    return 0;
}

// ---------------------------------------------------------------------------
// shared pieces of the two blend kernels
// ---------------------------------------------------------------------------
struct RasterUniforms {
    uint32_t tile_bw, num_tiles, img_w, img_h;
    float bg_r, bg_g, bg_b;
};

constexpr int SPLAT_STRIDE = 12;  // floats per staged splat (9 used): 16-B aligned rows
constexpr int BATCH = 64;

// exp(x) for the blend loop: the bh_expf sequence without its range guards
// (x <= 0 wherever the result is used; underflow goes to 0 through ldexp).
BH_DEV float exp_blend(float x) {
    const float k = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(k, -0.693359375f, x);
    r = __builtin_fmaf(k, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    return __builtin_ldexpf(y, (int)k);
}

// 8 XCDs take workgroups round-robin; give each XCD a contiguous band of tiles.
BH_DEV uint32_t tile_of_block(uint32_t b, uint32_t num_tiles) {
    const uint32_t per = (num_tiles + 7u) / 8u;
    return (b & 7u) * per + (b >> 3);
}

// Stage one batch of up to 64 splats of this tile into LDS (lane i stages splat i).
BH_DEV uint32_t stage_batch(const uint32_t* __restrict__ isect_gids, const float* __restrict__ projected,
                            uint32_t batch_start, uint32_t cnt, int lane, float* s_splat) {
    uint32_t cg = 0;
    if ((uint32_t)lane < cnt) {
        cg = isect_gids[batch_start + lane];
        const float* p = projected + (size_t)cg * 9;
        float* d = s_splat + lane * SPLAT_STRIDE;
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] = p[k];
    }
    return cg;
}

// ---------------------------------------------------------------------------
// K16: rasterize (kernels/rasterize.rs:27-190)
// ---------------------------------------------------------------------------
template <bool BWD_INFO, bool SMOOTH>
__global__ __launch_bounds__(64) void rasterize_kernel(RasterUniforms u, const uint32_t* __restrict__ isect_gids,
                                                      uint32_t* __restrict__ tile_offsets, const float* __restrict__ projected,
                                                      const uint32_t* __restrict__ global_from_compact,
                                                      float* __restrict__ out_img, uint32_t* __restrict__ out_packed,
                                                      float* __restrict__ visible) {
    __shared__ __attribute__((aligned(16))) float s_splat[BATCH * SPLAT_STRIDE];
    const uint32_t tile = tile_of_block(blockIdx.x, u.num_tiles);
    if (tile >= u.num_tiles) return;
    const int lane = threadIdx.x;
    const uint32_t tx0 = (tile % u.tile_bw) * TILE_WIDTH, ty0 = (tile / u.tile_bw) * TILE_WIDTH;
    const uint32_t px = tx0 + (lane & 15);
    const uint32_t py0 = ty0 + (lane >> 4);
    const float pcx = (float)px + 0.5f;
    float pcy[4];
    bool done[4];
    float t_acc[4], pr[4], pg[4], pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t py = py0 + 4 * j;
        pcy[j] = (float)py + 0.5f;
        done[j] = !(px < u.img_w && py < u.img_h);
        t_acc[j] = 1.0f;
        pr[j] = pg[j] = pb[j] = 0.0f;
    }
    const uint32_t range_lo = tile_offsets[tile * 2];
    const uint32_t range_hi = tile_offsets[tile * 2 + 1];
    uint32_t last_useful = range_lo;

    for (uint32_t batch_start = range_lo; batch_start < range_hi; batch_start += BATCH) {
        const bool all_done = done[0] && done[1] && done[2] && done[3];
        if (__ballot(!all_done) == 0ull) break;
        const uint32_t cnt = min((uint32_t)BATCH, range_hi - batch_start);
        __syncthreads();  // previous batch fully consumed (single wave: cheap)
        const uint32_t cg = stage_batch(isect_gids, projected, batch_start, cnt, lane, s_splat);
        __syncthreads();
        unsigned long long contrib_mask = 0ull;
        for (uint32_t t = 0; t < cnt; ++t) {
            const float4 s0 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE]);      // x y c00 c01
            const float4 s1 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE + 4]);  // c11 a r g
            const float sb = s_splat[t * SPLAT_STRIDE + 8];
            const float dx = pcx - s0.x;
            const float a_xx = (s0.z * dx) * dx;
            const float b_x = s0.w * dx;
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dy = pcy[j] - s0.y;
                const float q = __builtin_fmaf(s1.x * dy, dy, a_xx);
                const float sigma = __builtin_fmaf(b_x, dy, 0.5f * q);
                const float alpha = __builtin_fminf(0.999f, s1.y * exp_blend(-sigma));
                const float w_cut = SMOOTH ? alpha_cutoff_weight(alpha) : (alpha >= ALPHA_CUTOFF_MID ? 1.0f : 0.0f);
                if (!done[j] && sigma >= 0.0f && w_cut > 0.0f) {
                    const float alpha_eff = alpha * w_cut;
                    const float next_t = t_acc[j] * (1.0f - alpha_eff);
                    if (next_t <= 1.0e-4f) {
                        done[j] = true;
                    } else {
                        const float vis = alpha_eff * t_acc[j];
                        pr[j] += __builtin_fmaxf(s1.z, 0.0f) * vis;
                        pg[j] += __builtin_fmaxf(s1.w, 0.0f) * vis;
                        pb[j] += __builtin_fmaxf(sb, 0.0f) * vis;
                        t_acc[j] = next_t;
                        any = true;
                    }
                }
            }
            if (BWD_INFO) {
                if (__ballot(any) != 0ull) {
                    contrib_mask |= 1ull << t;
                    last_useful = batch_start + t + 1;
                }
            }
        }
        if (BWD_INFO) {
            // rasterize.rs:143-145: mark splats that touched at least one pixel
            if ((contrib_mask >> lane) & 1ull) visible[global_from_compact[cg]] = 1.0f;
        }
    }

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t py = py0 + 4 * j;
        if (px < u.img_w && py < u.img_h) {
            const float fr = pr[j] + t_acc[j] * u.bg_r;
            const float fg = pg[j] + t_acc[j] * u.bg_g;
            const float fb = pb[j] + t_acc[j] * u.bg_b;
            const float fa = 1.0f - t_acc[j];
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
    if (BWD_INFO && lane == 0) tile_offsets[tile * 2 + 1] = last_useful;
}

int launch_rasterize(bh_ctx* ctx, const ViewUniforms& vu, const float bg[3], bool bwd_info, bool smooth,
                     const uint32_t* isect_gids, uint32_t* tile_offsets, const float* projected,
                     const uint32_t* global_from_compact, float* out_img, uint32_t* out_packed, float* visible) {
    RasterUniforms u;
    u.tile_bw = vu.tile_bw;
    u.num_tiles = vu.tile_bw * vu.tile_bh;
    u.img_w = vu.img_w;
    u.img_h = vu.img_h;
    u.bg_r = bg[0]; u.bg_g = bg[1]; u.bg_b = bg[2];
    const uint32_t nblocks = ((u.num_tiles + 7u) / 8u) * 8u;
    const dim3 grid(nblocks), block(64);
    if (bwd_info && smooth)
        hipLaunchKernelGGL((rasterize_kernel<true, true>), grid, block, 0, ctx->stream, u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible);
    else if (bwd_info)
        hipLaunchKernelGGL((rasterize_kernel<true, false>), grid, block, 0, ctx->stream, u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible);
    else
        hipLaunchKernelGGL((rasterize_kernel<false, false>), grid, block, 0, ctx->stream, u, isect_gids, tile_offsets, projected, global_from_compact, out_img, out_packed, visible);
    BH_LAUNCH_CHECK(ctx, "rasterize_kernel");
    return 0;
}

// ---------------------------------------------------------------------------
// K17: rasterize_backwards (bwd/kernels/rasterize_backwards.rs:101-390)
// ---------------------------------------------------------------------------
BH_DEV float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <bool SMOOTH>
__global__ __launch_bounds__(64) void rasterize_backward_kernel(RasterUniforms u, const uint32_t* __restrict__ isect_gids,
                                                               const uint32_t* __restrict__ tile_offsets,
                                                               const float* __restrict__ projected,
                                                               const float* __restrict__ out_img,
                                                               const float* __restrict__ v_output,
                                                               float* __restrict__ v_combined) {
    __shared__ __attribute__((aligned(16))) float s_splat[BATCH * SPLAT_STRIDE];
    const uint32_t tile = tile_of_block(blockIdx.x, u.num_tiles);
    if (tile >= u.num_tiles) return;
    const uint32_t range_lo = tile_offsets[tile * 2];
    const uint32_t range_hi = tile_offsets[tile * 2 + 1];
    if (range_hi <= range_lo) return;
    const int lane = threadIdx.x;
    const uint32_t tx0 = (tile % u.tile_bw) * TILE_WIDTH, ty0 = (tile / u.tile_bw) * TILE_WIDTH;
    const uint32_t px = tx0 + (lane & 15);
    const uint32_t py0 = ty0 + (lane >> 4);
    const float pcx = (float)px + 0.5f;
    const float img_w_f = (float)u.img_w, img_h_f = (float)u.img_h;
    float pcy[4];
    // pixel replay state (rasterize_backwards.rs:186-228): remaining rgb and T
    float sx[4], sy[4], sz[4], sw[4];
    float vox[4], voy[4], voz[4], v_o_w[4], fa_c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t py = py0 + 4 * j;
        pcy[j] = (float)py + 0.5f;
        if (px < u.img_w && py < u.img_h) {
            const size_t pix = ((size_t)px + (size_t)py * u.img_w) * 4;
            const float4 o = *reinterpret_cast<const float4*>(&out_img[pix]);
            const float4 vo = *reinterpret_cast<const float4*>(&v_output[pix]);
            const float t_final = 1.0f - o.w;
            sx[j] = o.x - t_final * u.bg_r;
            sy[j] = o.y - t_final * u.bg_g;
            sz[j] = o.z - t_final * u.bg_b;
            sw[j] = 1.0f;
            vox[j] = vo.x; voy[j] = vo.y; voz[j] = vo.z;
            v_o_w[j] = (vo.w - (u.bg_r * vo.x + u.bg_g * vo.y + u.bg_b * vo.z)) * t_final;
            fa_c[j] = __builtin_fmaxf(o.w, 1.0e-5f);
        } else {
            sx[j] = sy[j] = sz[j] = sw[j] = 0.0f;
            vox[j] = voy[j] = voz[j] = v_o_w[j] = 0.0f;
            fa_c[j] = 1.0f;
        }
    }

    for (uint32_t batch_start = range_lo; batch_start < range_hi; batch_start += BATCH) {
        const uint32_t cnt = min((uint32_t)BATCH, range_hi - batch_start);
        __syncthreads();
        const uint32_t cg_mine = stage_batch(isect_gids, projected, batch_start, cnt, lane, s_splat);
        __syncthreads();
        for (uint32_t t = 0; t < cnt; ++t) {
            const float4 s0 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE]);
            const float4 s1 = *reinterpret_cast<const float4*>(&s_splat[t * SPLAT_STRIDE + 4]);
            const float sb = s_splat[t * SPLAT_STRIDE + 8];
            const float c00 = s0.z, c01 = s0.w, c11 = s1.x, color_a = s1.y;
            const float cr = __builtin_fmaxf(s1.z, 0.0f), cgc = __builtin_fmaxf(s1.w, 0.0f), cb = __builtin_fmaxf(sb, 0.0f);
            const float dxp = pcx - s0.x;   // pixel - mean (forward convention)
            const float a_xx = (c00 * dxp) * dxp;
            const float b_x = c01 * dxp;
            const float dx = s0.x - pcx;    // mean - pixel (backward convention, rasterize_backwards.rs:300-301)
            float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f, g5 = 0.f, g6 = 0.f, g7 = 0.f, g8 = 0.f, g9 = 0.f;
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dyp = pcy[j] - s0.y;
                const float q = __builtin_fmaf(c11 * dyp, dyp, a_xx);
                const float sigma = __builtin_fmaf(b_x, dyp, 0.5f * q);
                const float gaussian = exp_blend(-sigma);
                const float alpha = __builtin_fminf(0.999f, color_a * gaussian);
                const float w_cut = SMOOTH ? alpha_cutoff_weight(alpha) : (alpha >= ALPHA_CUTOFF_MID ? 1.0f : 0.0f);
                if (sw[j] > 1.0e-4f && sigma >= 0.0f && w_cut > 0.0f) {
                    const float alpha_eff = alpha * w_cut;
                    const float next_t = sw[j] * (1.0f - alpha_eff);
                    if (next_t <= 1.0e-4f) {
                        sw[j] = 0.0f;
                    } else {
                        const float dy = s0.y - pcy[j];
                        const float vis = alpha_eff * sw[j];
                        g5 += s1.z >= 0.0f ? vis * vox[j] : 0.0f;
                        g6 += s1.w >= 0.0f ? vis * voy[j] : 0.0f;
                        g7 += sb >= 0.0f ? vis * voz[j] : 0.0f;
                        const float ra = 1.0f / (1.0f - alpha_eff);
                        const float dot_rgb = ((sw[j] * cr - sx[j]) * vox[j] + (sw[j] * cgc - sy[j]) * voy[j] + (sw[j] * cb - sz[j]) * voz[j]) * ra;
                        const float v_alpha_eff = dot_rgb + v_o_w[j] * ra;
                        const float dw = SMOOTH ? alpha_cutoff_weight_deriv(alpha) : 0.0f * alpha;
                        const float v_alpha = v_alpha_eff * (w_cut + alpha * dw);
                        const float v_sigma = -alpha * v_alpha;
                        const float vxy_x = v_sigma * (c00 * dx + c01 * dy);
                        const float vxy_y = v_sigma * (c01 * dx + c11 * dy);
                        if (color_a * gaussian <= 0.999f) {
                            g2 += 0.5f * v_sigma * dx * dx;
                            g3 += v_sigma * dx * dy;
                            g4 += 0.5f * v_sigma * dy * dy;
                            g0 += vxy_x;
                            g1 += vxy_y;
                            g8 += v_alpha * gaussian;
                            const float len = __builtin_sqrtf(vxy_x * img_w_f * vxy_x * img_w_f + vxy_y * img_h_f * vxy_y * img_h_f);
                            g9 += len / fa_c[j];
                        }
                        sx[j] = sx[j] - vis * cr;
                        sy[j] = sy[j] - vis * cgc;
                        sz[j] = sz[j] - vis * cb;
                        sw[j] = next_t;
                        any = true;
                    }
                }
            }
            if (__ballot(any) != 0ull) {
                g0 = wave_sum(g0); g1 = wave_sum(g1); g2 = wave_sum(g2); g3 = wave_sum(g3); g4 = wave_sum(g4);
                g5 = wave_sum(g5); g6 = wave_sum(g6); g7 = wave_sum(g7); g8 = wave_sum(g8); g9 = wave_sum(g9);
                // lanes 0..9 each own one component -> a single 10-lane atomic instruction
                float mine = g0;
                mine = lane == 1 ? g1 : mine; mine = lane == 2 ? g2 : mine; mine = lane == 3 ? g3 : mine;
                mine = lane == 4 ? g4 : mine; mine = lane == 5 ? g5 : mine; mine = lane == 6 ? g6 : mine;
                mine = lane == 7 ? g7 : mine; mine = lane == 8 ? g8 : mine; mine = lane == 9 ? g9 : mine;
                const uint32_t cg = __shfl(cg_mine, (int)t);
                if (lane < 10) unsafeAtomicAdd(&v_combined[(size_t)cg * 10 + lane], mine);
            }
        }
    }
}

int launch_rasterize_backward(bh_ctx* ctx, const ViewUniforms& vu, const float bg[3], bool smooth,
                              const uint32_t* isect_gids, const uint32_t* tile_offsets, const float* projected,
                              const float* out_img, const float* v_output, float* v_combined) {
    RasterUniforms u;
    u.tile_bw = vu.tile_bw;
    u.num_tiles = vu.tile_bw * vu.tile_bh;
    u.img_w = vu.img_w;
    u.img_h = vu.img_h;
    u.bg_r = bg[0]; u.bg_g = bg[1]; u.bg_b = bg[2];
    const uint32_t nblocks = ((u.num_tiles + 7u) / 8u) * 8u;
    const dim3 grid(nblocks), block(64);
    if (smooth)
        hipLaunchKernelGGL(rasterize_backward_kernel<true>, grid, block, 0, ctx->stream, u, isect_gids, tile_offsets, projected, out_img, v_output, v_combined);
    else
        hipLaunchKernelGGL(rasterize_backward_kernel<false>, grid, block, 0, ctx->stream, u, isect_gids, tile_offsets, projected, out_img, v_output, v_combined);
    BH_LAUNCH_CHECK(ctx, "rasterize_backward_kernel");
    return 0;
}

}  // namespace bh
