// loss_fused.hip — the image loss as the train step uses it (train.rs:227-260):
// loss = mean(L1/SSIM map) [+ alpha-match], v_output = dloss/d(out_img).
//
// Reference: brush-loss/src/lib.rs:181-359 (forward), :371-661 (backward), and the
// autodiff node :1041-1104.  The reference's backward kernel recomputes the blurred
// moments on a 28x28 apron per 8x8 block because its forward only keeps the loss map.
// In a train step both passes always run back to back on the same image, so here
//   pass A  (one block per 16x32 tile, one colour plane at a time through the blur buffers): moments -> SSIM -> loss partial
//           sums per 16-row tile row AND the three per-pixel SSIM partials (dmu1, dsigma1, dsigma12) x chain, written
//           once as nine planes [c][j][H][W];
//   pass B  blurs those planes (26x42 halo per tile) and writes v_output [H,W,4] directly.
// Every thread owns two vertically adjacent outputs: their 11-tap column windows share 10 of 12 rows, which takes 45 % off
// the column pass's LDS reads and brings the halo overhead of everything a pass loads from 2.64x to 2.13x (round 2:
// 155 -> 136 us for the pair; the 16x16 one-output version is in the history).
// The apron recompute, the CHW loss map, its grid-wide sum, the v_output memset and the
// HWC<->CHW permutes (lib.rs:1076,1103) all disappear.  Per-output arithmetic keeps the
// tap-pair accumulation order of loss.hip, but this file is compiled with FMA contraction
// ON and uses v_rcp_f32 in the SSIM quotient chain: the loss has no integer-valued
// outputs to keep reproducible, and the results stay within the 2e-6 the parity tests
// allow against the oracle (tests/test_gpu_loss_optim.py::test_fused_loss_matches_oracle_and_standalone).
// Blocks take their tiles by XCD column bands (loss_tile below; round 4): the aprons of neighbouring tiles then come out of
// the XCD's own L2 instead of over the fabric — FETCH_SIZE per launch 95 -> 43 MB (pass A), 333 -> 123 MB (pass B), the pair
// 114 -> 108 us.  (Round 2 had measured a banded order "neutral" in time and left it out: the Infinity Cache served the
// repeated fetches; what they cost is fabric bandwidth the step's other kernels do not need at that moment.)
#include <cmath>

#include "context.h"

#pragma clang fp contract(fast)

namespace bh {

namespace {

constexpr int LB = 16;
constexpr int HALO = 5;
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;
constexpr float INV_255 = 1.0f / 255.0f;

struct Taps { float w[11]; };

Taps gauss_taps() {  // lib.rs:55-68
    Taps g;
    const float sigma = 1.5f;
    float sum = 0.0f;
    for (int i = 0; i < 11; ++i) {
        const float x = (float)i - 5.0f;
        g.w[i] = expf(-x * x / (2.0f * sigma * sigma));
        sum += g.w[i];
    }
    for (int i = 0; i < 11; ++i) g.w[i] /= sum;
    return g;
}

struct FusedArgs {
    uint32_t h, w;
    float l1_w, ssim_w;
    float bg[3];
    int composite, mask, alpha_match;
    float dl_rgb, dl_alpha;
    uint32_t ty_base;  // first 16-row tile row this launch covers (blocks are 32 rows tall, counted from it): strip-wise loss
    uint32_t row_end;  // pass B: first pixel row behind the window (its last block may reach past it)
    // pass B's first block also adds up pass A's per-block loss partials (a separate one-block launch costs 9 us of
    // latency; here the sum rides beside 2000 other blocks)
    const float* sum_src;
    int sum_n;
    float* loss_out;   // device scalar
    float* loss_host;  // pinned host scalar or NULL: the train step's loss lands there without a copy launch
    // pass A, optional: a pinned host word that receives `started_tag` as soon as the kernel starts — i.e. when everything queued
    // in front of it on the stream (the forward's blend) has finished: the host's "near pass is done" signal without an event
    uint32_t* started_host;
    uint32_t started_tag;
    // block -> tile: gx x gy_blocks tiles of 16 x 32 pixels.  band_w != 0: a 1-D grid of 8 * band_w * gy_blocks blocks; block b
    // belongs to XCD b & 7 (the dispatcher deals consecutive workgroups to the eight XCDs in turn) and takes the (b >> 3)-th
    // tile, row-major, of column band b & 7 (band_w tile columns wide) — see loss_tile()
    uint32_t gx, gy_blocks, band_w;
    Taps taps;
};

// Which 16 x 32 tile a block works on.  Row-major (band_w == 0), or by XCD column bands: a pass re-reads the 5-pixel apron of
// everything it loads (2.13x), and with neighbouring tiles dealt to eight different XCDs — each with an L2 of its own — every
// copy of an apron is fetched over the fabric: pass B's FETCH_SIZE was 341 MB per launch for 116 MB of planes, image and GT,
// 5.6 TB/s of L2 misses, which is what the kernel's 61 us were made of.  Inside a band consecutive blocks of one XCD walk a
// strip of band_w tiles row by row, so a tile's left / right / upper neighbours were loaded by the same L2 moments before.
BH_DEV bool loss_tile(const FusedArgs& a, uint32_t& bx, uint32_t& by, uint32_t& linear) {
    if (a.band_w == 0u) {
        bx = blockIdx.x; by = blockIdx.y;
        linear = by * a.gx + bx;
        return true;
    }
    const uint32_t b = blockIdx.x, xcd = b & 7u, j = b >> 3;
    const uint32_t x0 = xcd * a.band_w;
    if (x0 >= a.gx) return false;
    const uint32_t wb = min(a.band_w, a.gx - x0);
    by = j / wb;
    bx = x0 + (j - by * wb);
    linear = b;
    return by < a.gy_blocks;
}

BH_DEV float gt_ch(uint32_t val, uint32_t c) { return (float)((val >> (c * 8u)) & 0xffu) * INV_255; }

}  // namespace

// ---------------------------------------------------------------------------
// Tile shape of both passes: 16 columns x 32 rows per 256-thread block, thread (lx, ly) owns the two vertically adjacent
// outputs (2 ly, 2 ly + 1) of column lx.  The two 11-tap column windows overlap in 10 of 12 rows, so the vertical blur reads
// 12 rows from LDS for two outputs instead of 22 (-45 %), the horizontal blur runs on 42 rows for 32 outputs instead of
// 2 x 26 (-19 %), and so does the halo re-read of whatever the pass loads (2.13x instead of 2.64x).  One colour plane at a
// time goes through the blur buffers, which keeps the block at 40 KB (pass A) / 24 KB (pass B) of LDS.
// ---------------------------------------------------------------------------
namespace {
constexpr int TW = 16, TH = 32;
constexpr int SW = TW + 2 * HALO;   // 26
constexpr int SR = TH + 2 * HALO;   // 42
/* 88, not TW * 5 = 80: a thread's rows are 2 * 88 = 176 floats = 48 banks apart, the four ly of a wave sit 0 / 48 / 32 / 16 banks
   apart and lx * 5 covers every residue mod 16 once — the column pass's 60 reads per plane are conflict-free (80: the rows of ly and
   ly + 2 share their banks).  Only with the block still at 40 KB = four per CU (s_h's last row unpadded, s_red inside it): 68.2 ->
   63.5 us; at 41.0 KB (three blocks per CU) the same pitch cost +5 us. */
#define BH_LOSS_HP 88
#define BH_LOSS_H2P 24   /* 2 rows apart = 48 floats = 16 banks: the two output rows of a 32-lane half never share a bank (65.0 vs 66.2 us) */
constexpr int HP = BH_LOSS_HP;      // row pitch (floats) of pass A's horizontally blurred moments
constexpr int H2P = BH_LOSS_H2P;    // ... of pass B's
}  // namespace

// ---------------------------------------------------------------------------
// pass A
// ---------------------------------------------------------------------------
// partials: nine planes [c][j][H][W] (c = colour, j = dmu1 / dsigma1 / dsigma12, each times the chain factor): planar, so
// that pass A's stores and pass B's loads are coalesced 4-byte runs along a row
__global__ __launch_bounds__(256) void loss_fused_forward_kernel(const float* __restrict__ img /*[H,W,4]*/,
                                                                const uint32_t* __restrict__ gt,
                                                                float* __restrict__ partials /*[3][3][H][W]*/,
                                                                float* __restrict__ block_sums /*per 16-row tile row*/, uint32_t gy, FusedArgs a) {
    __shared__ float2 s_tile[3][SR * SW];          // (pred, gt_eff) per colour plane
    // horizontally blurred moments of ONE plane; the last row carries no pitch padding: with HP = 88 the block is then
    // exactly 40 KB = a quarter of the CU's LDS (conflict-free column reads AND four blocks per CU)
    __shared__ float s_h[(SR - 1) * HP + TW * 5];
    float* s_red = s_h;   // the four wave partials of the scalar loss reuse it after the last plane
    if (a.started_host && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && threadIdx.y == 0)
        *reinterpret_cast<volatile uint32_t*>(a.started_host) = a.started_tag;
    uint32_t bx, by, blin;
    if (!loss_tile(a, bx, by, blin)) return;   // block-uniform
    const int tx0 = (int)bx * TW, ty0 = (int)a.ty_base * LB + (int)by * TH;
    const int lx = threadIdx.x, ly = threadIdx.y;
    const int rank = ly * TW + lx;
    const size_t plane = (size_t)a.h * a.w;
    // Every global load of the block's prologue — the tile's (image, GT) pixels, five per thread, and the GT alpha of the thread's
    // two outputs — is issued before the first one is consumed: as a rolled loop with a load, a wait and three LDS stores per
    // trip the prologue was SEVEN dependent global round trips (the "tile load alone" of the round-3 probes: 23 of the 64 us).
    constexpr int TILE_LOADS = (SR * SW + 255) / 256;
    float lx3[TILE_LOADS][3];
    uint32_t lval[TILE_LOADS];
    bool lin[TILE_LOADS];
    const int pxx = tx0 + lx;
    const int py[2] = {ty0 + 2 * ly, ty0 + 2 * ly + 1};
    const bool inside[2] = {pxx < (int)a.w && py[0] < (int)a.h, pxx < (int)a.w && py[1] < (int)a.h};
    uint32_t own_val[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < TILE_LOADS; ++k) {
        const int i = rank + 256 * k;
        const int r = i / SW, q = i - r * SW;
        const int y = ty0 + r - HALO, x = tx0 + q - HALO;
        lin[k] = i < SR * SW && y >= 0 && x >= 0 && y < (int)a.h && x < (int)a.w;   // zero padding (lib.rs:110-176)
        // (UNCONDITIONAL loads from a clamped, always valid address, masked afterwards: a load under `if` ends in a wait where
        //  the branches join, which serialises the five again)
        const int yc = y < 0 ? 0 : (y >= (int)a.h ? (int)a.h - 1 : y), xc = x < 0 ? 0 : (x >= (int)a.w ? (int)a.w - 1 : x);
        const size_t p = (size_t)yc * a.w + (size_t)xc;
        const float* ip = &img[p * 4];
        lx3[k][0] = ip[0]; lx3[k][1] = ip[1]; lx3[k][2] = ip[2];
        lval[k] = gt[p];
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int yc = py[o] >= (int)a.h ? (int)a.h - 1 : py[o], xc = pxx >= (int)a.w ? (int)a.w - 1 : pxx;
        own_val[o] = gt[(size_t)yc * a.w + (size_t)xc];
    }
#pragma unroll
    for (int k = 0; k < TILE_LOADS; ++k) {
        const int i = rank + 256 * k;
        if (i < SR * SW) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (!lin[k]) lx3[k][0] = lx3[k][1] = lx3[k][2] = 0.0f;
            if (lin[k]) {
                const uint32_t val = lval[k];
                const float ga = gt_ch(val, 3);
                g0 = gt_ch(val, 0); g1 = gt_ch(val, 1); g2 = gt_ch(val, 2);
                if (a.composite) {
                    g0 = g0 + (1.0f - ga) * a.bg[0];
                    g1 = g1 + (1.0f - ga) * a.bg[1];
                    g2 = g2 + (1.0f - ga) * a.bg[2];
                }
            }
            s_tile[0][i] = make_float2(lx3[k][0], g0);
            s_tile[1][i] = make_float2(lx3[k][1], g1);
            s_tile[2][i] = make_float2(lx3[k][2], g2);
        }
    }
    // this thread's two pixels
    float ga[2] = {0.f, 0.f}, chain[2] = {0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        if (inside[o]) {
            ga[o] = gt_ch(own_val[o], 3);
            chain[o] = a.mask ? a.dl_rgb * ga[o] : a.dl_rgb;
        }
    }
    float acc_rgb = 0.0f;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        __syncthreads();   // the tile is loaded (c == 0) / the previous plane's column pass is done with s_h
        // horizontal blur of (x, x^2, y, y^2, xy): 42 rows x 8 column PAIRS — an item loads the 12 pixels its two adjacent
        // outputs share once and squares each of them once
        const int probe_rows = SR * (TW / 2);
        for (int i = rank; i < probe_rows; i += 256) {
            const int r = i / (TW / 2), pair = i - r * (TW / 2);
            const float2* row = &s_tile[c][r * SW + 2 * pair];   // row[k] = pixel at tile column 2*pair - HALO + k
            float x[12], y[12], xx[12], yy[12], xy[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const float2 v = row[k];
                x[k] = v.x; y[k] = v.y;
                xx[k] = v.x * v.x; yy[k] = v.y * v.y; xy[k] = v.x * v.y;
            }
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int cc = HALO + o;
                float sx = 0, sx2 = 0, sy = 0, sy2 = 0, sxy = 0;
#pragma unroll
                for (int d = 1; d < 6; ++d) {
                    const float wd = a.taps.w[5 - d];
                    sx += (x[cc - d] + x[cc + d]) * wd;
                    sx2 += (xx[cc - d] + xx[cc + d]) * wd;
                    sy += (y[cc - d] + y[cc + d]) * wd;
                    sy2 += (yy[cc - d] + yy[cc + d]) * wd;
                    sxy += (xy[cc - d] + xy[cc + d]) * wd;
                }
                const float wc = a.taps.w[5];
                sx += x[cc] * wc;
                sx2 += xx[cc] * wc;
                sy += y[cc] * wc;
                sy2 += yy[cc] * wc;
                sxy += xy[cc] * wc;
                float* op = &s_h[r * HP + (2 * pair + o) * 5];
                op[0] = sx; op[1] = sx2; op[2] = sy; op[3] = sy2; op[4] = sxy;
            }
        }
        __syncthreads();
        // vertical blur: rows 2 ly .. 2 ly + 11 of s_h serve both outputs (output o: rows o .. o + 10, centre o + 5)
        float v[12][5];
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const float* t = &s_h[(2 * ly + r) * HP + lx * 5];
#pragma unroll
            for (int k = 0; k < 5; ++k) v[r][k] = t[k];
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 1; d < 6; ++d) {
                const float wd = a.taps.w[5 - d];
#pragma unroll
                for (int k = 0; k < 5; ++k) m[k] += (v[o + 5 - d][k] + v[o + 5 + d][k]) * wd;
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) m[k] += v[o + 5][k] * a.taps.w[5];
            if (!inside[o]) continue;
            const float mu1 = m[0], mu2 = m[2];
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2;
            const float s1 = __builtin_fmaxf(0.0f, m[1] - mu1_sq), s2 = __builtin_fmaxf(0.0f, m[3] - mu2_sq);
            const float s12 = m[4] - mu1 * mu2;
            const float A = mu1_sq + mu2_sq + SSIM_C1;
            const float B = s1 + s2 + SSIM_C2;
            const float c_top = 2.0f * mu1 * mu2 + SSIM_C1;
            const float d_top = 2.0f * s12 + SSIM_C2;
            // forward value (lib.rs:331-358); one reciprocal each of A and B serves both passes
            const float inv_a = __builtin_amdgcn_rcpf(A), inv_b = __builtin_amdgcn_rcpf(B);
            const float inv_ab = inv_a * inv_b;
            const float cd = c_top * d_top * inv_ab;
            const float ssim = clampf(cd, -1.0f, 1.0f);
            const float2 pg = s_tile[c][(2 * ly + o + HALO) * SW + lx + HALO];
            float lv = a.l1_w * __builtin_fabsf(pg.x - pg.y) + a.ssim_w * ssim;
            if (a.mask) lv = lv * ga[o];
            acc_rgb += lv;
            // SSIM partials for the backward (lib.rs:455-520)
            const bool clamped = cd < -1.0f || cd > 1.0f;
            const float dmu1 = clamped ? 0.0f : 2.0f * mu2 * inv_ab * (d_top - c_top) - 2.0f * mu1 * cd * (inv_a - inv_b);
            const float ds1 = clamped ? 0.0f : -cd * inv_b;
            const float ds12 = clamped ? 0.0f : 2.0f * c_top * inv_ab;
            const size_t p = (size_t)py[o] * a.w + (size_t)pxx;
            float* o3 = partials + (size_t)(c * 3) * plane + p;
            o3[0] = dmu1 * chain[o];
            o3[plane] = ds1 * chain[o];
            o3[2 * plane] = ds12 * chain[o];
        }
    }
    float acc_alpha = 0.0f;
    if (a.alpha_match) {  // lib.rs:203-214
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            if (inside[o]) {
                const float pa = img[((size_t)py[o] * a.w + (size_t)pxx) * 4 + 3];
                float vv = __builtin_fabsf(pa - ga[o]);
                if (a.mask) vv = vv * ga[o];
                acc_alpha += vv;
            }
        }
    }
    // partial of the scalar loss per 16-row tile row (the strip-wise loss sums whole tile rows): waves 0-1 hold the block's
    // upper 16 rows (ly < 8), waves 2-3 the lower ones.   dl_rgb * sum(rgb planes) + dl_alpha * sum(alpha plane)
    float acc = acc_rgb * a.dl_rgb + acc_alpha * a.dl_alpha;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __syncthreads();   // every thread is done with the last plane's s_h
    if ((rank & 63) == 0) s_red[rank >> 6] = acc;
    __syncthreads();
    if (rank < 2) {
        const uint32_t tile_row = a.ty_base + 2u * by + (uint32_t)rank;
        if (tile_row < gy) block_sums[(size_t)tile_row * a.gx + bx] = s_red[2 * rank] + s_red[2 * rank + 1];
    }
}

// ---------------------------------------------------------------------------
// pass B
// ---------------------------------------------------------------------------
#define BH_LOSSB_WAVES 1
__global__ __launch_bounds__(256, BH_LOSSB_WAVES) void loss_fused_backward_kernel(const float* __restrict__ img, const uint32_t* __restrict__ gt,
                                                                 const float* __restrict__ partials /*[3][3][H][W]*/,
                                                                 float* __restrict__ v_output /*[H,W,4]*/, FusedArgs a) {
    __shared__ float s_part[3][SR * SW];       // chain * (dmu1, dsigma1, dsigma12) of ONE colour plane
    __shared__ float s_h2[3][SR * H2P];
    // strip-wise loss: the launch covers pixel rows [row0, row1) (whole 16-row tile rows); blocks are 32 rows tall
    const int lx = threadIdx.x, ly = threadIdx.y;
    const int rank = ly * TW + lx;
    const size_t plane = (size_t)a.h * a.w;
    if (blockIdx.x == 0 && blockIdx.y == 0 && a.loss_out) {   // block-uniform (block 0 owns a tile in either mapping)
        __shared__ float s_w[4];
        float acc = 0.0f;
        for (int i = rank; i < a.sum_n; i += 256) acc += a.sum_src[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if ((rank & 63) == 0) s_w[rank >> 6] = acc;
        __syncthreads();
        if (rank == 0) {
            const float total = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
            a.loss_out[0] = total;
            if (a.loss_host) a.loss_host[0] = total;
        }
    }
    uint32_t bx, by, blin;
    if (!loss_tile(a, bx, by, blin)) return;   // block-uniform
    const int tx0 = (int)bx * TW, ty0 = (int)a.ty_base * LB + (int)by * TH;
    const int pxx = tx0 + lx;
    const int py[2] = {ty0 + 2 * ly, ty0 + 2 * ly + 1};
    // rows at or behind row_end belong to the next strip (the window ends on a 16-row boundary, blocks are 32 tall)
    const bool inside[2] = {pxx < (int)a.w && py[0] < (int)a.h && py[0] < (int)a.row_end, pxx < (int)a.w && py[1] < (int)a.h && py[1] < (int)a.row_end};
    float4 pv[2];
    uint32_t val[2] = {0u, 0u};
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        pv[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inside[o]) {
            const size_t p = (size_t)py[o] * a.w + (size_t)pxx;
            pv[o] = *reinterpret_cast<const float4*>(&img[p * 4]);
            val[o] = gt[p];
        }
    }
    float out[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // block-uniform: rows of the partial planes can be fetched as aligned float4s
    const bool wide_rows = (a.w & 3u) == 0u && (plane & 3u) == 0u && (reinterpret_cast<uintptr_t>(partials) & 15u) == 0;
    // A tile row is 26 floats from column tx0 - 5: fetched as the 8 aligned float4s from tx0 - 8 (128 contiguous bytes; W % 4 == 0,
    // so a float4 lies entirely inside or outside the image) — 3 x 336 16-byte loads per colour instead of 3 x 1092 4-byte ones,
    // four per thread, and they are issued one colour AHEAD into registers: a block is a chain of nine barrier-separated phases
    // and only six blocks fit a CU, so a global round trip per colour standing in front of its blur passes was most of the
    // block's life (probe builds: the plane loads cost 12-16 us, the two blur passes 11, the rest of the 65-us kernel was waiting).
    constexpr int ROW_ITEMS = 3 * SR * 8;
    auto fetch_rows = [&](const int c, float4 (&dst)[4]) {
        const float* pc = partials + (size_t)(c * 3) * plane;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = rank + 256 * t;
            dst[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (i < ROW_ITEMS) {
                const int j = i / (SR * 8), rem = i - j * (SR * 8);
                const int r = rem >> 3, k = rem & 7;
                const int y = ty0 + r - HALO, x = tx0 - 8 + 4 * k;
                const bool in = y >= 0 && x >= 0 && y < (int)a.h && x < (int)a.w;
                if (in) dst[t] = *reinterpret_cast<const float4*>(&pc[(size_t)j * plane + (size_t)y * a.w + (size_t)x]);
            }
        }
    };
    auto stash_rows = [&](const float4 (&src)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = rank + 256 * t;
            if (i < ROW_ITEMS) {
                const int j = i / (SR * 8), rem = i - j * (SR * 8);
                const int r = rem >> 3, k = rem & 7;
                const float vv[4] = {src[t].x, src[t].y, src[t].z, src[t].w};
                float* dst = &s_part[j][r * SW];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = 4 * k + u - 3;   // tile column of this component
                    if (q >= 0 && q < SW) dst[q] = vv[u];
                }
            }
        }
    };
    float4 pre[4];
    if (wide_rows) fetch_rows(0, pre);
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        __syncthreads();   // the previous plane's column pass is done with the buffers
        const float* pc = partials + (size_t)(c * 3) * plane;
        if (wide_rows) {
            stash_rows(pre);
            if (c < 2) fetch_rows(c + 1, pre);   // in flight while this plane goes through the two blur passes
        } else {
            for (int i = rank; i < SR * SW; i += 256) {
                const int r = i / SW, q = i - r * SW;
                const int y = ty0 + r - HALO, x = tx0 + q - HALO;
                const bool in = y >= 0 && x >= 0 && y < (int)a.h && x < (int)a.w;
                const size_t p = in ? (size_t)y * a.w + (size_t)x : 0;
                s_part[0][i] = in ? pc[p] : 0.0f;
                s_part[1][i] = in ? pc[plane + p] : 0.0f;
                s_part[2][i] = in ? pc[2 * plane + p] : 0.0f;
            }
        }
        __syncthreads();
        const int probe_items = 3 * SR * TW;
        for (int i = rank; i < probe_items; i += 256) {
            const int j = i / (SR * TW), rem = i - j * (SR * TW);
            const int r = rem / TW, col = (rem - r * TW) + HALO;
            const float* row = &s_part[j][r * SW];
            float acc = 0.0f;
#pragma unroll
            for (int d = 1; d < 6; ++d) acc += (row[col - d] + row[col + d]) * a.taps.w[5 - d];
            acc += row[col] * a.taps.w[5];
            s_h2[j][r * H2P + (col - HALO)] = acc;
        }
        __syncthreads();
        float v[3][12];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 12; ++r) v[j][r] = s_h2[j][(2 * ly + r) * H2P + lx];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float sres[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int d = 1; d < 6; ++d) acc += (v[j][o + 5 - d] + v[j][o + 5 + d]) * a.taps.w[5 - d];
                acc += v[j][o + 5] * a.taps.w[5];
                sres[j] = acc;
            }
            const float ga = gt_ch(val[o], 3);
            float chain_c = a.dl_rgb;
            if (a.mask) chain_c = chain_c * ga;
            float ge = gt_ch(val[o], c);
            if (a.composite) ge = ge + (1.0f - ga) * a.bg[c];
            const float p1 = c == 0 ? pv[o].x : (c == 1 ? pv[o].y : pv[o].z);
            const float ssim_grad = sres[0] + (2.0f * p1) * sres[1] + ge * sres[2];
            const float diff = p1 - ge;
            const float l1_sign = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
            out[o][c] = a.ssim_w * ssim_grad + a.l1_w * l1_sign * chain_c;
        }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        if (!inside[o]) continue;
        if (a.alpha_match) {  // lib.rs:392-412
            const float ga = gt_ch(val[o], 3);
            const float diff = pv[o].w - ga;
            const float sign = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
            float chain = a.dl_alpha;
            if (a.mask) chain = chain * ga;
            out[o][3] = sign * chain;
        }
        *reinterpret_cast<float4*>(&v_output[((size_t)py[o] * a.w + (size_t)pxx) * 4]) = make_float4(out[o][0], out[o][1], out[o][2], out[o][3]);
    }
}

// loss scalar -> loss_out[0];  dloss/d(out_img) -> v_output [H,W,4].
// Whole image: tile_y0 = 0, tile_y1 = ceil(h/16), v_output fully overwritten.
// Strip-wise (one frame partitioned over ranks): the caller owns tile rows [tile_y0, tile_y1) and has
// valid image rows for one more tile row plus the 5-px SSIM halo on each side (21 px).  Pass A then also runs on the two
// neighbouring tile rows (pass B blurs the SSIM partials of rows up to 5 px outside the strip), pass B and the loss sum on
// the strip only: loss_out is this strip's share of the mean (the ranks' shares add up), v_output is written for the
// strip's pixels.  Image-border zero padding is unchanged.
int launch_image_loss_fused_window(bh_ctx* ctx, const float* img_hwc4, const uint32_t* gt, uint32_t h, uint32_t w, const BhLossConfig& cfg,
                                   bool alpha_match, float dl_rgb, float dl_alpha, uint32_t tile_y0, uint32_t tile_y1, float* loss_out,
                                   float* v_output, float* loss_host, uint32_t* started_host, uint32_t started_tag) {
    const uint32_t gx = (w + LB - 1) / LB, gy = (h + LB - 1) / LB;
    if (tile_y1 > gy) tile_y1 = gy;
    if (tile_y0 >= tile_y1) return set_error(ctx, BH_ERR_INVALID_ARG, "image loss: empty tile-row window");
    const uint32_t a0 = tile_y0 > 0 ? tile_y0 - 1 : 0, a1 = tile_y1 < gy ? tile_y1 + 1 : gy;  // pass A window
    const dim3 block(TW, 16);
    const bool banded = ctx->knob_loss_bands != 0u && gx >= 16u;
    const size_t hw = (size_t)h * w;
    auto* partials = (float*)ensure(ctx, SLOT_LOSS_MAP, hw * 12 * sizeof(float));   // 9 floats per pixel used (the slot is shared with loss.hip's map)
    auto* block_sums = (float*)ensure(ctx, SLOT_MISC, (size_t)gx * gy * sizeof(float));
    if (!partials || !block_sums) return BH_ERR_OOM;
    FusedArgs a;
    a.h = h; a.w = w;
    a.l1_w = cfg.l1_weight; a.ssim_w = cfg.ssim_weight;
    a.bg[0] = cfg.bg[0]; a.bg[1] = cfg.bg[1]; a.bg[2] = cfg.bg[2];
    a.composite = cfg.composite_bg; a.mask = cfg.mask; a.alpha_match = alpha_match ? 1 : 0;
    a.dl_rgb = dl_rgb; a.dl_alpha = dl_alpha;
    a.taps = gauss_taps();
    {
        ProfScope ps(ctx, "ImageLoss");
        a.ty_base = a0;
        a.row_end = h;
        a.sum_src = nullptr; a.sum_n = 0; a.loss_out = nullptr; a.loss_host = nullptr;
        a.started_host = started_host; a.started_tag = started_tag;
        // (blocks are two tile rows tall: with an odd window the last block's lower half lies outside it — rows this rank may not
        //  have rendered — and its loss partial is not stored: the limit is the window's end a1, not the image's gy)
        a.gx = gx; a.gy_blocks = (a1 - a0 + 1) / 2;
        a.band_w = banded ? (gx + 7u) / 8u : 0u;
        const dim3 grid_a = banded ? dim3(8u * a.band_w * a.gy_blocks) : dim3(gx, a.gy_blocks);
        hipLaunchKernelGGL(loss_fused_forward_kernel, grid_a, block, 0, ctx->stream, img_hwc4, gt, partials, block_sums, a1, a);
        BH_LAUNCH_CHECK(ctx, "loss_fused_forward_kernel");
    }
    {
        ProfScope ps(ctx, "ImageLossBackward");
        a.ty_base = tile_y0;
        a.row_end = tile_y1 * LB < h ? tile_y1 * LB : h;
        // the loss = the strip's own tiles only (block_sums is indexed by absolute tile row)
        a.sum_src = block_sums + (size_t)tile_y0 * gx;
        a.sum_n = (int)((tile_y1 - tile_y0) * gx);
        a.loss_out = loss_out;
        a.loss_host = loss_host;
        a.started_host = nullptr; a.started_tag = 0;
        a.gy_blocks = (tile_y1 - tile_y0 + 1) / 2;
        const dim3 grid_b = banded ? dim3(8u * a.band_w * a.gy_blocks) : dim3(gx, a.gy_blocks);
        hipLaunchKernelGGL(loss_fused_backward_kernel, grid_b, block, 0, ctx->stream, img_hwc4, gt, partials, v_output, a);
        BH_LAUNCH_CHECK(ctx, "loss_fused_backward_kernel");
    }
    return 0;
}

int launch_image_loss_fused(bh_ctx* ctx, const float* img_hwc4, const uint32_t* gt, uint32_t h, uint32_t w, const BhLossConfig& cfg,
                            bool alpha_match, float dl_rgb, float dl_alpha, float* loss_out, float* v_output, float* loss_host, uint32_t* started_host,
                            uint32_t started_tag) {
    return launch_image_loss_fused_window(ctx, img_hwc4, gt, h, w, cfg, alpha_match, dl_rgb, dl_alpha, 0, (h + LB - 1) / LB, loss_out, v_output, loss_host,
                                          started_host, started_tag);
}

}  // namespace bh
