// refine.hip — SplatTrainer::refine on the device: prune, weighted resampling of the
// pruned budget, screen-size and gradient driven splits, covariance-aware split
// arithmetic, Adam-moment reset, opacity decay, percentile bounds.
//
// Reference: brush-train/src/train.rs:431-663 (refine), :665-822 (refine_splats),
// :848-893 (prune_points), multinomial.rs, splat_init.rs:130-160 (bounds_from_pos).
// The reference does this with ~8 host readbacks, argwhere + select chains and CPU
// weighted sampling from an unseeded RNG every `refine_every` steps.  Here:
//   plan   one pass classifies every splat (prune reasons, sampling keys, candidate
//          flags), a few scans / two radix sorts / single-thread control kernels turn the
//          reference's scalar logic into device-side scalars, and ONE 64-byte readback
//          returns the counts (the caller needs new_n to allocate the outputs);
//   apply  one kernel gathers the kept rows of all nine tensors into the caller's new
//          buffers, rewrites split parents, appends the children and applies the opacity
//          decay.
// Weighted sampling without replacement = the k smallest exponential clocks
// -ln(u_i)/w_i (Efraimidis-Spirakis), u_i from a counter-based hash of (seed, i): every
// data-parallel rank that passes the same seed takes the identical decision (the
// reference's rand::rng() would diverge across ranks).  Which valid sample is drawn is not
// contractual; everything downstream of the chosen indices is checked against the oracle.
#include <cmath>
#include <cstring>

#include "context.h"

namespace bh {

namespace {

constexpr int RF_WG = 256;
constexpr float MIN_OPACITY = 1.0f / 255.0f;  // train.rs:32
constexpr float FRAC_1_SQRT_2 = 0.70710678118654752440f;

enum Ctl : int {
    C_NKEEP = 0, C_NONFINITE, C_NPOS1, C_THR, C_NPOS3, C_OVER_TOTAL, C_K1, C_BUDGET_OVER, C_NOVER, C_K3, C_NSPLIT, C_NHIGH_NEW, C_COUNT = 16
};

struct RefineArgs {
    uint32_t n, coeffs;
    float max_allowed_bounds, cx, cy, cz;
    float growth_grad_threshold, split_at_screen_size;
    uint64_t seed;
    int no_prune;
};

BH_DEV float hash_unit(uint64_t seed, uint64_t stream, uint64_t i) {
    uint64_t z = seed + (i + 1ull) * 0x9E3779B97F4A7C15ull + stream * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return ((float)(uint32_t)(z >> 41) + 0.5f) * (1.0f / 8388608.0f);  // 23 bits: k + 0.5 is exact, u in (0,1) with uniform spacing
}

// exponential clock of weight w: smaller = sampled earlier; +inf when the weight is not a
// positive finite number (multinomial.rs:8-14 maps those to weight 0)
BH_DEV float sample_key(float w, float u) {
    if (!(w > 0.0f) || !is_finite_f32(w)) return __builtin_inff();
    return -bh_logf(u) / w;
}

__global__ __launch_bounds__(RF_WG) void refine_classify_kernel(RefineArgs a, const float* __restrict__ transforms,
                                                               const float* __restrict__ sh, const float* __restrict__ raw_opac,
                                                               const float* __restrict__ refine_norm, const float* __restrict__ vis_weight,
                                                               const float* __restrict__ max_screen, uint32_t* __restrict__ keep,
                                                               uint32_t* __restrict__ key1, uint32_t* __restrict__ key3,
                                                               uint32_t* __restrict__ over, uint32_t* __restrict__ ctl) {
    const uint32_t i = blockIdx.x * RF_WG + threadIdx.x;
    uint32_t k = 0, nonfinite = 0, pos1 = 0, thr = 0, pos3 = 0;
    if (i < a.n) {
        const float* t = transforms + (size_t)i * 10;
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 10; ++c) bad = bad || !is_finite_f32(t[c]);
        const float* s = sh + (size_t)i * a.coeffs * 3;
        for (uint32_t c = 0; c < a.coeffs * 3; ++c) bad = bad || !is_finite_f32(s[c]);
        const float ro = raw_opac[i];
        bad = bad || !is_finite_f32(ro);
        const float opac = sigmoid(ro);
        const bool alpha_low = opac < MIN_OPACITY;                                   // train.rs:493
        const bool scale_big = bh_expf(t[7]) > a.max_allowed_bounds || bh_expf(t[8]) > a.max_allowed_bounds || bh_expf(t[9]) > a.max_allowed_bounds;
        const bool out_of_bounds = __builtin_fabsf(t[0] - a.cx) > a.max_allowed_bounds || __builtin_fabsf(t[1] - a.cy) > a.max_allowed_bounds ||
                                   __builtin_fabsf(t[2] - a.cz) > a.max_allowed_bounds;
        const bool prune = !a.no_prune && (alpha_low || scale_big || out_of_bounds || bad);
        k = prune ? 0u : 1u;
        nonfinite = bad ? 1u : 0u;
        const bool vis = vis_weight[i] > 0.0f;                                         // stats.rs:52-54
        const float w1 = (k && vis) ? opac : 0.0f;                                     // train.rs:530-533
        const float k1 = sample_key(w1, hash_unit(a.seed, 1, i));
        const float rn = refine_norm[i];
        const bool above = rn > a.growth_grad_threshold && vis;                        // stats.rs:23-28
        const float w3 = (k && above) ? rn : 0.0f;                                     // train.rs:613
        const float k3 = sample_key(w3, hash_unit(a.seed, 3, i));
        keep[i] = k;
        key1[i] = f2u(k1);
        key3[i] = f2u(k3);
        over[i] = (k && a.split_at_screen_size > 0.0f && max_screen[i] > a.split_at_screen_size && vis) ? 1u : 0u;  // stats.rs:32-37
        pos1 = k1 < __builtin_inff() ? 1u : 0u;
        thr = (k && above) ? 1u : 0u;   // counted on the pruned set like the reference (the refiner is pruned first)
        pos3 = k3 < __builtin_inff() ? 1u : 0u;
    }
    const uint32_t c_bad = (uint32_t)__popcll(__ballot(nonfinite)), c_p1 = (uint32_t)__popcll(__ballot(pos1));
    const uint32_t c_thr = (uint32_t)__popcll(__ballot(thr)), c_p3 = (uint32_t)__popcll(__ballot(pos3));
    if ((threadIdx.x & 63) == 0) {
        if (c_bad) atomicAdd(&ctl[C_NONFINITE], c_bad);
        if (c_p1) atomicAdd(&ctl[C_NPOS1], c_p1);
        if (c_thr) atomicAdd(&ctl[C_THR], c_thr);
        if (c_p3) atomicAdd(&ctl[C_NPOS3], c_p3);
    }
}

__global__ void refine_control1_kernel(uint32_t n, uint32_t max_splats, const uint32_t* keep, const uint32_t* keep_excl, uint32_t* ctl) {
    const uint32_t n_keep = keep_excl[n - 1] + keep[n - 1];
    const uint32_t pruned = n - n_keep;
    const uint32_t k1 = pruned < ctl[C_NPOS1] ? pruned : ctl[C_NPOS1];  // always refill the pruned budget (train.rs:527-540)
    ctl[C_NKEEP] = n_keep;
    ctl[C_K1] = k1;
    const uint32_t cur = n_keep + k1;
    ctl[C_BUDGET_OVER] = max_splats > cur ? max_splats - cur : 0u;        // train.rs:572-576 (saturating)
}

// split[idx[r]] = 1 for the ctl[kslot] smallest keys; counts the newly set ones in ctl[count_slot]
__global__ __launch_bounds__(RF_WG) void refine_mark_sorted_kernel(uint32_t n, const uint32_t* __restrict__ sorted_idx, uint32_t* __restrict__ split,
                                                                  uint32_t* ctl, int kslot, int count_slot) {
    const uint32_t r = blockIdx.x * RF_WG + threadIdx.x;
    bool fresh = false;
    if (r < n && r < ctl[kslot]) {
        const uint32_t i = sorted_idx[r];
        fresh = split[i] == 0u;
        split[i] = 1u;
    }
    const uint32_t c = (uint32_t)__popcll(__ballot(fresh));
    if (count_slot >= 0 && (threadIdx.x & 63) == 0 && c) atomicAdd(&ctl[count_slot], c);
}

__global__ __launch_bounds__(RF_WG) void refine_over_candidates_kernel(uint32_t n, const uint32_t* __restrict__ over, const uint32_t* __restrict__ split,
                                                                      uint32_t* __restrict__ cand) {
    const uint32_t i = blockIdx.x * RF_WG + threadIdx.x;
    if (i < n) cand[i] = (over[i] && !split[i]) ? 1u : 0u;
}

// oversized splats in index order until the budget runs out (train.rs:577-585)
__global__ __launch_bounds__(RF_WG) void refine_mark_over_kernel(uint32_t n, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ cand_excl,
                                                                uint32_t* __restrict__ split, const uint32_t* ctl) {
    const uint32_t i = blockIdx.x * RF_WG + threadIdx.x;
    if (i < n && cand[i] && cand_excl[i] < ctl[C_BUDGET_OVER]) split[i] = 1u;
}

__global__ void refine_control2_kernel(uint32_t n, uint32_t max_splats, float growth_select_fraction, int growing, const uint32_t* cand,
                                       const uint32_t* cand_excl, uint32_t* ctl) {
    const uint32_t total = cand_excl[n - 1] + cand[n - 1];
    const uint32_t n_over = total < ctl[C_BUDGET_OVER] ? total : ctl[C_BUDGET_OVER];
    ctl[C_OVER_TOTAL] = total;
    ctl[C_NOVER] = n_over;
    uint32_t k3 = 0;
    if (growing) {  // train.rs:589-623
        const uint32_t pruned = n - ctl[C_NKEEP];
        const float want = __builtin_roundf((float)ctl[C_THR] * growth_select_fraction);  // f32::round: half away from zero
        const uint32_t grow = want <= 0.0f ? 0u : (want >= 4294967040.0f ? 0xFFFFFFFFu : (uint32_t)want);
        const uint32_t sample_high = grow > pruned ? grow - pruned : 0u;
        const uint32_t cur = ctl[C_NKEEP] + ctl[C_K1] + n_over;
        const uint32_t headroom = max_splats > cur ? max_splats - cur : 0u;
        k3 = sample_high < headroom ? sample_high : headroom;
        k3 = k3 < ctl[C_NPOS3] ? k3 : ctl[C_NPOS3];
    }
    ctl[C_K3] = k3;
}

__global__ void refine_control3_kernel(uint32_t n, const uint32_t* split, const uint32_t* split_excl, uint32_t* ctl) {
    ctl[C_NSPLIT] = split_excl[n - 1] + split[n - 1];
}

struct ApplyArgs {
    uint32_t n, coeffs;
    float split_at_screen_size, minus_opac;
};

BH_DEV float inv_sigmoid(float x) { return bh_logf(x / (1.0f - x)); }           // train.rs:92-94
BH_DEV float powf_pos(float x, float y) { return x > 0.0f ? bh_expf(y * bh_logf(x)) : 0.0f; }
// train.rs:812-816
BH_DEV float decay_opacity(float raw, float minus_opac) { return inv_sigmoid(clampf(sigmoid(raw) - minus_opac, 1.0e-12f, 1.0f - 1.0e-12f)); }

__global__ __launch_bounds__(RF_WG) void refine_apply_kernel(
    ApplyArgs a, const uint32_t* __restrict__ ctl, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ keep_excl,
    const uint32_t* __restrict__ split, const uint32_t* __restrict__ split_excl, const float* __restrict__ max_screen,
    const float* __restrict__ tr_in, const float* __restrict__ sh_in, const float* __restrict__ op_in,
    const float* __restrict__ m1t_in, const float* __restrict__ m2t_in, const float* __restrict__ m1s_in, const float* __restrict__ m2s_in,
    const float* __restrict__ m1o_in, const float* __restrict__ m2o_in,
    float* __restrict__ tr, float* __restrict__ sh, float* __restrict__ op,
    float* __restrict__ m1t, float* __restrict__ m2t, float* __restrict__ m1s, float* __restrict__ m2s, float* __restrict__ m1o, float* __restrict__ m2o) {
    const uint32_t i = blockIdx.x * RF_WG + threadIdx.x;
    if (i >= a.n || !keep[i]) return;
    const uint32_t j = keep_excl[i];
    const uint32_t sc = a.coeffs * 3;
    float t[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) t[c] = tr_in[(size_t)i * 10 + c];
    float raw = op_in[i];
    const bool is_split = split[i] != 0u;
    if (!is_split) {
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            m1t[(size_t)j * 10 + c] = m1t_in[(size_t)i * 10 + c];
            m2t[(size_t)j * 10 + c] = m2t_in[(size_t)i * 10 + c];
        }
        for (uint32_t c = 0; c < sc; ++c) {
            sh[(size_t)j * sc + c] = sh_in[(size_t)i * sc + c];
            m1s[(size_t)j * sc + c] = m1s_in[(size_t)i * sc + c];
        }
        m2s[j] = m2s_in[i];
        m1o[j] = m1o_in[i];
        m2o[j] = m2o_in[i];
    } else {
        // ---- refine_splats (train.rs:665-806) ------------------------------------------------
        const uint32_t child = ctl[C_NKEEP] + split_excl[i];
        const float qn = __builtin_fmaxf(__builtin_sqrtf(((t[3] * t[3] + t[4] * t[4]) + t[5] * t[5]) + t[6] * t[6]), 1.0e-32f);
        const float qw = t[3] / qn, qx = t[4] / qn, qy = t[5] / qn, qz = t[6] / qn;
        const float s0 = bh_expf(t[7]), s1 = bh_expf(t[8]), s2 = bh_expf(t[9]);
        const float cur_opac = sigmoid(raw);
        const float inv_opac = 1.0f - cur_opac;
        const float new_opac = 1.0f - powf_pos(inv_opac, FRAC_1_SQRT_2);
        const float new_raw = inv_sigmoid(clampf(new_opac, MIN_OPACITY, 1.0f - MIN_OPACITY));
        const float q0 = s0 * s0, q1 = s1 * s1, q2 = s2 * s2;
        const float max_sq = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(q0, q1), q2), 1.0e-30f);
        float k_max = FRAC_1_SQRT_2;
        if (a.split_at_screen_size > 0.0f)
            k_max = __builtin_fminf((1.0f / __builtin_fmaxf(max_screen[i], 1.0e-6f)) * a.split_at_screen_size, FRAC_1_SQRT_2);
        const float sv[3] = {s0, s1, s2}, qv[3] = {q0, q1, q2};
        float off[3], new_log[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ratio = qv[c] / max_sq;
            const float k_axis = -(ratio * (-k_max + 1.0f)) + 1.0f;
            const float offset_factor = __builtin_sqrtf(__builtin_fmaxf(-(k_axis * k_axis) + 1.0f, 0.0f));
            off[c] = offset_factor * sv[c];
            new_log[c] = t[7 + c] + bh_logf(k_axis);
        }
        // quaternion_vec_multiply (quat_vec.rs:4-45)
        const float qw2 = qw * qw, qx2 = qx * qx, qy2 = qy * qy, qz2 = qz * qz;
        const float xy = qx * qy, xz = qx * qz, yz = qy * qz, wx = qw * qx, wy = qw * qy, wz = qw * qz;
        const float vx = off[0], vy = off[1], vz = off[2];
        const float sx = (qw2 + qx2 - qy2 - qz2) * vx + (xy * vy + xz * vz + wy * vz - wz * vy) * 2.0f;
        const float sy = (qw2 - qx2 + qy2 - qz2) * vy + (xy * vx + yz * vz + wz * vx - wx * vz) * 2.0f;
        const float sz = (qw2 - qx2 - qy2 + qz2) * vz + (xz * vx + yz * vy + wx * vy - wy * vx) * 2.0f;
        float* ct = tr + (size_t)child * 10;
        ct[0] = t[0] + sx; ct[1] = t[1] + sy; ct[2] = t[2] + sz;
        ct[3] = qw; ct[4] = qx; ct[5] = qy; ct[6] = qz;
        ct[7] = new_log[0]; ct[8] = new_log[1]; ct[9] = new_log[2];
        op[child] = decay_opacity(new_raw, a.minus_opac);
        // parent: scatter-ADD of (-samples) and of the log-scale / opacity differences (train.rs:738-752)
        t[0] = t[0] + (-sx); t[1] = t[1] + (-sy); t[2] = t[2] + (-sz);
#pragma unroll
        for (int c = 0; c < 3; ++c) t[7 + c] = t[7 + c] + (new_log[c] - t[7 + c]);
        raw = raw + (new_raw - raw);
        for (uint32_t c = 0; c < sc; ++c) {
            const float v = sh_in[(size_t)i * sc + c];
            sh[(size_t)j * sc + c] = v;
            sh[(size_t)child * sc + c] = v;
            m1s[(size_t)j * sc + c] = 0.0f;
            m1s[(size_t)child * sc + c] = 0.0f;
        }
        // both halves of a split start with zero Adam moments (train.rs:762-806)
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            m1t[(size_t)j * 10 + c] = 0.0f; m2t[(size_t)j * 10 + c] = 0.0f;
            m1t[(size_t)child * 10 + c] = 0.0f; m2t[(size_t)child * 10 + c] = 0.0f;
        }
        m2s[j] = 0.0f; m2s[child] = 0.0f;
        m1o[j] = 0.0f; m2o[j] = 0.0f; m1o[child] = 0.0f; m2o[child] = 0.0f;
    }
#pragma unroll
    for (int c = 0; c < 10; ++c) tr[(size_t)j * 10 + c] = t[c];
    op[j] = decay_opacity(raw, a.minus_opac);
}

// ---- percentile bounds (splat_init.rs:130-160) ------------------------------------------------
// float -> u32 whose unsigned order is f32::total_cmp order; non-finite values go last
BH_DEV uint32_t total_order_key(float v) {
    if (!is_finite_f32(v)) return 0xFFFFFFFFu;
    const uint32_t b = f2u(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
BH_DEV float from_total_order_key(uint32_t k) { return u2f((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

__global__ __launch_bounds__(RF_WG) void bounds_keys_kernel(uint32_t n, const float* __restrict__ transforms, int axis, uint32_t* __restrict__ keys,
                                                           uint32_t* __restrict__ finite_count) {
    const uint32_t i = blockIdx.x * RF_WG + threadIdx.x;
    bool fin = false;
    if (i < n) {
        const float v = transforms[(size_t)i * 10 + axis];
        fin = is_finite_f32(v);
        keys[i] = total_order_key(v);
    }
    const uint32_t c = (uint32_t)__popcll(__ballot(fin));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&finite_count[axis], c);
}

__global__ void bounds_pick_kernel(const uint32_t* sorted_keys, const uint32_t* finite_count, int axis, float percentile, float* out /*[6]: min xyz, max xyz*/) {
    const uint32_t n = finite_count[axis];
    if (n == 0) { out[axis] = __builtin_nanf(""); out[3 + axis] = __builtin_nanf(""); return; }
    const uint32_t lo = (uint32_t)((1.0f - percentile) / 2.0f * (float)n);
    uint32_t hi = (uint32_t)((1.0f + percentile) / 2.0f * (float)n);
    hi = hi < n - 1 ? hi : n - 1;
    out[axis] = from_total_order_key(sorted_keys[lo < n ? lo : n - 1]);
    out[3 + axis] = from_total_order_key(sorted_keys[hi]);
}

}  // namespace

}  // namespace bh

using namespace bh;

extern "C" {

static int refine_plan_impl(bh_ctx* ctx, const BhRefineConfig* cfg, const BhTrainState* st, BhRefineStats* out, int no_prune);

int bh_refine_plan(bh_ctx* ctx, const BhRefineConfig* cfg, const BhTrainState* st, BhRefineStats* out) {
    return refine_plan_impl(ctx, cfg, st, out, 0);
}

static int refine_plan_impl(bh_ctx* ctx, const BhRefineConfig* cfg, const BhTrainState* st, BhRefineStats* out, int no_prune) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!cfg || !st || !out) return set_error(ctx, BH_ERR_INVALID_ARG, "refine_plan: null argument");
    const uint32_t n = st->n;
    if (n == 0) return set_error(ctx, BH_ERR_INVALID_ARG, "refine_plan: no splats");
    if (!st->transforms || !st->sh_coeffs || !st->raw_opacities || !st->refine_weight_norm || !st->vis_weight || !st->max_screen_size)
        return set_error(ctx, BH_ERR_INVALID_ARG, "refine_plan: null state tensor");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    // plan buffers: 9 x [n] u32 + control block
    uint32_t* buf = (uint32_t*)ensure(ctx, SLOT_REFINE, ((size_t)n * 10 + C_COUNT) * 4);
    if (!buf) return BH_ERR_OOM;
    uint32_t* ctl = buf;
    uint32_t* keep = buf + C_COUNT;
    uint32_t* keep_excl = keep + n;
    uint32_t* split = keep_excl + n;
    uint32_t* split_excl = split + n;
    uint32_t* key = split_excl + n;      // key1, later key3
    uint32_t* key3 = key + n;
    uint32_t* over = key3 + n;           // later reused as the candidate flags
    uint32_t* tmp_a = over + n;          // sorted keys / candidate scan
    uint32_t* tmp_b = tmp_a + n;         // sorted indices
    uint32_t* cand = tmp_b + n;
    BH_HIP(ctx, hipMemsetAsync(ctl, 0, C_COUNT * 4, ctx->stream));
    BH_HIP(ctx, hipMemsetAsync(split, 0, (size_t)n * 4, ctx->stream));
    RefineArgs a;
    a.n = n;
    a.coeffs = (st->sh_degree + 1) * (st->sh_degree + 1);
    const float ext_max = std::fmax(std::fmax(cfg->bounds_extent[0], cfg->bounds_extent[1]), cfg->bounds_extent[2]);
    a.max_allowed_bounds = ext_max * 100.0f;  // train.rs:485
    a.cx = cfg->bounds_center[0]; a.cy = cfg->bounds_center[1]; a.cz = cfg->bounds_center[2];
    a.growth_grad_threshold = cfg->growth_grad_threshold;
    a.split_at_screen_size = cfg->split_at_screen_size;
    a.seed = cfg->seed;
    a.no_prune = no_prune;
    const dim3 grid((n + RF_WG - 1) / RF_WG), block(RF_WG);
    hipLaunchKernelGGL(refine_classify_kernel, grid, block, 0, ctx->stream, a, st->transforms, st->sh_coeffs, st->raw_opacities,
                       st->refine_weight_norm, st->vis_weight, st->max_screen_size, keep, key, key3, over, ctl);
    BH_LAUNCH_CHECK(ctx, "refine_classify_kernel");
    BH_TRY(prefix_sum(ctx, keep, nullptr, n, keep_excl, true));
    hipLaunchKernelGGL(refine_control1_kernel, dim3(1), dim3(1), 0, ctx->stream, n, cfg->max_splats, keep, keep_excl, ctl);
    // 1) refill the pruned budget: sample by opacity x visibility
    BH_TRY(radix_argsort(ctx, key, nullptr, n, 32, tmp_a, tmp_b));
    hipLaunchKernelGGL(refine_mark_sorted_kernel, grid, block, 0, ctx->stream, n, tmp_b, split, ctl, (int)C_K1, -1);
    // 2) oversized on screen, in index order, within the max_splats budget
    hipLaunchKernelGGL(refine_over_candidates_kernel, grid, block, 0, ctx->stream, n, over, split, cand);
    BH_TRY(prefix_sum(ctx, cand, nullptr, n, tmp_a, true));
    hipLaunchKernelGGL(refine_mark_over_kernel, grid, block, 0, ctx->stream, n, cand, tmp_a, split, ctl);
    const int growing = cfg->iter < cfg->growth_stop_iter ? 1 : 0;
    hipLaunchKernelGGL(refine_control2_kernel, dim3(1), dim3(1), 0, ctx->stream, n, cfg->max_splats, cfg->growth_select_fraction, growing, cand, tmp_a, ctl);
    // 3) high positional gradient: sample by refine weight among those above the threshold
    BH_TRY(radix_argsort(ctx, key3, nullptr, n, 32, tmp_a, tmp_b));
    hipLaunchKernelGGL(refine_mark_sorted_kernel, grid, block, 0, ctx->stream, n, tmp_b, split, ctl, (int)C_K3, (int)C_NHIGH_NEW);
    BH_TRY(prefix_sum(ctx, split, nullptr, n, split_excl, true));
    hipLaunchKernelGGL(refine_control3_kernel, dim3(1), dim3(1), 0, ctx->stream, n, split, split_excl, ctl);
    BH_LAUNCH_CHECK(ctx, "refine plan kernels");
    uint32_t* hc = ctx->host_counters;
    BH_HIP(ctx, hipMemcpyAsync(hc, ctl, C_COUNT * 4, hipMemcpyDeviceToHost, ctx->stream));
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    deliver_pending_loss(ctx);   // a train step queued before this plan has finished now: its loss word is final
    if (hc[C_NKEEP] == 0 && !no_prune) {  // prune_points: "Trying to create empty splat!" -> nothing is pruned (train.rs:866-869)
        return refine_plan_impl(ctx, cfg, st, out, 1);
    }
    out->num_pruned = n - hc[C_NKEEP];
    out->num_pruned_non_finite = hc[C_NONFINITE];
    out->num_added = hc[C_NSPLIT];
    out->num_split_oversized = hc[C_NOVER];
    out->num_split_high_grad = hc[C_NHIGH_NEW];
    out->total_splats = hc[C_NKEEP] + hc[C_NSPLIT];
    out->num_resampled = hc[C_K1];
    ctx->refine_n = n;
    ctx->refine_new_n = out->total_splats;
    return 0;
}

const uint32_t* bh_refine_plan_flags(bh_ctx* ctx, int which) {
    if (!ctx || !ctx->slots[SLOT_REFINE].ptr || ctx->refine_n == 0) return nullptr;
    uint32_t* buf = (uint32_t*)ctx->slots[SLOT_REFINE].ptr;
    const size_t n = ctx->refine_n;
    uint32_t* keep = buf + C_COUNT;
    switch (which) {
        case 0: return keep;             // keep flags
        case 1: return keep + n;         // exclusive scan of keep (new index of a kept splat)
        case 2: return keep + 2 * n;     // split flags
        case 3: return keep + 3 * n;     // exclusive scan of split (child slot = n_keep + this)
        default: return nullptr;
    }
}

int bh_refine_apply(bh_ctx* ctx, const BhRefineConfig* cfg, const BhTrainState* in, BhTrainState* out) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!cfg || !in || !out) return set_error(ctx, BH_ERR_INVALID_ARG, "refine_apply: null argument");
    if (ctx->refine_n == 0 || ctx->refine_n != in->n) return set_error(ctx, BH_ERR_STATE, "refine_apply needs a preceding bh_refine_plan on the same state");
    if (out->n != ctx->refine_new_n || out->sh_degree != in->sh_degree) return set_error(ctx, BH_ERR_INVALID_ARG, "refine_apply: output state must be sized for total_splats of the plan");
    if (!out->transforms || !out->sh_coeffs || !out->raw_opacities || !out->m1_transforms || !out->m2_transforms || !out->m1_sh || !out->m2_sh ||
        !out->m1_opac || !out->m2_opac || !out->refine_weight_norm || !out->vis_weight || !out->max_screen_size)
        return set_error(ctx, BH_ERR_INVALID_ARG, "refine_apply: null output tensor");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = in->n;
    uint32_t* buf = (uint32_t*)ctx->slots[SLOT_REFINE].ptr;
    uint32_t* ctl = buf;
    uint32_t* keep = buf + C_COUNT;
    ApplyArgs a;
    a.n = n;
    a.coeffs = (in->sh_degree + 1) * (in->sh_degree + 1);
    a.split_at_screen_size = cfg->split_at_screen_size;
    const float train_t = std::fmin(std::fmax((float)cfg->iter / (float)cfg->total_train_iters, 0.0f), 1.0f);  // train.rs:808-811
    a.minus_opac = cfg->opac_decay * (1.0f - train_t);
    hipLaunchKernelGGL(refine_apply_kernel, dim3((n + RF_WG - 1) / RF_WG), dim3(RF_WG), 0, ctx->stream, a, ctl, keep, keep + n, keep + 2 * (size_t)n,
                       keep + 3 * (size_t)n, in->max_screen_size, in->transforms, in->sh_coeffs, in->raw_opacities, in->m1_transforms, in->m2_transforms,
                       in->m1_sh, in->m2_sh, in->m1_opac, in->m2_opac, out->transforms, out->sh_coeffs, out->raw_opacities, out->m1_transforms,
                       out->m2_transforms, out->m1_sh, out->m2_sh, out->m1_opac, out->m2_opac);
    BH_LAUNCH_CHECK(ctx, "refine_apply_kernel");
    // a fresh RefineRecord (train.rs:442-445 takes it, step() re-creates it with zeros)
    const size_t nb = (size_t)out->n * 4;
    BH_HIP(ctx, hipMemsetAsync(out->refine_weight_norm, 0, nb, ctx->stream));
    BH_HIP(ctx, hipMemsetAsync(out->vis_weight, 0, nb, ctx->stream));
    BH_HIP(ctx, hipMemsetAsync(out->max_screen_size, 0, nb, ctx->stream));
    out->step_count = in->step_count;
    ctx->refine_n = 0;
    return 0;
}

int bh_splat_bounds(bh_ctx* ctx, const float* transforms, uint32_t n, float percentile, float* center /*host[3]*/, float* extent /*host[3]*/) {
    if (!ctx) return BH_ERR_INVALID_ARG;
    if (!center || !extent || (n > 0 && !transforms)) return set_error(ctx, BH_ERR_INVALID_ARG, "splat_bounds: null argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    float mn[3] = {-1.f, -1.f, -1.f}, mx[3] = {1.f, 1.f, 1.f};  // fallback unit box (splat_init.rs:141-143)
    if (n > 0) {
        uint32_t* buf = (uint32_t*)ensure(ctx, SLOT_REFINE_BOUNDS, ((size_t)n * 3 + 16) * 4);
        if (!buf) return BH_ERR_OOM;
        uint32_t* counts = buf;           // [3] finite counts, then [6] floats
        float* picks = (float*)(buf + 4);
        uint32_t* keys = buf + 16;
        uint32_t* sorted = keys + n;
        uint32_t* idx = sorted + n;
        BH_HIP(ctx, hipMemsetAsync(buf, 0, 64, ctx->stream));
        for (int axis = 0; axis < 3; ++axis) {
            hipLaunchKernelGGL(bounds_keys_kernel, dim3((n + RF_WG - 1) / RF_WG), dim3(RF_WG), 0, ctx->stream, n, transforms, axis, keys, counts);
            BH_TRY(radix_argsort(ctx, keys, nullptr, n, 32, sorted, idx));
            hipLaunchKernelGGL(bounds_pick_kernel, dim3(1), dim3(1), 0, ctx->stream, sorted, counts, axis, percentile, picks);
        }
        BH_LAUNCH_CHECK(ctx, "bounds kernels");
        float* hp = reinterpret_cast<float*>(ctx->host_counters);
        BH_HIP(ctx, hipMemcpyAsync(hp, picks, 24, hipMemcpyDeviceToHost, ctx->stream));
        BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        deliver_pending_loss(ctx);
        bool ok = true;
        for (int k = 0; k < 6; ++k) ok = ok && std::isfinite(hp[k]);
        if (ok) {
            for (int k = 0; k < 3; ++k) { mn[k] = hp[k]; mx[k] = hp[3 + k]; }
        }
    }
    for (int k = 0; k < 3; ++k) {  // BoundingBox::from_min_max (bounding_box.rs:8-13)
        center[k] = (mx[k] + mn[k]) / 2.0f;
        extent[k] = (mx[k] - mn[k]) / 2.0f;
    }
    return 0;
}

}  // extern "C"
