// optim.hip — AdamScaled step, refine statistics, visibility-gated mean noise.
//
// Reference: brush-train/src/adam_scaled.rs:75-147, stats.rs:40-50,
// train.rs:389-416.  The reference expresses these as a few dozen burn tensor ops
// (fused opportunistically by burn-fusion); here each is ONE pass over HBM:
// 28 B/element for Adam (p,g,m,v read; p,m,v written), the floor for the update.
#include <cstdlib>

#include "context.h"
#include "device_rng.h"

namespace bh {

constexpr int OPT_WG = 256;

struct AdamArgs {
    float beta1, beta2, f1, f2, bc1, bc2, eps, lr;
    uint32_t first;  // t == 1: initialise the moments instead of decaying them
};

BH_DEV bool same_bits(float a, float b) { return f2u(a) == f2u(b); }
BH_DEV bool same_bits4(const float4& a, const float4& b) {
    return ((f2u(a.x) ^ f2u(b.x)) | (f2u(a.y) ^ f2u(b.y)) | (f2u(a.z) ^ f2u(b.z)) | (f2u(a.w) ^ f2u(b.w))) == 0u;
}

BH_DEV void adam_elem(float& p, float g, float& m1, float v, const AdamArgs& a, float step) {
    const float m1c = m1 / a.bc1;
    const float m2c = v / a.bc2;
    const float upd = m1c / (__builtin_sqrtf(m2c) + a.eps);
    p = p - upd * step;
}

// full second moment: one thread per element
__global__ __launch_bounds__(OPT_WG) void adam_full_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                          float* __restrict__ m1, float* __restrict__ m2, uint64_t count,
                                                          uint32_t row_len, const float* __restrict__ col_scale, AdamArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * OPT_WG + threadIdx.x;
    if (i >= count) return;
    const float g = grad[i];
    float mm1 = a.first ? g * a.f1 : m1[i] * a.beta1 + g * a.f1;
    const float gsq = g * g;
    const float mm2 = a.first ? gsq * a.f2 : m2[i] * a.beta2 + gsq * a.f2;
    m1[i] = mm1;
    m2[i] = mm2;
    const float step = col_scale ? col_scale[i % row_len] * a.lr : a.lr;
    float p = param[i];
    adam_elem(p, g, mm1, mm2, a, step);
    param[i] = p;
}

// second moment reduced to one scalar per row (adam_scaled.rs:99-104,152-165).
// A block owns 256 consecutive rows: the gradient tile is staged through LDS with
// coalesced loads (row pitch row_len+1: conflict-free for the per-row pass), thread r
// forms row r's sum of squares sequentially in index order (the oracle's order, so the
// result is bit-identical), and the update pass runs element-wise, coalesced, reading
// the row's v back from LDS.  (One thread per row made every access a 4*row_len-byte
// stride: 4.3 ms at 1 M splats / SH degree 3; this layout moves the same bytes at HBM speed.)
constexpr int ADAM_ROWS = 256;

__global__ __launch_bounds__(OPT_WG) void adam_rowreduced_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                                float* __restrict__ m1, float* __restrict__ m2, uint64_t rows,
                                                                uint32_t row_len, const float* __restrict__ col_scale, AdamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float* s_g = s_dyn;                                   // [ADAM_ROWS][row_len + 1]
    float* s_v = s_dyn + ADAM_ROWS * (row_len + 1);       // [ADAM_ROWS]
    const uint64_t row0 = (uint64_t)blockIdx.x * ADAM_ROWS;
    const uint32_t nrows = (uint32_t)(rows - row0 < (uint64_t)ADAM_ROWS ? rows - row0 : (uint64_t)ADAM_ROWS);
    const uint32_t count = nrows * row_len;
    const uint64_t base = row0 * row_len;
    const float rcp_len = 1.0f / (float)row_len;
    const uint32_t pitch = row_len + 1;
    for (uint32_t e = threadIdx.x; e < count; e += OPT_WG) {
        const uint32_t r = (uint32_t)(((float)e + 0.5f) * rcp_len);  // e / row_len, exact for e < 2^16
        const uint32_t c = e - r * row_len;
        s_g[r * pitch + c] = grad[base + e];
    }
    __syncthreads();
    if (threadIdx.x < nrows) {
        const float* g = s_g + threadIdx.x * pitch;
        float s = 0.0f;
        for (uint32_t c = 0; c < row_len; ++c) s += g[c] * g[c];
        const float row_gsq = s / (float)row_len;
        const uint64_t r = row0 + threadIdx.x;
        const float v = a.first ? row_gsq * a.f2 : m2[r] * a.beta2 + row_gsq * a.f2;
        m2[r] = v;
        s_v[threadIdx.x] = v;
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < count; e += OPT_WG) {
        const uint32_t r = (uint32_t)(((float)e + 0.5f) * rcp_len);
        const uint32_t c = e - r * row_len;
        const uint64_t i = base + e;
        const float gi = s_g[r * pitch + c];
        float mm1 = a.first ? gi * a.f1 : m1[i] * a.beta1 + gi * a.f1;
        m1[i] = mm1;
        const float step = col_scale ? col_scale[c] * a.lr : a.lr;
        float p = param[i];
        adam_elem(p, gi, mm1, s_v[r], a, step);
        param[i] = p;
    }
}

// compiler-rt __powisf2: the lowering of Rust's f32::powi (adam_scaled.rs:131-138)
static float powi_f32(float a, int b) {
    const bool recip = b < 0;
    float r = 1.0f;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

int launch_adam(bh_ctx* ctx, float* param, const float* grad, float* m1, float* m2, uint64_t rows, uint32_t row_len,
                const float* col_scale, float lr, uint32_t t, bool reduce_m2, float beta1, float beta2, float eps) {
    if (rows == 0 || row_len == 0) return 0;
    if (t == 0) return set_error(ctx, BH_ERR_INVALID_ARG, "adam: t is 1-based");
    AdamArgs a;
    a.beta1 = beta1; a.beta2 = beta2;
    a.f1 = 1.0f - beta1; a.f2 = 1.0f - beta2;
    a.bc1 = 1.0f - powi_f32(beta1, (int)t);
    a.bc2 = 1.0f - powi_f32(beta2, (int)t);
    a.eps = eps; a.lr = lr;
    a.first = t == 1 ? 1u : 0u;
    if (reduce_m2) {
        if (row_len > 255) return set_error(ctx, BH_ERR_UNSUPPORTED, "adam (reduced second moment): row_len must be <= 255");
        const uint64_t nb = (rows + ADAM_ROWS - 1) / ADAM_ROWS;
        const size_t lds = ((size_t)ADAM_ROWS * (row_len + 1) + ADAM_ROWS) * sizeof(float);
        if (lds > 64 * 1024 && !ctx->adam_lds_raised) {  // above the default dynamic-LDS limit (row_len > 62): opt in once per ctx (the attribute is per device)
            BH_HIP(ctx, hipFuncSetAttribute((const void*)adam_rowreduced_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ctx->adam_lds_raised = true;
        }
        hipLaunchKernelGGL(adam_rowreduced_kernel, dim3((unsigned)nb), dim3(OPT_WG), lds, ctx->stream, param, grad, m1, m2, rows, row_len, col_scale, a);
        BH_LAUNCH_CHECK(ctx, "adam_rowreduced_kernel");
    } else {
        const uint64_t count = rows * row_len;
        const uint64_t nb = (count + OPT_WG - 1) / OPT_WG;
        hipLaunchKernelGGL(adam_full_kernel, dim3((unsigned)nb), dim3(OPT_WG), 0, ctx->stream, param, grad, m1, m2, count, row_len, col_scale, a);
        BH_LAUNCH_CHECK(ctx, "adam_full_kernel");
    }
    return 0;
}

// ---------------------------------------------------------------------------
// The train step's whole "after the backward" tail in ONE launch: RefineRecord::gather_stats
// (stats.rs:40-50) + the three AdamScaled updates (train.rs:300-381) with the data-parallel
// 1/K gradient scale folded in.  A block owns 256 consecutive splats, so every tensor it touches
// is one contiguous run: transforms 2560 floats, SH 256*3C floats (staged through LDS for the
// per-row second moment, as adam_rowreduced_kernel), opacity / statistics 256 floats.  The
// element arithmetic is that of the stand-alone kernels above (bit-identical results); what
// goes away is five launches, the lr-table upload and the separate gradient-scale pass.
// ---------------------------------------------------------------------------
struct UpdateArgs {
    AdamArgs a;            // betas / bias corrections of step t (shared: the three params step together)
    float lr_sh, lr_opac;  // transforms use lr 1.0 with the per-column table (train.rs:328-350)
    float gscale;          // 1/world for data parallel over cameras, else 1
    uint32_t n, sh_len;    // splats, 3*C
    uint32_t vis_clamp;    // tile-partitioned frame: visible arrives summed over strips -> min(v, 1)
    uint32_t dormant_skip; // 0 = BH_UPDATE_NO_DORMANT (A/B, tests): dormant splats are fetched and updated like everyone else
    uint32_t masked;       // the gradient tensors were not zero-filled: row i holds a gradient iff the sign bit of refine_weight[i] is set (K18's mark), else it is 0
    float tab_t[10];       // lr_mean x3, lr_rotation x4, lr_scale x3
    float tab_sh[75];      // 1 for the DC coefficient, 1/lr_coeffs_sh_scale for the rest
    // visibility-gated noise on the means (train.rs:389-416) drawn on the device and added right behind the Adam update
    uint32_t noise_on, noise_step;
    float noise_scale, noise_clamp;
    uint64_t noise_seed;
};

// VEC: every tensor base is 16-byte aligned -> 128-bit loads / stores (a block's region starts at a multiple of
// 256 rows, so it is aligned whenever the tensor is; a float4 may straddle two rows, rows/columns are resolved per
// component).  The element arithmetic is identical either way.
// ROWS: splats per block (multiple of 4, <= OPT_WG) — fewer for long SH rows keeps more blocks resident per CU.
// S_IT: float4s of the block's SH rows per thread when the loads are issued up front (0: every section fetches its own inputs).
template <bool VEC, int ROWS, int S_IT_>
__global__ __launch_bounds__(OPT_WG) void train_update_kernel(
    float* __restrict__ transforms, float* __restrict__ m1_t, float* __restrict__ m2_t, const float* __restrict__ g_t,
    float* __restrict__ sh, float* __restrict__ m1_sh, float* __restrict__ m2_sh, const float* __restrict__ g_sh,
    float* __restrict__ opac, float* __restrict__ m1_o, float* __restrict__ m2_o, const float* __restrict__ g_o,
    float* __restrict__ refine_weight_norm, float* __restrict__ vis_weight, float* __restrict__ max_screen_size,
    const float* __restrict__ refine_weight, const float* __restrict__ visible, const float* __restrict__ screen_radius,
    UpdateArgs u) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const AdamArgs& a = u.a;
    const uint64_t row0 = (uint64_t)blockIdx.x * (uint32_t)ROWS;
    const uint32_t nrows = (uint32_t)((uint64_t)u.n - row0 < (uint64_t)(uint32_t)ROWS ? (uint64_t)u.n - row0 : (uint64_t)(uint32_t)ROWS);
    const uint32_t row_len = u.sh_len, pitch = row_len + 1;
    float* s_g = s_dyn;                       // [rows][row_len + 1]
    float* s_v = s_dyn + (uint32_t)ROWS * pitch;      // [rows]
    float* s_mask = s_v + (uint32_t)ROWS;             // [rows] 1 = the row's gradient was written, 2 = the splat is dormant (only with u.masked)
    float* s_nz = s_mask + (uint32_t)ROWS;            // [rows] != 0: some moment of the splat is non-zero after this step (only with u.masked)
    float* s_noise = s_nz + (uint32_t)ROWS;           // [rows][3], only with noise_on
    // masked (block-uniform): the gradient tensors were not zero-filled — row r of them counts iff K18 marked the splat: the sign
    // bit of its (non-negative) refine weight, a vector that WAS cleared.
    // No barrier stands in front of what the block fetches: the SH staging below reads the marks it needs straight from global
    // memory (its gradients are wanted last, two round trips hide) and skips the rows nobody wrote — they would come from
    // cold HBM: +7 us at SH degree 3 when loaded and thrown away; the transforms' gradients are loaded unconditionally, their
    // marks come through LDS from the per-splat section (which reads the refine weight anyway), behind its barrier.  (Marks
    // staged through LDS up front, every gradient load predicated on them: +6 us at SH degree 0.)
    const bool masked = u.masked != 0u;
    // DORMANT splats (single-GPU step).  Nine tenths of a scene like the bench's never receive a gradient: every Adam moment of such
    // a splat is zero, its gradient row is not written this step either, the view did not reach it (no noise) — its update is
    // exactly nothing, and fetching its 26 moments, 14 parameters and 14 gradient slots (240 B at SH degree 0, 1.6 KB at degree 3)
    // only to find that out was most of this kernel's traffic.  The fact "all moments of this splat are zero" is kept where the
    // caller's tensors already have a spare bit: the SIGN of the row-reduced SH second moment m2_sh[i] — a sum of squares,
    // never negative — is set (the value is -0.0f) by the step that finds every new moment of the splat zero, and is cleared by the
    // first step that writes a real second moment (any gradient).  -0.0 equals +0.0 in every comparison and in the arithmetic that
    // reads it ((-0) b2 + g f2, sqrt, + eps), so the tensors stay what the reference's are; the mark travels with the caller's
    // data through refine's row gathers and through checkpoints, and a tensor that was zero-filled or loaded from elsewhere simply
    // carries no marks (every splat is processed until it is found dormant again).  A dormant splat costs its eleven per-splat
    // words (44 B: statistics, marks, opacity moments) and nothing else.  Not on the first step (the moment tensors may hold anything).
    const bool dorm_ok = masked && !a.first && u.dormant_skip != 0u;
    const uint32_t* mark_rows = reinterpret_cast<const uint32_t*>(refine_weight) + row0;
    const float rcp_len = 1.0f / (float)row_len;
    const uint32_t sh_count = nrows * row_len;
    const uint64_t sh_base = row0 * row_len;
    // ---- the sections below are separated by barriers (marks and noise travel through LDS, the SH rows need their second moment
    // first) and each fetches its own inputs: a block is a chain of dependent global round trips.  What shortened it (round 4):
    // the per-splat section's eleven loads are unconditional and issued together, the transforms loop has a fixed trip count
    // (unrolled: its loads leave together).  EARLY (S_IT_ > 0, BH_UPDATE_EARLY=1) goes further — every load of the block in front
    // of the first barrier — and loses: see launch_train_update.
    constexpr bool EARLY = S_IT_ > 0;
    constexpr int T_IT = (ROWS * 10 / 4 + OPT_WG - 1) / OPT_WG;          // float4s of the block's transforms per thread
    constexpr int S_IT = S_IT_ > 0 ? S_IT_ : 1;                          // ... of its SH rows (launch_train_update's choice)
    const uint32_t t_count = nrows * 10u, t_vec_end = VEC ? (t_count & ~3u) : 0u;
    const uint64_t t_base = row0 * 10u;
    const uint32_t s_vec_end = VEC ? (sh_count & ~3u) : 0u;
    const bool sh_early = EARLY && VEC && s_vec_end <= (uint32_t)S_IT * OPT_WG * 4u;   // block-uniform (false only under BH_UPDATE_ROWS)
    float4 tg[T_IT], tm1[T_IT], tm2[T_IT], tp[T_IT], sm1[S_IT], sp[S_IT];
    if (EARLY && VEC) {
#pragma unroll
        for (int k = 0; k < T_IT; ++k) {
            const uint32_t e = (threadIdx.x + (uint32_t)k * OPT_WG) * 4u;
            const uint64_t i = t_base + (e < t_vec_end ? e : 0u);   // (clamped: unconditional loads, masked at the use)
            tg[k] = *reinterpret_cast<const float4*>(&g_t[i]);
            tm1[k] = *reinterpret_cast<const float4*>(&m1_t[i]);
            tm2[k] = *reinterpret_cast<const float4*>(&m2_t[i]);
            tp[k] = *reinterpret_cast<const float4*>(&transforms[i]);
        }
#pragma unroll
        for (int k = 0; k < S_IT; ++k) {
            const uint32_t e = (threadIdx.x + (uint32_t)k * OPT_WG) * 4u;
            const uint64_t i = sh_base + ((sh_early && e < s_vec_end) ? e : 0u);
            sm1[k] = *reinterpret_cast<const float4*>(&m1_sh[i]);
            sp[k] = *reinterpret_cast<const float4*>(&sh[i]);
        }
    }
    // this thread's splat (clamped for the threads behind the block's last row)
    const uint64_t si = row0 + (threadIdx.x < nrows ? threadIdx.x : 0u);
    float in_rw = refine_weight[si], in_rn = refine_weight_norm[si], in_vis = visible[si], in_vw = vis_weight[si];
    float in_ms = max_screen_size[si], in_sr = screen_radius[si], in_go = g_o[si], in_m1o = m1_o[si], in_m2o = m2_o[si], in_op = opac[si];
    float in_m2sh = m2_sh[si];
    if (EARLY) asm volatile("" ::: "memory");   // (compiler-only: the loads above are issued HERE, not sunk to their uses behind the staging loop)
    // ---- SH gradients -> LDS (coalesced), its loads queue behind the ones above
    {
        const uint32_t vec_end = VEC ? (sh_count & ~3u) : 0u;
        for (uint32_t e = threadIdx.x * 4u; e < vec_end; e += OPT_WG * 4u) {
            // (a float4 spans at most two rows: the first and the last component's)
            const uint32_t ra = (uint32_t)(((float)e + 0.5f) * rcp_len), rb = (uint32_t)(((float)(e + 3u) + 0.5f) * rcp_len);
            const bool wa = !masked || (mark_rows[ra] >> 31) != 0u, wb = !masked || (mark_rows[rb] >> 31) != 0u;
            float4 g4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (wa || wb) g4 = *reinterpret_cast<const float4*>(&g_sh[sh_base + e]);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t ee = e + k;
                const uint32_t r = (uint32_t)(((float)ee + 0.5f) * rcp_len);  // ee / row_len, exact for ee < 2^16
                s_g[r * pitch + (ee - r * row_len)] = (r == ra ? wa : wb) ? gv[k] * u.gscale : 0.0f;
            }
        }
        for (uint32_t e = vec_end + threadIdx.x; e < sh_count; e += OPT_WG) {
            const uint32_t r = (uint32_t)(((float)e + 0.5f) * rcp_len);
            s_g[r * pitch + (e - r * row_len)] = (!masked || (mark_rows[r] >> 31) != 0u) ? g_sh[sh_base + e] * u.gscale : 0.0f;
        }
    }
    // (one wait for all of it)
    if (EARLY && VEC) {
#pragma unroll
        for (int k = 0; k < T_IT; ++k)
            asm volatile("" : "+v"(tg[k].x), "+v"(tg[k].y), "+v"(tg[k].z), "+v"(tg[k].w), "+v"(tm1[k].x), "+v"(tm1[k].y), "+v"(tm1[k].z), "+v"(tm1[k].w),
                              "+v"(tm2[k].x), "+v"(tm2[k].y), "+v"(tm2[k].z), "+v"(tm2[k].w), "+v"(tp[k].x), "+v"(tp[k].y), "+v"(tp[k].z), "+v"(tp[k].w));
#pragma unroll
        for (int k = 0; k < S_IT; ++k)
            asm volatile("" : "+v"(sm1[k].x), "+v"(sm1[k].y), "+v"(sm1[k].z), "+v"(sm1[k].w), "+v"(sp[k].x), "+v"(sp[k].y), "+v"(sp[k].z), "+v"(sp[k].w));
    }
    if (EARLY) asm volatile("" : "+v"(in_rw), "+v"(in_rn), "+v"(in_vis), "+v"(in_vw), "+v"(in_ms), "+v"(in_sr), "+v"(in_go), "+v"(in_m1o), "+v"(in_m2o), "+v"(in_op), "+v"(in_m2sh));
    // ---- statistics + opacity: one splat per thread
    if (threadIdx.x < nrows) {
        const uint64_t i = row0 + threadIdx.x;
        const float rw_raw = in_rw;
        const bool written = !masked || (f2u(rw_raw) >> 31) != 0u;
        // (the mark is checked against the two moments this thread has loaded anyway: a caller that restored or edited the opacity
        //  moments without rewriting m2_sh cannot make the step skip a splat whose moments are not zero)
        const bool dormant = dorm_ok && !written && f2u(in_m2sh) == 0x80000000u && in_vis == 0.0f && in_m1o == 0.0f && in_m2o == 0.0f;
        if (masked) s_mask[threadIdx.x] = written ? 1.0f : (dormant ? 2.0f : 0.0f);
        // (masked: K18 stored the weight with the sign bit as the mark; an unmarked entry is the zero the forward left)
        {
            const float rn_old = in_rn, vw_old = in_vw, ms_old = in_ms;
            const float rn = __builtin_fmaxf(masked ? __builtin_fabsf(rw_raw) : rw_raw, rn_old);
            const float v = u.vis_clamp ? __builtin_fminf(in_vis, 1.0f) : in_vis;
            const float vw = vw_old + v;
            const float ms = __builtin_fmaxf(in_sr, ms_old);
            if (!same_bits(rn, rn_old)) refine_weight_norm[i] = rn;
            if (!same_bits(vw, vw_old)) vis_weight[i] = vw;
            if (!same_bits(ms, ms_old)) max_screen_size[i] = ms;
        }
        const float g = written ? in_go * u.gscale : 0.0f;
        const float m1_old = in_m1o, m2_old = in_m2o;
        float mm1 = a.first ? g * a.f1 : m1_old * a.beta1 + g * a.f1;
        const float gsq = g * g;
        const float mm2 = a.first ? gsq * a.f2 : m2_old * a.beta2 + gsq * a.f2;
        const float p_old = in_op;
        float p = p_old;
        adam_elem(p, g, mm1, mm2, a, u.lr_opac);
        // (stores of values that did not change are left out, here and below: see the note at the transforms)
        if (a.first || !same_bits(mm1, m1_old)) m1_o[i] = mm1;
        if (a.first || !same_bits(mm2, m2_old)) m2_o[i] = mm2;
        if (!same_bits(p, p_old)) opac[i] = p;
        if (masked) s_nz[threadIdx.x] = (mm1 != 0.0f || mm2 != 0.0f) ? 1.0f : 0.0f;
        if (u.noise_on) {   // the gate reads the UPDATED opacity (train.rs:389)
            const float w = mean_noise_gate(p, in_vis);
            float nz[3] = {0.0f, 0.0f, 0.0f};
            if (w != 0.0f) {
                const float wm = w * u.noise_scale;
                normal3(u.noise_seed, u.noise_step, (uint32_t)i, nz[0], nz[1], nz[2]);
#pragma unroll
                for (int k = 0; k < 3; ++k) nz[k] = clampf(nz[k] * wm, -u.noise_clamp, u.noise_clamp);
            }
            s_noise[threadIdx.x * 3u] = nz[0];
            s_noise[threadIdx.x * 3u + 1u] = nz[1];
            s_noise[threadIdx.x * 3u + 2u] = nz[2];
        }
    }
    if (u.noise_on || masked) __syncthreads();   // block-uniform
    // ---- transforms: full second moment, per-column lr
    {
        const uint32_t count = nrows * 10u;
        const uint64_t base = row0 * 10u;
        auto one = [&](float g_raw, bool written, float m1v, float m2v, float pv, uint32_t e, float& o_m1, float& o_m2, float& o_p) {
            const uint32_t r = (e * 52429u) >> 19;  // e / 10 (e < 2560)
            const uint32_t c = e - r * 10u;
            const float g = written ? g_raw * u.gscale : 0.0f;
            const float mm1 = a.first ? g * a.f1 : m1v * a.beta1 + g * a.f1;
            const float gsq = g * g;
            const float mm2 = a.first ? gsq * a.f2 : m2v * a.beta2 + gsq * a.f2;
            o_m1 = mm1;
            o_m2 = mm2;
            float p = pv;
            float m1c = mm1;
            adam_elem(p, g, m1c, mm2, a, u.tab_t[c] * 1.0f);
            if (u.noise_on && c < 3u) p = p + s_noise[r * 3u + c];
            o_p = p;
        };
        const uint32_t vec_end = t_vec_end;
#pragma unroll
        for (int k = 0; k < T_IT; ++k) {
            const uint32_t e = (threadIdx.x + (uint32_t)k * OPT_WG) * 4u;
            if (e >= vec_end) break;
            const uint64_t i = base + e;
            const uint32_t ra = (e * 52429u) >> 19, rb = ((e + 3u) * 52429u) >> 19;   // the float4's first and last row
            const float ka = masked ? s_mask[ra] : 1.0f, kb = masked ? s_mask[rb] : 1.0f;
            if (ka == 2.0f && kb == 2.0f) continue;   // both rows dormant: nothing to fetch, nothing moves
            const float4 g4 = EARLY ? tg[k] : *reinterpret_cast<const float4*>(&g_t[i]);
            float4 m14 = EARLY ? tm1[k] : *reinterpret_cast<const float4*>(&m1_t[i]);
            float4 m24 = EARLY ? tm2[k] : *reinterpret_cast<const float4*>(&m2_t[i]);
            float4 p4 = EARLY ? tp[k] : *reinterpret_cast<const float4*>(&transforms[i]);
            const bool wa = ka == 1.0f, wb = kb == 1.0f;
            const float4 m1_old = m14, m2_old = m24, p_old = p4;
            one(g4.x, wa, m14.x, m24.x, p4.x, e, m14.x, m24.x, p4.x);
            one(g4.y, (((e + 1u) * 52429u) >> 19) == ra ? wa : wb, m14.y, m24.y, p4.y, e + 1, m14.y, m24.y, p4.y);
            one(g4.z, (((e + 2u) * 52429u) >> 19) == ra ? wa : wb, m14.z, m24.z, p4.z, e + 2, m14.z, m24.z, p4.z);
            one(g4.w, wb, m14.w, m24.w, p4.w, e + 3, m14.w, m24.w, p4.w);
            if (masked) {   // which rows still carry a non-zero moment (the dormant mark is set from this at the end)
                const uint32_t r1 = ((e + 1u) * 52429u) >> 19, r2 = ((e + 2u) * 52429u) >> 19;
                if (m14.x != 0.0f || m24.x != 0.0f) s_nz[ra] = 1.0f;
                if (m14.y != 0.0f || m24.y != 0.0f) s_nz[r1] = 1.0f;
                if (m14.z != 0.0f || m24.z != 0.0f) s_nz[r2] = 1.0f;
                if (m14.w != 0.0f || m24.w != 0.0f) s_nz[rb] = 1.0f;
            }
            // A store whose four values are bit for bit what was loaded is left out.  That is the case for every splat that has
            // never received a gradient (moments 0, gradient 0: the moments stay 0 and the parameter does not move) — nine tenths
            // of the bench scene, where most splats lie behind saturated tiles in every view; a scene whose splats all reach a
            // pixel now and then writes everything, as before.  The update is HBM-bound and 3 of its 7 streams are stores.
            if (a.first || !same_bits4(m14, m1_old)) *reinterpret_cast<float4*>(&m1_t[i]) = m14;
            if (a.first || !same_bits4(m24, m2_old)) *reinterpret_cast<float4*>(&m2_t[i]) = m24;
            if (!same_bits4(p4, p_old)) *reinterpret_cast<float4*>(&transforms[i]) = p4;
        }
        for (uint32_t e = vec_end + threadIdx.x; e < count; e += OPT_WG) {
            const uint64_t i = base + e;
            float o1, o2, op;
            const uint32_t r = (e * 52429u) >> 19;
            one(g_t[i], !masked || s_mask[r] == 1.0f, m1_t[i], m2_t[i], transforms[i], e, o1, o2, op);
            m1_t[i] = o1;
            m2_t[i] = o2;
            transforms[i] = op;
            if (masked && (o1 != 0.0f || o2 != 0.0f)) s_nz[r] = 1.0f;
        }
    }
    // ---- SH: per-row second moment (adam_scaled.rs:99-104,152-165), row sums in index order
    __syncthreads();
    if (threadIdx.x < nrows) {
        const float* g = s_g + threadIdx.x * pitch;
        float acc = 0.0f;
        for (uint32_t c = 0; c < row_len; ++c) acc += g[c] * g[c];
        const float row_gsq = acc / (float)row_len;
        const uint64_t r = row0 + threadIdx.x;
        const float v_old = in_m2sh;
        const float v = a.first ? row_gsq * a.f2 : v_old * a.beta2 + row_gsq * a.f2;
        const bool is_dormant = masked && s_mask[threadIdx.x] == 2.0f;
        // (a dormant row keeps its -0.0: the recurrence would turn it into +0.0 and un-mark it every step)
        if (!is_dormant && (a.first || !same_bits(v, v_old))) m2_sh[r] = v;
        s_v[threadIdx.x] = v;
        if (masked && v != 0.0f) s_nz[threadIdx.x] = 1.0f;
    }
    __syncthreads();
    {
        auto one = [&](float m1v, float pv, uint32_t e, float& o_m1, float& o_p) {
            const uint32_t r = (uint32_t)(((float)e + 0.5f) * rcp_len);
            const uint32_t c = e - r * row_len;
            const float gi = s_g[r * pitch + c];
            float mm1 = a.first ? gi * a.f1 : m1v * a.beta1 + gi * a.f1;
            o_m1 = mm1;
            float p = pv;
            adam_elem(p, gi, mm1, s_v[r], a, u.tab_sh[c] * u.lr_sh);
            o_p = p;
        };
        const uint32_t vec_end = s_vec_end;
        auto row_of = [&](uint32_t e) { return (uint32_t)(((float)e + 0.5f) * rcp_len); };
        auto four = [&](float4 m14, float4 p4, uint32_t e) {
            const uint64_t i = sh_base + e;
            const float4 m1_old = m14, p_old = p4;
            one(m14.x, p4.x, e, m14.x, p4.x);
            one(m14.y, p4.y, e + 1, m14.y, p4.y);
            one(m14.z, p4.z, e + 2, m14.z, p4.z);
            one(m14.w, p4.w, e + 3, m14.w, p4.w);
            if (masked) {
                if (m14.x != 0.0f) s_nz[row_of(e)] = 1.0f;
                if (m14.y != 0.0f) s_nz[row_of(e + 1u)] = 1.0f;
                if (m14.z != 0.0f) s_nz[row_of(e + 2u)] = 1.0f;
                if (m14.w != 0.0f) s_nz[row_of(e + 3u)] = 1.0f;
            }
            if (a.first || !same_bits4(m14, m1_old)) *reinterpret_cast<float4*>(&m1_sh[i]) = m14;
            if (!same_bits4(p4, p_old)) *reinterpret_cast<float4*>(&sh[i]) = p4;
        };
        if (sh_early) {
#pragma unroll
            for (int k = 0; k < S_IT; ++k) {
                const uint32_t e = (threadIdx.x + (uint32_t)k * OPT_WG) * 4u;
                if (e >= vec_end) break;
                four(sm1[k], sp[k], e);
            }
        } else {
            for (uint32_t e = threadIdx.x * 4u; e < vec_end; e += OPT_WG * 4u) {
                // (a float4 spans at most two rows for row_len >= 3: the first and the last component's)
                if (masked && s_mask[row_of(e)] == 2.0f && s_mask[row_of(e + 3u)] == 2.0f) continue;   // dormant rows: nothing to fetch
                four(*reinterpret_cast<const float4*>(&m1_sh[sh_base + e]), *reinterpret_cast<const float4*>(&sh[sh_base + e]), e);
            }
        }
        for (uint32_t e = vec_end + threadIdx.x; e < sh_count; e += OPT_WG) {
            const uint64_t i = sh_base + e;
            float o1, op;
            one(m1_sh[i], sh[i], e, o1, op);
            m1_sh[i] = o1;
            sh[i] = op;
            if (masked && o1 != 0.0f) s_nz[row_of(e)] = 1.0f;
        }
    }
    // ---- a splat whose moments are ALL zero after this step is dormant from now on: the mark is the sign of its m2_sh (== -0.0f)
    if (masked) {   // block-uniform
        __syncthreads();
        if (threadIdx.x < nrows && s_mask[threadIdx.x] != 2.0f && s_nz[threadIdx.x] == 0.0f) m2_sh[row0 + threadIdx.x] = -0.0f;
    }
}

int launch_train_update(bh_ctx* ctx, const BhTrainState* st, const float* g_t, const float* g_sh, const float* g_o,
                        const float* refine_weight, const float* visible, const float* screen_radius, float gscale,
                        bool vis_clamp, const float* tab_t, float lr_sh, float sh_rest_scale, float lr_opac, uint32_t t,
                        float beta1, float beta2, float eps, const NoiseArgs* noise, bool masked_rows) {
    const uint32_t n = st->n, C = (st->sh_degree + 1) * (st->sh_degree + 1);
    if (n == 0) return 0;
    if (t == 0) return set_error(ctx, BH_ERR_INVALID_ARG, "adam: t is 1-based");
    UpdateArgs u;
    u.a.beta1 = beta1; u.a.beta2 = beta2;
    u.a.f1 = 1.0f - beta1; u.a.f2 = 1.0f - beta2;
    u.a.bc1 = 1.0f - powi_f32(beta1, (int)t);
    u.a.bc2 = 1.0f - powi_f32(beta2, (int)t);
    u.a.eps = eps; u.a.lr = 1.0f;
    u.a.first = t == 1 ? 1u : 0u;
    u.lr_sh = lr_sh; u.lr_opac = lr_opac; u.gscale = gscale;
    u.n = n; u.sh_len = 3 * C; u.vis_clamp = vis_clamp ? 1u : 0u;
    u.masked = masked_rows ? 1u : 0u;
    // Marks are trusted only on a state this context updated at the previous step (same tensors, consecutive step count): a state
    // seen for the first time, re-bound to other tensors (refine, a checkpoint restore) or with a step counter that jumped is
    // processed in full once — that step re-derives every mark from the moments it finds.
    const bool same_state = ctx->marks_m2_sh == st->m2_sh && ctx->marks_m1_t == st->m1_transforms && ctx->marks_n == n && ctx->marks_step + 1u == t;
    ctx->marks_m2_sh = st->m2_sh; ctx->marks_m1_t = st->m1_transforms; ctx->marks_n = n; ctx->marks_step = t;
    u.dormant_skip = (ctx->knob_no_dormant || !same_state) ? 0u : 1u;
    u.noise_on = noise ? 1u : 0u;
    u.noise_step = noise ? noise->step : 0u;
    u.noise_scale = noise ? noise->scale : 0.0f;
    u.noise_clamp = noise ? noise->clamp_abs : 0.0f;
    u.noise_seed = noise ? noise->seed : 0ull;
    for (int i = 0; i < 10; ++i) u.tab_t[i] = tab_t[i];
    for (uint32_t k = 0; k < 75; ++k) u.tab_sh[k] = (k / 3 == 0) ? 1.0f : sh_rest_scale;
    // splats per block: ~13 KB of LDS-staged SH gradients keeps >= 8 blocks resident per CU (measured at 1 M splats:
    // SH degree 3: 0.368 ms @256, 0.300 @128, 0.290 @64, 0.327 @32; degree 0 is best at 256)
    uint32_t rows = u.sh_len <= 12 ? 256u : (u.sh_len <= 27 ? 128u : 64u);
    if (ctx->knob_update_rows) rows = ctx->knob_update_rows;  // developer knob BH_UPDATE_ROWS (read once at bh_create)
    const size_t lds = ((size_t)rows * (u.sh_len + 1) + 3 * rows + (noise ? 3 * rows : 0)) * sizeof(float);
    const unsigned nb = (unsigned)(((uint64_t)n + rows - 1) / rows);
    const void* vec_ptrs[] = {st->transforms, st->m1_transforms, st->m2_transforms, g_t, st->sh_coeffs, st->m1_sh, g_sh};
    bool vec = true;
    for (const void* q : vec_ptrs) vec = vec && ((uintptr_t)q & 15u) == 0;
#define BH_LAUNCH_UPDATE(V, R, S)                                                                                                       \
    hipLaunchKernelGGL((train_update_kernel<V, R, S>), dim3(nb), dim3(OPT_WG), lds, ctx->stream, st->transforms, st->m1_transforms,      \
                       st->m2_transforms, g_t, st->sh_coeffs, st->m1_sh, st->m2_sh, g_sh, st->raw_opacities, st->m1_opac, st->m2_opac, \
                       g_o, st->refine_weight_norm, st->vis_weight, st->max_screen_size, refine_weight, visible, screen_radius, u)
    // float4s of a block's SH rows per thread: 1 (<= 1024 floats), 4 (128 x 27), 5 (64 x 75); 0 = no up-front loads
    const uint32_t sh_f4 = (rows * u.sh_len / 4u + OPT_WG - 1) / OPT_WG;
    // (off by default: measured again with the store skipping and the fixed-trip loops in place, the sections' own loads win at
    //  every degree — degree 3: 158 vs 209 us on the bench scene, 194 vs 233 us with every store forced — the 57 instead of 99
    //  VGPRs are worth more than the shorter chain; BH_UPDATE_EARLY=1 selects the up-front loads)
    const bool early = vec && rows != 256u && ctx->knob_update_early;
    if (rows == 256u) { if (vec) BH_LAUNCH_UPDATE(true, 256, 0); else BH_LAUNCH_UPDATE(false, 256, 0); }
    else if (rows == 128u) {
        if (!vec) BH_LAUNCH_UPDATE(false, 128, 0);
        else if (!early) BH_LAUNCH_UPDATE(true, 128, 0);
        else if (sh_f4 <= 1u) BH_LAUNCH_UPDATE(true, 128, 1);
        else BH_LAUNCH_UPDATE(true, 128, 4);
    } else {
        if (!vec) BH_LAUNCH_UPDATE(false, 64, 0);
        else if (!early) BH_LAUNCH_UPDATE(true, 64, 0);
        else if (sh_f4 <= 1u) BH_LAUNCH_UPDATE(true, 64, 1);
        else BH_LAUNCH_UPDATE(true, 64, 5);
    }
#undef BH_LAUNCH_UPDATE
    BH_LAUNCH_CHECK(ctx, "train_update_kernel");
    return 0;
}

// stats.rs:40-50
__global__ __launch_bounds__(OPT_WG) void gather_stats_kernel(float* __restrict__ refine_weight_norm, float* __restrict__ vis_weight,
                                                             float* __restrict__ max_screen_size, const float* __restrict__ refine_weight,
                                                             const float* __restrict__ visible, const float* __restrict__ screen_radius,
                                                             uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * OPT_WG + threadIdx.x;
    if (i >= n) return;
    refine_weight_norm[i] = __builtin_fmaxf(refine_weight[i], refine_weight_norm[i]);
    vis_weight[i] = vis_weight[i] + visible[i];
    max_screen_size[i] = __builtin_fmaxf(screen_radius[i], max_screen_size[i]);
}

int launch_gather_stats(bh_ctx* ctx, float* refine_weight_norm, float* vis_weight, float* max_screen_size,
                        const float* refine_weight, const float* visible, const float* screen_radius, uint64_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(gather_stats_kernel, dim3((unsigned)((n + OPT_WG - 1) / OPT_WG)), dim3(OPT_WG), 0, ctx->stream, refine_weight_norm, vis_weight, max_screen_size, refine_weight, visible, screen_radius, n);
    BH_LAUNCH_CHECK(ctx, "gather_stats_kernel");
    return 0;
}

// train.rs:389-416: means += clamp(sample * (1 - sigmoid(raw_opac))^150 * visible * scale, +-clamp_abs)
// samples == NULL: the N(0,1) samples are drawn on the spot (device_rng.h, keyed by seed / step / splat).
__global__ __launch_bounds__(OPT_WG) void mean_noise_kernel(float* __restrict__ transforms, const float* __restrict__ raw_opac,
                                                           const float* __restrict__ visible, const float* __restrict__ samples,
                                                           uint64_t n, float noise_scale, float clamp_abs, uint64_t seed, uint32_t step) {
    const uint64_t i = (uint64_t)blockIdx.x * OPT_WG + threadIdx.x;
    if (i >= n) return;
    const float w = mean_noise_gate(raw_opac[i], visible[i]);
    const float wm = w * noise_scale;
    float smp[3];
    if (samples) {
        smp[0] = samples[i * 3]; smp[1] = samples[i * 3 + 1]; smp[2] = samples[i * 3 + 2];
    } else {
        if (w == 0.0f) return;   // sample * 0 = 0 for every finite sample: the means do not move
        normal3(seed, step, (uint32_t)i, smp[0], smp[1], smp[2]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float nz = clampf(smp[k] * wm, -clamp_abs, clamp_abs);
        transforms[i * 10 + k] = transforms[i * 10 + k] + nz;
    }
}

int launch_mean_noise(bh_ctx* ctx, float* transforms, const float* raw_opac, const float* visible, const float* samples,
                      uint64_t n, float noise_scale, float clamp_abs, uint64_t seed, uint32_t step) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(mean_noise_kernel, dim3((unsigned)((n + OPT_WG - 1) / OPT_WG)), dim3(OPT_WG), 0, ctx->stream, transforms, raw_opac, visible, samples, n, noise_scale, clamp_abs, seed, step);
    BH_LAUNCH_CHECK(ctx, "mean_noise_kernel");
    return 0;
}

// the raw N(0,1) samples a step would draw: out [n,3] (tests; callers that want the reference's `samples` tensor)
__global__ __launch_bounds__(OPT_WG) void normal_samples_kernel(float* __restrict__ out, uint64_t n, uint64_t seed, uint32_t step) {
    const uint64_t i = (uint64_t)blockIdx.x * OPT_WG + threadIdx.x;
    if (i >= n) return;
    float a, b, c;
    normal3(seed, step, (uint32_t)i, a, b, c);
    out[i * 3] = a; out[i * 3 + 1] = b; out[i * 3 + 2] = c;
}

int launch_normal_samples(bh_ctx* ctx, float* out, uint64_t n, uint64_t seed, uint32_t step) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(normal_samples_kernel, dim3((unsigned)((n + OPT_WG - 1) / OPT_WG)), dim3(OPT_WG), 0, ctx->stream, out, n, seed, step);
    BH_LAUNCH_CHECK(ctx, "normal_samples_kernel");
    return 0;
}

}  // namespace bh
