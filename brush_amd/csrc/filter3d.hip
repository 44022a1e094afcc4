// filter3d.hip — Mip-Splatting 3D smoothing filter: the per-splat world-space scale floor.
//
// Reference: brush-render/src/gaussian_splats.rs:86-111 (fold_min_scale, a chain of burn
// elementwise tensor ops differentiated by burn's autodiff) and brush-train/src/train.rs:102-125
// (compute_min_scale).  Paths under /root/reference/crates.
//
// MI355X shape: one pass each.  The fold reads a [N,10] row + 2 scalars and writes the folded
// row + 1 scalar (88 B/splat: pure HBM streaming, rows staged through LDS so both the 40-byte
// row reads and writes are coalesced float4 bursts); its VJP is written out by hand (the
// reference gets it from the autodiff tape: ~25 elementwise kernels and as many [N,3]
// temporaries) and rewrites the three log-scale gradient columns and the opacity gradient in
// place.  The floor itself is one pass over the means with the K view centres in kernel
// arguments' constant memory (the reference loops K times over [N,3] tensors).
#include "context.h"

namespace bh {

constexpr int F3_WG = 256;

struct FoldTerms {
    float new_log[3];  // 0.5 ln(s^2 + f^2)
    float w[3];        // s^2 / (s^2 + f^2) = d new_log / d log_s
    float coef;        // sqrt(prod s^2 / prod (s^2 + f^2))
};

BH_DEV FoldTerms fold_terms(float l0, float l1, float l2, float f) {
    FoldTerms t;
    const float f2 = f * f;
    const float a0 = bh_expf(l0 * 2.0f), a1 = bh_expf(l1 * 2.0f), a2 = bh_expf(l2 * 2.0f);
    const float b0 = a0 + f2, b1 = a1 + f2, b2 = a2 + f2;
    t.new_log[0] = bh_logf(b0) * 0.5f;
    t.new_log[1] = bh_logf(b1) * 0.5f;
    t.new_log[2] = bh_logf(b2) * 0.5f;
    t.w[0] = a0 / b0;
    t.w[1] = a1 / b1;
    t.w[2] = a2 / b2;
    const float det1 = a0 * a1 * a2;
    const float det2 = b0 * b1 * b2;
    t.coef = __builtin_sqrtf(det1 / det2);
    return t;
}

// gaussian_splats.rs:86-111
__global__ __launch_bounds__(F3_WG) void fold_min_scale_kernel(const float* transforms, const float* raw_opac,  /* out may alias in (bake) */
                                                               const float* __restrict__ min_scale, uint32_t n,
                                                               float* out_transforms, float* out_raw_opac) {
    __shared__ float s_rows[F3_WG * 10];
    const uint32_t base = blockIdx.x * F3_WG;
    const uint32_t rows = min((uint32_t)F3_WG, n - base);
    // coalesced load of this block's rows (40 B rows -> consecutive dwords)
    for (uint32_t e = threadIdx.x; e < rows * 10; e += F3_WG) s_rows[e] = transforms[(size_t)base * 10 + e];
    __syncthreads();
    const uint32_t i = base + threadIdx.x;
    if (threadIdx.x < rows) {
        float* r = s_rows + threadIdx.x * 10;
        const FoldTerms t = fold_terms(r[7], r[8], r[9], min_scale[i]);
        r[7] = t.new_log[0];
        r[8] = t.new_log[1];
        r[9] = t.new_log[2];
        const float opac = clampf(sigmoid(raw_opac[i]) * t.coef, 1e-6f, 1.0f - 1e-6f);
        out_raw_opac[i] = bh_logf(opac / (-opac + 1.0f));
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < rows * 10; e += F3_WG) out_transforms[(size_t)base * 10 + e] = s_rows[e];
}

// VJP of the fold (f constant).  v_transforms[:,7:10] and v_raw_opac: in = gradient w.r.t. the folded
// tensors, out = w.r.t. the raw parameters.
__global__ __launch_bounds__(F3_WG) void fold_min_scale_backward_kernel(const float* __restrict__ transforms, const float* __restrict__ raw_opac,
                                                                        const float* __restrict__ min_scale, uint32_t n,
                                                                        float* __restrict__ v_transforms, float* __restrict__ v_raw_opac) {
    const uint32_t i = blockIdx.x * F3_WG + threadIdx.x;
    if (i >= n) return;
    const float* tr = transforms + (size_t)i * 10;
    const FoldTerms t = fold_terms(tr[7], tr[8], tr[9], min_scale[i]);
    const float sg = sigmoid(raw_opac[i]);
    const float pre = sg * t.coef;
    const bool inside = pre >= 1e-6f && pre <= 1.0f - 1e-6f;
    const float opac = clampf(pre, 1e-6f, 1.0f - 1e-6f);
    const float v_pre = inside ? v_raw_opac[i] / (opac * (1.0f - opac)) : 0.0f;
    v_raw_opac[i] = v_pre * t.coef * sg * (1.0f - sg);
    const float v_coef_coef = v_pre * sg * t.coef;
    float* vt = v_transforms + (size_t)i * 10;
#pragma unroll
    for (int k = 0; k < 3; ++k) vt[7 + k] = vt[7 + k] * t.w[k] + v_coef_coef * (1.0f - t.w[k]);
}

// train.rs:102-125
constexpr int MAX_VIEWS_PER_LAUNCH = 64;
struct ViewCams { float c[MAX_VIEWS_PER_LAUNCH][4]; };  // centre xyz, max(focal, 1e-6)

__global__ __launch_bounds__(F3_WG) void compute_min_scale_kernel(const float* __restrict__ transforms, uint32_t n, ViewCams cams, uint32_t k,
                                                                  float sqrt_factor, int first, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * F3_WG + threadIdx.x;
    if (i >= n) return;
    const float* m = transforms + (size_t)i * 10;
    const float mx = m[0], my = m[1], mz = m[2];
    // later chunks of a long view list continue from the running (unscaled) minimum
    float best = first ? 0.0f : out[i];
    for (uint32_t v = 0; v < k; ++v) {
        const float dx = mx - cams.c[v][0], dy = my - cams.c[v][1], dz = mz - cams.c[v][2];
        const float dist = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
        const float ratio = dist / cams.c[v][3];
        best = (first && v == 0) ? ratio : __builtin_fminf(best, ratio);
    }
    out[i] = best;
}

__global__ __launch_bounds__(F3_WG) void scale_kernel(float* __restrict__ x, uint32_t n, float s) {
    const uint32_t i = blockIdx.x * F3_WG + threadIdx.x;
    if (i < n) x[i] = x[i] * s;
}

int launch_fold_min_scale(bh_ctx* ctx, const float* transforms, const float* raw_opac, const float* min_scale, uint32_t n,
                          float* out_transforms, float* out_raw_opac) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fold_min_scale_kernel, dim3((n + F3_WG - 1) / F3_WG), dim3(F3_WG), 0, ctx->stream, transforms, raw_opac, min_scale, n,
                       out_transforms, out_raw_opac);
    BH_LAUNCH_CHECK(ctx, "fold_min_scale_kernel");
    return 0;
}

int launch_fold_min_scale_backward(bh_ctx* ctx, const float* transforms, const float* raw_opac, const float* min_scale, uint32_t n,
                                   float* v_transforms, float* v_raw_opac) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fold_min_scale_backward_kernel, dim3((n + F3_WG - 1) / F3_WG), dim3(F3_WG), 0, ctx->stream, transforms, raw_opac,
                       min_scale, n, v_transforms, v_raw_opac);
    BH_LAUNCH_CHECK(ctx, "fold_min_scale_backward_kernel");
    return 0;
}

int launch_compute_min_scale(bh_ctx* ctx, const float* transforms, uint32_t n, const float* view_cams, uint32_t k, float factor, float* out) {
    if (n == 0) return 0;
    const dim3 grid((n + F3_WG - 1) / F3_WG), block(F3_WG);
    for (uint32_t v0 = 0; v0 < k; v0 += MAX_VIEWS_PER_LAUNCH) {
        ViewCams cams;
        const uint32_t cnt = k - v0 < (uint32_t)MAX_VIEWS_PER_LAUNCH ? k - v0 : (uint32_t)MAX_VIEWS_PER_LAUNCH;
        for (uint32_t v = 0; v < cnt; ++v) {
            const float* c = view_cams + (size_t)(v0 + v) * 4;
            cams.c[v][0] = c[0]; cams.c[v][1] = c[1]; cams.c[v][2] = c[2];
            cams.c[v][3] = c[3] > 1e-6f ? c[3] : 1e-6f;
        }
        hipLaunchKernelGGL(compute_min_scale_kernel, grid, block, 0, ctx->stream, transforms, n, cams, cnt, 0.0f, v0 == 0 ? 1 : 0, out);
        BH_LAUNCH_CHECK(ctx, "compute_min_scale_kernel");
    }
    hipLaunchKernelGGL(scale_kernel, grid, block, 0, ctx->stream, out, n, std::sqrt(factor));
    BH_LAUNCH_CHECK(ctx, "scale_kernel");
    return 0;
}

}  // namespace bh
