// device_math.h — gfx950 device-side arithmetic for the splat kernels.
//
// Arithmetic contract (must be compiled with -ffp-contract=off):
//   * + - * / sqrt: IEEE binary32, correctly rounded, in the operation order of
//     the reference's #[cube] math (brush-cube/src/lib.rs); no implicit FMA.
//   * exp / ln: fixed polynomials bh_expf / bh_logf (≈1 ulp).  The reference
//     inherits whatever the WGSL compiler gives; fixing the polynomial makes the
//     integer outputs of the pipeline (tile assignment, sort order) reproducible
//     bit-for-bit on any device that implements the same sequence.
//   * calc_sigma: two explicit fma (the only deliberate contraction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BH_DEV __device__ __forceinline__

namespace bh {

constexpr uint32_t TILE_WIDTH = 16;   // brush-render/src/kernels/helpers.rs:15
constexpr uint32_t TILE_SIZE = 256;
constexpr float ALPHA_CUTOFF_MID = 1.0f / 255.0f;  // helpers.rs:23
constexpr float ALPHA_CUTOFF_BAND = 1.0e-3f;       // helpers.rs:24

BH_DEV uint32_t f2u(float x) { return __float_as_uint(x); }
BH_DEV float u2f(uint32_t u) { return __uint_as_float(u); }
// brush-cube/src/lib.rs:566-570
BH_DEV bool is_finite_f32(float x) { return ((f2u(x) >> 23) & 0xFFu) != 0xFFu; }

BH_DEV float bh_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return __builtin_inff();
    if (x < -103.9f) return 0.0f;
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

BH_DEV float bh_logf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (!is_finite_f32(x)) return x;
    uint32_t bits = f2u(x);
    int e_adj = 0;
    if (((bits >> 23) & 0xFFu) == 0u) {
        x = x * 8388608.0f;
        bits = f2u(x);
        e_adj = -23;
    }
    int e = (int)((bits >> 23) & 0xFFu) - 126 + e_adj;
    float m = u2f((bits & 0x807FFFFFu) | 0x3F000000u);
    if (m < 0.70710678118654752440f) {
        e = e - 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = __builtin_fmaf(y, m, -1.1514610310e-1f);
    y = __builtin_fmaf(y, m, 1.1676998740e-1f);
    y = __builtin_fmaf(y, m, -1.2420140846e-1f);
    y = __builtin_fmaf(y, m, 1.4249322787e-1f);
    y = __builtin_fmaf(y, m, -1.6668057665e-1f);
    y = __builtin_fmaf(y, m, 2.0000714765e-1f);
    y = __builtin_fmaf(y, m, -2.4999993993e-1f);
    y = __builtin_fmaf(y, m, 3.3333331174e-1f);
    y = y * m * z;
    const float fe = (float)e;
    y = __builtin_fmaf(-2.12194440e-4f, fe, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(0.693359375f, fe, r);
    return r;
}

// brush-cube/src/lib.rs:560-563
BH_DEV float sigmoid(float x) { return 1.0f / (1.0f + bh_expf(-x)); }
BH_DEV float clampf(float x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }

// ---- aggregates (brush-cube/src/lib.rs:39-538) --------------------------------
struct Vec3A { float x, y, z; };
struct Vec2 { float x, y; };
struct Quat { float w, x, y, z; };
struct Mat3 { float c0x, c0y, c0z, c1x, c1y, c1z, c2x, c2y, c2z; };
struct Mat2x3 { Vec2 c0, c1, c2; };
struct Sym2 { float c00, c01, c11; };
struct Sym3 { float c00, c01, c02, c11, c12, c22; };

BH_DEV Vec3A v3(float x, float y, float z) { return Vec3A{x, y, z}; }
BH_DEV Vec3A add(Vec3A a, Vec3A b) { return Vec3A{a.x + b.x, a.y + b.y, a.z + b.z}; }
BH_DEV Vec3A sub(Vec3A a, Vec3A b) { return Vec3A{a.x - b.x, a.y - b.y, a.z - b.z}; }
BH_DEV Vec3A scale(Vec3A a, float s) { return Vec3A{a.x * s, a.y * s, a.z * s}; }
BH_DEV float dot(Vec3A a, Vec3A b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + 0.0f; }
BH_DEV float length(Vec3A a) { return __builtin_sqrtf(dot(a, a)); }
BH_DEV Vec3A normalize(Vec3A a) { return scale(a, 1.0f / length(a)); }
BH_DEV bool finite3(Vec3A a) { return is_finite_f32(a.x) && is_finite_f32(a.y) && is_finite_f32(a.z); }
BH_DEV Vec2 add(Vec2 a, Vec2 b) { return Vec2{a.x + b.x, a.y + b.y}; }
BH_DEV Vec2 scale(Vec2 a, float s) { return Vec2{a.x * s, a.y * s}; }
BH_DEV float dot(Vec2 a, Vec2 b) { return a.x * b.x + a.y * b.y; }
BH_DEV float qdot(Quat a, Quat b) { return ((a.w * b.w + a.x * b.x) + a.y * b.y) + a.z * b.z; }
BH_DEV Quat qscale(Quat q, float s) { return Quat{q.w * s, q.x * s, q.y * s, q.z * s}; }
BH_DEV Quat qnormalize(Quat q) { return qscale(q, 1.0f / __builtin_sqrtf(qdot(q, q))); }

BH_DEV Mat3 quat_to_mat3(Quat q) {
    const float w = q.w, qx = q.x, qy = q.y, qz = q.z;
    const float x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
    const float xy = qx * qy, xz = qx * qz, yz = qy * qz;
    const float wx = w * qx, wy = w * qy, wz = w * qz;
    Mat3 m;
    m.c0x = 1.0f - 2.0f * (y2 + z2);
    m.c0y = 2.0f * (xy + wz);
    m.c0z = 2.0f * (xz - wy);
    m.c1x = 2.0f * (xy - wz);
    m.c1y = 1.0f - 2.0f * (x2 + z2);
    m.c1z = 2.0f * (yz + wx);
    m.c2x = 2.0f * (xz + wy);
    m.c2y = 2.0f * (yz - wx);
    m.c2z = 1.0f - 2.0f * (x2 + y2);
    return m;
}
BH_DEV Vec3A col0(const Mat3& m) { return Vec3A{m.c0x, m.c0y, m.c0z}; }
BH_DEV Vec3A col1(const Mat3& m) { return Vec3A{m.c1x, m.c1y, m.c1z}; }
BH_DEV Vec3A col2(const Mat3& m) { return Vec3A{m.c2x, m.c2y, m.c2z}; }
BH_DEV Vec3A row0(const Mat3& m) { return Vec3A{m.c0x, m.c1x, m.c2x}; }
BH_DEV Vec3A row1(const Mat3& m) { return Vec3A{m.c0y, m.c1y, m.c2y}; }
BH_DEV Vec3A row2(const Mat3& m) { return Vec3A{m.c0z, m.c1z, m.c2z}; }
BH_DEV Mat3 from_cols(Vec3A a, Vec3A b, Vec3A c) { return Mat3{a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z}; }
BH_DEV Vec3A mul_vec3(const Mat3& m, Vec3A v) { return add(add(scale(col0(m), v.x), scale(col1(m), v.y)), scale(col2(m), v.z)); }
BH_DEV Vec3A transpose_mul_vec3(const Mat3& m, Vec3A v) { return Vec3A{dot(col0(m), v), dot(col1(m), v), dot(col2(m), v)}; }
BH_DEV Mat3 mul_mat3(const Mat3& m, const Mat3& n) { return from_cols(mul_vec3(m, col0(n)), mul_vec3(m, col1(n)), mul_vec3(m, col2(n))); }
BH_DEV Mat3 mul_diag(const Mat3& m, Vec3A s) { return from_cols(scale(col0(m), s.x), scale(col1(m), s.y), scale(col2(m), s.z)); }
BH_DEV Sym3 outer_product_self(const Mat3& m) {
    const Vec3A r0 = row0(m), r1 = row1(m), r2 = row2(m);
    return Sym3{dot(r0, r0), dot(r0, r1), dot(r0, r2), dot(r1, r1), dot(r1, r2), dot(r2, r2)};
}
BH_DEV Vec2 mul_vec3(const Mat2x3& m, Vec3A v) { return add(add(scale(m.c0, v.x), scale(m.c1, v.y)), scale(m.c2, v.z)); }
BH_DEV Mat2x3 mul_mat3(const Mat2x3& m, const Mat3& n) { return Mat2x3{mul_vec3(m, col0(n)), mul_vec3(m, col1(n)), mul_vec3(m, col2(n))}; }
BH_DEV Vec3A row0(const Mat2x3& m) { return Vec3A{m.c0.x, m.c1.x, m.c2.x}; }
BH_DEV Vec3A row1(const Mat2x3& m) { return Vec3A{m.c0.y, m.c1.y, m.c2.y}; }
BH_DEV Sym2 gram_matrix(const Mat2x3& m) {
    Sym2 s;
    s.c00 = m.c0.x * m.c0.x + m.c1.x * m.c1.x + m.c2.x * m.c2.x;
    s.c01 = m.c0.x * m.c0.y + m.c1.x * m.c1.y + m.c2.x * m.c2.y;
    s.c11 = m.c0.y * m.c0.y + m.c1.y * m.c1.y + m.c2.y * m.c2.y;
    return s;
}
BH_DEV Vec2 sym2_mul_vec2(Sym2 s, Vec2 v) { return add(scale(Vec2{s.c00, s.c01}, v.x), scale(Vec2{s.c01, s.c11}, v.y)); }
BH_DEV Mat2x3 sym2_mul_mat2x3(Sym2 s, const Mat2x3& n) { return Mat2x3{sym2_mul_vec2(s, n.c0), sym2_mul_vec2(s, n.c1), sym2_mul_vec2(s, n.c2)}; }
BH_DEV Sym3 transpose_congruence_sym2(const Mat2x3& m, Sym2 sym) {
    const Vec2 sc0 = sym2_mul_vec2(sym, m.c0), sc1 = sym2_mul_vec2(sym, m.c1), sc2 = sym2_mul_vec2(sym, m.c2);
    return Sym3{dot(m.c0, sc0), dot(m.c0, sc1), dot(m.c0, sc2), dot(m.c1, sc1), dot(m.c1, sc2), dot(m.c2, sc2)};
}
BH_DEV Sym2 sym2_scale(Sym2 s, float k) { return Sym2{s.c00 * k, s.c01 * k, s.c11 * k}; }
BH_DEV float sym2_max_abs(Sym2 s) { return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(s.c00), __builtin_fabsf(s.c11)), __builtin_fabsf(s.c01)); }
BH_DEV Sym2 sym2_inverse(Sym2 s) {
    const float det = s.c00 * s.c11 - s.c01 * s.c01;
    const float inv_det = det > 0.0f ? 1.0f / det : 0.0f;
    return Sym2{s.c11 * inv_det, -s.c01 * inv_det, s.c00 * inv_det};
}
BH_DEV float det2_strict(Sym2 s) {
    const float ad = s.c00 * s.c11;
    const float bc = s.c01 * s.c01;
    return ad - bc;
}
BH_DEV bool sym2_finite(Sym2 s) { return is_finite_f32(s.c00) && is_finite_f32(s.c11) && is_finite_f32(s.c01); }
BH_DEV Vec3A s3row0(Sym3 s) { return Vec3A{s.c00, s.c01, s.c02}; }
BH_DEV Vec3A s3row1(Sym3 s) { return Vec3A{s.c01, s.c11, s.c12}; }
BH_DEV Vec3A s3row2(Sym3 s) { return Vec3A{s.c02, s.c12, s.c22}; }
BH_DEV Vec3A sym3_mul_vec3(Sym3 s, Vec3A v) { return add(add(scale(s3row0(s), v.x), scale(s3row1(s), v.y)), scale(s3row2(s), v.z)); }
BH_DEV Sym3 sym3_scale(Sym3 s, float k) { return Sym3{s.c00 * k, s.c01 * k, s.c02 * k, s.c11 * k, s.c12 * k, s.c22 * k}; }
BH_DEV Mat3 sym3_mul_mat3(Sym3 s, const Mat3& m) { return from_cols(sym3_mul_vec3(s, col0(m)), sym3_mul_vec3(s, col1(m)), sym3_mul_vec3(s, col2(m))); }
BH_DEV Sym3 congruence(Sym3 s, const Mat3& m) {
    const Vec3A sr0 = sym3_mul_vec3(s, row0(m)), sr1 = sym3_mul_vec3(s, row1(m)), sr2 = sym3_mul_vec3(s, row2(m));
    return Sym3{dot(row0(m), sr0), dot(row0(m), sr1), dot(row0(m), sr2), dot(row1(m), sr1), dot(row1(m), sr2), dot(row2(m), sr2)};
}
BH_DEV Sym3 transpose_congruence(Sym3 s, const Mat3& m) {
    const Vec3A sc0 = sym3_mul_vec3(s, col0(m)), sc1 = sym3_mul_vec3(s, col1(m)), sc2 = sym3_mul_vec3(s, col2(m));
    return Sym3{dot(col0(m), sc0), dot(col0(m), sc1), dot(col0(m), sc2), dot(col1(m), sc1), dot(col1(m), sc2), dot(col2(m), sc2)};
}

// brush-cube/src/lib.rs:573-578, with the two sums single-rounded.
BH_DEV float calc_sigma(float px, float py, Sym2 conic, float xy_x, float xy_y) {
    const float dx = px - xy_x;
    const float dy = py - xy_y;
    const float q = __builtin_fmaf(conic.c11 * dy, dy, (conic.c00 * dx) * dx);
    return __builtin_fmaf(conic.c01 * dx, dy, 0.5f * q);
}

// helpers.rs:26-47
BH_DEV float alpha_cutoff_weight(float alpha) {
    const float t = clampf((alpha - (ALPHA_CUTOFF_MID - 0.5f * ALPHA_CUTOFF_BAND)) / ALPHA_CUTOFF_BAND, 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
BH_DEV float alpha_cutoff_weight_deriv(float alpha) {
    const float low = ALPHA_CUTOFF_MID - 0.5f * ALPHA_CUTOFF_BAND;
    const float high = ALPHA_CUTOFF_MID + 0.5f * ALPHA_CUTOFF_BAND;
    const bool inside = alpha > low && alpha < high;
    const float t = (alpha - low) / ALPHA_CUTOFF_BAND;
    return inside ? (6.0f * t - 6.0f * t * t) / ALPHA_CUTOFF_BAND : 0.0f;
}

// ---- view uniforms (kernels/types.rs:53-81), passed by value to kernels -------
struct ViewUniforms {
    float vm[12];
    float fx, fy, cx, cy;
    float lim_pos_x, lim_pos_y, lim_neg_x, lim_neg_y;
    float cam_x, cam_y, cam_z;
    uint32_t img_w, img_h, tile_bw, tile_bh;
    uint32_t tile_y0, tile_y1;  // tile-row window [y0, y1) rendered by this call (0, tile_bh = whole image)
    uint32_t model;             // BH_CAMERA_*
    float dist[8];              // distortion parameters of the model
    float half_fov;             // view-angle cull bound of the non-pinhole models
};
BH_DEV Mat3 view_rotation(const ViewUniforms& u) { return Mat3{u.vm[0], u.vm[1], u.vm[2], u.vm[3], u.vm[4], u.vm[5], u.vm[6], u.vm[7], u.vm[8]}; }
BH_DEV Vec3A view_translation(const ViewUniforms& u) { return Vec3A{u.vm[9], u.vm[10], u.vm[11]}; }
BH_DEV Vec3A camera_pos(const ViewUniforms& u) { return Vec3A{u.cam_x, u.cam_y, u.cam_z}; }
// helpers.rs:317-320
BH_DEV Vec3A world_to_cam(Vec3A mean, const ViewUniforms& u) { return add(mul_vec3(view_rotation(u), mean), view_translation(u)); }

// camera_model/pinhole.rs:25-57
BH_DEV void project_pinhole(Vec3A p, const ViewUniforms& u, float& ox, float& oy) {
    const float inv_z = 1.0f / p.z;
    ox = u.fx * p.x * inv_z + u.cx;
    oy = u.fy * p.y * inv_z + u.cy;
}
BH_DEV Mat2x3 jacobian_pinhole(Vec3A p, const ViewUniforms& u) {
    const float inv_z = 1.0f / p.z;
    const float dx = u.fx * inv_z;
    const float dy = u.fy * inv_z;
    const float clamped_x = clampf(p.x * inv_z, u.lim_neg_x, u.lim_pos_x);
    const float clamped_y = clampf(p.y * inv_z, u.lim_neg_y, u.lim_pos_y);
    Mat2x3 j;
    j.c0 = Vec2{dx, 0.0f};
    j.c1 = Vec2{0.0f, dy};
    j.c2 = Vec2{-dx * clamped_x, -dy * clamped_y};
    return j;
}

// helpers.rs:180-195
template <bool MIP>
BH_DEV Sym2 compensate_cov2d(Sym2 c, float& filter_comp) {
    const float cov_blur = MIP ? 0.1f : 0.3f;
    const Sym2 blurred = Sym2{c.c00 + cov_blur, c.c01, c.c11 + cov_blur};
    filter_comp = 1.0f;
    if (MIP) {
        const float det_raw = __builtin_fmaxf(det2_strict(c), 0.0f);
        const float det_blurred = det2_strict(blurred);
        filter_comp = __builtin_sqrtf(det_raw / det_blurred);
    }
    return blurred;
}

// helpers.rs:83-94
BH_DEV void compute_bbox_extent(Sym2 conic, float power_threshold, float& ex, float& ey) {
    const float det = conic.c00 * conic.c11 - conic.c01 * conic.c01;
    const bool degenerate = det <= 0.0f;
    const float inv_det = degenerate ? 0.0f : 1.0f / det;
    const float e_x = __builtin_sqrtf(2.0f * power_threshold * conic.c11 * inv_det);
    const float e_y = __builtin_sqrtf(2.0f * power_threshold * conic.c00 * inv_det);
    ex = degenerate ? -1.0f : e_x;
    ey = degenerate ? -1.0f : e_y;
}

struct TileBbox { uint32_t min_x, min_y, max_x, max_y; };

// helpers.rs:110-140
// `y0`, `y1`: the tile-row window (0, tile_bh for a whole-image render: the reference's clamp).
BH_DEV TileBbox get_tile_bbox(float cx, float cy, float ex, float ey, uint32_t bw, uint32_t y0, uint32_t y1) {
    const float tw = (float)TILE_WIDTH;
    const float x = cx / tw, y = cy / tw, dx = ex / tw, dy = ey / tw;
    const float bwf = (float)bw, y0f = (float)y0, y1f = (float)y1;
    TileBbox b;
    b.min_x = (uint32_t)clampf(x - dx, 0.0f, bwf);
    b.min_y = (uint32_t)clampf(y - dy, y0f, y1f);
    b.max_x = (uint32_t)clampf(x + dx + 1.0f, 0.0f, bwf);
    b.max_y = (uint32_t)clampf(y + dy + 1.0f, y0f, y1f);
    return b;
}

// helpers.rs:226-264 — the ONE tile/gaussian test shared by the counting pass
// (project_forward) and the emitting pass (map_gaussians): same function, same TU
// flags, so the two walks cannot drift.
BH_DEV bool will_primitive_contribute(uint32_t tx, uint32_t ty, float mx, float my, Sym2 conic, float power_threshold) {
    const float rmin_x = (float)(tx * TILE_WIDTH);
    const float rmin_y = (float)(ty * TILE_WIDTH);
    const float rmax_x = rmin_x + (float)TILE_WIDTH;
    const float rmax_y = rmin_y + (float)TILE_WIDTH;
    const bool x_left = mx < rmin_x;
    const bool x_right = mx > rmax_x;
    const bool in_x_range = !(x_left || x_right);
    const bool y_above = my < rmin_y;
    const bool y_below = my > rmax_y;
    const bool in_y_range = !(y_above || y_below);
    bool hit = in_x_range && in_y_range;
    if (!hit) {
        const float corner_x = x_left ? rmin_x : rmax_x;
        const float corner_y = y_above ? rmin_y : rmax_y;
        const float width = rmax_x - rmin_x;
        const float height = rmax_y - rmin_y;
        const float dxf = x_left ? width : -width;
        const float dyf = y_above ? height : -height;
        const float diff_x = mx - corner_x;
        const float diff_y = my - corner_y;
        const float tx_raw = (dxf * conic.c00 * diff_x + dxf * conic.c01 * diff_y) / (dxf * conic.c00 * dxf);
        const float ty_raw = (dyf * conic.c01 * diff_x + dyf * conic.c11 * diff_y) / (dyf * conic.c11 * dyf);
        const float t_x = in_y_range ? 0.0f : clampf(tx_raw, 0.0f, 1.0f);
        const float t_y = in_x_range ? 0.0f : clampf(ty_raw, 0.0f, 1.0f);
        const float max_x = corner_x + t_x * dxf;
        const float max_y = corner_y + t_y * dyf;
        hit = calc_sigma(max_x, max_y, conic, mx, my) <= power_threshold;
    }
    return hit;
}

BH_DEV constexpr uint32_t num_sh_coeffs(uint32_t degree) { return (degree + 1) * (degree + 1); }

}  // namespace bh
