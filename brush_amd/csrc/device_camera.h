// device_camera.h — the four camera models of the project kernels (gfx950 device code).
//
// Reference: brush-render/src/kernels/camera_model/{mod,pinhole,kannala_brandt_4,
// radial_tangential_8,thin_prism_fisheye}.rs (paths under /root/reference/crates).  The
// reference specialises its kernels on the model AND its parameter values at JIT time
// (#[comptime] CameraModel); here the kernels are specialised on "pinhole or not" only
// (template<bool PINHOLE>): the pinhole instance carries no distortion code at all, the
// other instance dispatches on ViewUniforms::model with wave-uniform scalar branches and
// reads the parameters from kernel arguments (SGPRs), so a new lens never recompiles.
//
// Arithmetic contract: device_math.h (IEEE f32 in the reference's operation order, no
// contraction).  atan2 — left to the shader compiler by the reference — is the fixed
// polynomial bh_atan2f below, so culling and tile assignment are reproducible bit-for-bit.
#pragma once
#include "device_math.h"

namespace bh {

constexpr uint32_t CAM_PINHOLE = 0, CAM_KB4 = 1, CAM_RT8 = 2, CAM_TPF = 3;

// Cephes-style single-precision arctangent on [0, inf): range reduction at tan(pi/8) and
// tan(3pi/8), odd degree-9 polynomial evaluated with explicit fma.
BH_DEV float bh_atanf_pos(float x) {
    float y0 = 0.0f;
    if (x > 2.414213562373095f) {
        y0 = 1.5707963267948966f;
        x = -1.0f / x;
    } else if (x > 0.4142135623730950f) {
        y0 = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    }
    const float z = x * x;
    float p = 8.05374449538e-2f;
    p = __builtin_fmaf(p, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    return y0 + __builtin_fmaf(p * z, x, x);
}
BH_DEV float bh_atan2f(float y, float x) {
    constexpr float PI_F = 3.14159265358979323846f;
    if (x != x || y != y) return x + y;
    const bool x_neg = (f2u(x) >> 31) != 0u, y_neg = (f2u(y) >> 31) != 0u;
    if (y == 0.0f) return (x < 0.0f || (x == 0.0f && x_neg)) ? (y_neg ? -PI_F : PI_F) : y;
    const float ay = __builtin_fabsf(y), ax = __builtin_fabsf(x);
    float a;
    if (ax == __builtin_inff() && ay == __builtin_inff()) a = 0.7853981633974483f;
    else a = bh_atanf_pos(ay / ax);
    if (x < 0.0f) a = PI_F - a;
    return y < 0.0f ? -a : a;
}

// ---- Kannala-Brandt 4 (kannala_brandt_4.rs) ------------------------------------------------
// :19-53.  kk = k1..k4
BH_DEV void project_kb4(Vec3A point, const ViewUniforms& u, const float* kk, float& ou, float& ov) {
    const float x = point.x, y = point.y, z = point.z;
    const float k1 = kk[0], k2 = kk[1], k3 = kk[2], k4 = kk[3];
    const float inv_z = 1.0f / z;
    const float pinhole_u = u.fx * x * inv_z + u.cx;
    const float pinhole_v = u.fy * y * inv_z + u.cy;
    const float r = __builtin_sqrtf(x * x + y * y);
    const float theta = bh_atan2f(r, z);
    const float theta2 = theta * theta;
    const float theta4 = theta2 * theta2;
    const float theta6 = theta2 * theta4;
    const float theta8 = theta4 * theta4;
    const float d = theta * (1.0f + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
    const float inv_r = 1.0f / r;
    const float fisheye_u = u.fx * (d * x * inv_r) + u.cx;
    const float fisheye_v = u.fy * (d * y * inv_r) + u.cy;
    const bool near_axis = r < 1e-6f;
    ou = near_axis ? pinhole_u : fisheye_u;
    ov = near_axis ? pinhole_v : fisheye_v;
}

// :57-152 (no Jacobian clamp: the fisheye law stays bounded as theta grows)
BH_DEV Mat2x3 jacobian_kb4(Vec3A point, const ViewUniforms& u, const float* kk) {
    const float fx = u.fx, fy = u.fy;
    const float k1 = kk[0], k2 = kk[1], k3 = kk[2], k4 = kk[3];
    const float x = point.x, y = point.y, z = point.z;
    const float inv_z = 1.0f / z;
    const float x2 = x * x, y2 = y * y, xy = x * y;
    const float r2 = x2 + y2;
    const float r = __builtin_sqrtf(r2);
    const float inv_r = 1.0f / r;
    const float inv_r3 = inv_r * inv_r * inv_r;
    const float rho2 = r2 + z * z;
    const float inv_rho2 = 1.0f / rho2;
    const float inv_rho2_r = inv_rho2 * inv_r;
    const float theta = bh_atan2f(r, z);
    const float theta2 = theta * theta;
    const float theta4 = theta2 * theta2;
    const float theta6 = theta4 * theta2;
    const float theta8 = theta4 * theta4;
    const float d = theta * (1.0f + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
    const float dd_dtheta = 1.0f + 3.0f * k1 * theta2 + 5.0f * k2 * theta4 + 7.0f * k3 * theta6 + 9.0f * k4 * theta8;
    // d theta / d(x,y,z)
    const float dth_dx = x * z * inv_rho2_r;
    const float dth_dy = y * z * inv_rho2_r;
    const float dth_dz = -r * inv_rho2;
    const float dd_dx = dd_dtheta * dth_dx;
    const float dd_dy = dd_dtheta * dth_dy;
    const float dd_dz = dd_dtheta * dth_dz;
    // u row
    const float xr = x * inv_r;
    const float dxr_dx = y2 * inv_r3;
    const float dxr_dy = -xy * inv_r3;
    const float du_dx = fx * (dd_dx * xr + d * dxr_dx);
    const float du_dy = fx * (dd_dy * xr + d * dxr_dy);
    const float du_dz = fx * (dd_dz * xr);
    // v row
    const float yr = y * inv_r;
    const float dyr_dx = -xy * inv_r3;
    const float dyr_dy = x2 * inv_r3;
    const float dv_dx = fy * (dd_dx * yr + d * dyr_dx);
    const float dv_dy = fy * (dd_dy * yr + d * dyr_dy);
    const float dv_dz = fy * (dd_dz * yr);
    // on the optical axis: the pinhole Jacobian
    const bool near_axis = r < 1e-6f;
    const float dx = fx * inv_z;
    const float dy = fy * inv_z;
    const float ph_du_dz = -dx * x * inv_z;
    const float ph_dv_dz = -dy * y * inv_z;
    Mat2x3 j;
    j.c0 = Vec2{near_axis ? dx : du_dx, near_axis ? 0.0f : dv_dx};
    j.c1 = Vec2{near_axis ? 0.0f : du_dy, near_axis ? dy : dv_dy};
    j.c2 = Vec2{near_axis ? ph_du_dz : du_dz, near_axis ? ph_dv_dz : dv_dz};
    return j;
}

// The six entries of 2 * (v_cov2d * J) * cov_c shared by the fisheye VJPs.
struct VJ6 { float u0, u1, u2, v0, v1, v2; };
BH_DEV VJ6 vjp_two_vcov_j_cov(Sym2 v_cov2d, const Mat2x3& jac, Sym3 cov_c) {
    const Mat2x3 tmp = sym2_mul_mat2x3(v_cov2d, jac);
    VJ6 r;
    r.u0 = 2.0f * dot(row0(tmp), s3row0(cov_c));
    r.u1 = 2.0f * dot(row0(tmp), s3row1(cov_c));
    r.u2 = 2.0f * dot(row0(tmp), s3row2(cov_c));
    r.v0 = 2.0f * dot(row1(tmp), s3row0(cov_c));
    r.v1 = 2.0f * dot(row1(tmp), s3row1(cov_c));
    r.v2 = 2.0f * dot(row1(tmp), s3row2(cov_c));
    return r;
}

// :154-337
BH_DEV Vec3A projection_vjp_kb4(const Mat2x3& jac, Vec3A mean_c, Sym3 cov_c, const ViewUniforms& u, Sym2 v_cov2d, Vec2 v_mean2d,
                                const float* kk) {
    const float fx = u.fx, fy = u.fy;
    const float k1 = kk[0], k2 = kk[1], k3 = kk[2], k4 = kk[3];
    const float mx = mean_c.x, my = mean_c.y, mz = mean_c.z;
    const float r2 = mx * mx + my * my;
    const float r = __builtin_fmaxf(__builtin_sqrtf(r2), 1.0e-8f);
    const float rho2 = r2 + mz * mz;
    const float theta = bh_atan2f(r, mz);
    const float th2 = theta * theta;
    const float th4 = th2 * th2;
    const float th6 = th4 * th2;
    const float th8 = th4 * th4;
    const float theta_d = theta * (1.0f + k1 * th2 + k2 * th4 + k3 * th6 + k4 * th8);
    const float p1 = 1.0f + 3.0f * k1 * th2 + 5.0f * k2 * th4 + 7.0f * k3 * th6 + 9.0f * k4 * th8;                     // d theta_d / d theta
    const float p2 = 6.0f * k1 * theta + 20.0f * k2 * theta * th2 + 42.0f * k3 * theta * th4 + 72.0f * k4 * theta * th6;  // second derivative
    const float inv_r = 1.0f / r;
    const float inv_r3 = inv_r * inv_r * inv_r;
    const float inv_r5 = inv_r3 * inv_r * inv_r;
    const float inv_rho2 = 1.0f / rho2;
    const float inv_rho2_sq = inv_rho2 * inv_rho2;
    const float inv_rho2_r = inv_rho2 * inv_r;
    const float dth_x = mx * mz * inv_rho2_r;
    const float dth_y = my * mz * inv_rho2_r;
    const float dth_z = -r * inv_rho2;
    const float xr = mx * inv_r;
    const float yr = my * inv_r;
    const float dxr_x = my * my * inv_r3;
    const float dxr_y = -mx * my * inv_r3;
    const float dyr_x = dxr_y;
    const float dyr_y = mx * mx * inv_r3;
    const float dg_x = p1 * dth_x;
    const float dg_y = p1 * dth_y;
    const float dg_z = p1 * dth_z;
    // path 1: J^T v_mean2d
    float v_mx = dot(v_mean2d, jac.c0);
    float v_my = dot(v_mean2d, jac.c1);
    float v_mz = dot(v_mean2d, jac.c2);
    const VJ6 vj = vjp_two_vcov_j_cov(v_cov2d, jac, cov_c);
    // Hessian of theta
    const float three_r2_z2 = 3.0f * r2 + mz * mz;
    const float r2_minus_z2 = r2 - mz * mz;
    const float h_th_00 = mz * (r2 * rho2 - mx * mx * three_r2_z2) * inv_r3 * inv_rho2_sq;
    const float h_th_11 = mz * (r2 * rho2 - my * my * three_r2_z2) * inv_r3 * inv_rho2_sq;
    const float h_th_01 = -mx * my * mz * three_r2_z2 * inv_r3 * inv_rho2_sq;
    const float h_th_02 = mx * r2_minus_z2 * inv_r * inv_rho2_sq;
    const float h_th_12 = my * r2_minus_z2 * inv_r * inv_rho2_sq;
    const float h_th_22 = 2.0f * mz * r * inv_rho2_sq;
    // Hessians of x/r and y/r (xy block only)
    const float two_x2_my2 = 2.0f * mx * mx - my * my;
    const float two_y2_mx2 = 2.0f * my * my - mx * mx;
    const float h_xr_00 = -3.0f * mx * my * my * inv_r5;
    const float h_xr_01 = my * two_x2_my2 * inv_r5;
    const float h_xr_11 = mx * two_y2_mx2 * inv_r5;
    const float h_yr_00 = my * two_x2_my2 * inv_r5;
    const float h_yr_01 = mx * two_y2_mx2 * inv_r5;
    const float h_yr_11 = -3.0f * mx * mx * my * inv_r5;
    // path 2: sum_j v_J[i,j] * dJ[i,j]/dk, in the reference's (j, k) order
    {
        const float d2g = p2 * dth_x * dth_x + p1 * h_th_00;
        const float d_ju = fx * (d2g * xr + dg_x * dxr_x + dg_x * dxr_x + theta_d * h_xr_00);
        const float d_jv = fy * (d2g * yr + dg_x * dyr_x + dg_x * dyr_x + theta_d * h_yr_00);
        v_mx += vj.u0 * d_ju + vj.v0 * d_jv;
    }
    {
        const float d2g = p2 * dth_y * dth_x + p1 * h_th_01;
        const float d_ju = fx * (d2g * xr + dg_y * dxr_x + dg_x * dxr_y + theta_d * h_xr_01);
        const float d_jv = fy * (d2g * yr + dg_y * dyr_x + dg_x * dyr_y + theta_d * h_yr_01);
        v_mx += vj.u1 * d_ju + vj.v1 * d_jv;
    }
    {
        const float d2g = p2 * dth_z * dth_x + p1 * h_th_02;
        const float d_ju = fx * (d2g * xr + dg_x * 0.0f + dg_z * dxr_x);
        const float d_jv = fy * (d2g * yr + dg_x * 0.0f + dg_z * dyr_x);
        v_mx += vj.u2 * d_ju + vj.v2 * d_jv;
    }
    {
        const float d2g = p2 * dth_x * dth_y + p1 * h_th_01;
        const float d_ju = fx * (d2g * xr + dg_x * dxr_y + dg_y * dxr_x + theta_d * h_xr_01);
        const float d_jv = fy * (d2g * yr + dg_x * dyr_y + dg_y * dyr_x + theta_d * h_yr_01);
        v_my += vj.u0 * d_ju + vj.v0 * d_jv;
    }
    {
        const float d2g = p2 * dth_y * dth_y + p1 * h_th_11;
        const float d_ju = fx * (d2g * xr + dg_y * dxr_y + dg_y * dxr_y + theta_d * h_xr_11);
        const float d_jv = fy * (d2g * yr + dg_y * dyr_y + dg_y * dyr_y + theta_d * h_yr_11);
        v_my += vj.u1 * d_ju + vj.v1 * d_jv;
    }
    {
        const float d2g = p2 * dth_z * dth_y + p1 * h_th_12;
        const float d_ju = fx * (d2g * xr + dg_y * 0.0f + dg_z * dxr_y);
        const float d_jv = fy * (d2g * yr + dg_y * 0.0f + dg_z * dyr_y);
        v_my += vj.u2 * d_ju + vj.v2 * d_jv;
    }
    {
        const float d2g = p2 * dth_x * dth_z + p1 * h_th_02;
        const float d_ju = fx * (d2g * xr + dg_z * dxr_x + dg_x * 0.0f);
        const float d_jv = fy * (d2g * yr + dg_z * dyr_x + dg_x * 0.0f);
        v_mz += vj.u0 * d_ju + vj.v0 * d_jv;
    }
    {
        const float d2g = p2 * dth_y * dth_z + p1 * h_th_12;
        const float d_ju = fx * (d2g * xr + dg_z * dxr_y + dg_y * 0.0f);
        const float d_jv = fy * (d2g * yr + dg_z * dyr_y + dg_y * 0.0f);
        v_mz += vj.u1 * d_ju + vj.v1 * d_jv;
    }
    {
        const float d2g = p2 * dth_z * dth_z + p1 * h_th_22;
        const float d_ju = fx * (d2g * xr);
        const float d_jv = fy * (d2g * yr);
        v_mz += vj.u2 * d_ju + vj.v2 * d_jv;
    }
    return Vec3A{v_mx, v_my, v_mz};
}

// ---- radial-tangential 8 (radial_tangential_8.rs); dd = k1 k2 k3 k4 k5 k6 p1 p2 ------------
// :23-67
BH_DEV void project_rt8(Vec3A point, const ViewUniforms& u, const float* dd, float& ou, float& ov) {
    const float k1 = dd[0], k2 = dd[1], k3 = dd[2], k4 = dd[3], k5 = dd[4], k6 = dd[5], p1 = dd[6], p2 = dd[7];
    const float xn = point.x / point.z;
    const float yn = point.y / point.z;
    const float xn2 = xn * xn;
    const float yn2 = yn * yn;
    const float r2 = xn2 + yn2;
    const float r4 = r2 * r2;
    const float r6 = r4 * r2;
    const float d = (1.0f + k1 * r2 + k2 * r4 + k3 * r6) / (1.0f + k4 * r2 + k5 * r4 + k6 * r6);
    const float xyn = xn * yn;
    const float xd = xn * d + 2.0f * p1 * xyn + p2 * (r2 + 2.0f * xn2);
    const float yd = yn * d + 2.0f * p2 * xyn + p1 * (r2 + 2.0f * yn2);
    ou = u.fx * xd + u.cx;
    ov = u.fy * yd + u.cy;
}

// Radial law R(r2) = N/Dn, its derivatives and the 2x2 distortion Jacobian D at (x, y).
struct Rt8Local {
    float rr, rrp, rrpp;        // R, R', R''  (w.r.t. r2)
    float d00, d01, d11;        // D (symmetric)
};
template <bool SECOND>
BH_DEV Rt8Local rt8_local(float x, float y, const float* dd) {
    const float k1 = dd[0], k2 = dd[1], k3 = dd[2], k4 = dd[3], k5 = dd[4], k6 = dd[5], p1 = dd[6], p2 = dd[7];
    const float r2 = x * x + y * y;
    const float r4 = r2 * r2;
    Rt8Local o;
    float n_poly, dn_poly;
    if (SECOND) {  // the VJP writes the r^6 terms as k * r2 * r4 (radial_tangential_8.rs:203-204)
        n_poly = 1.0f + k1 * r2 + k2 * r4 + k3 * r2 * r4;
        dn_poly = 1.0f + k4 * r2 + k5 * r4 + k6 * r2 * r4;
    } else {       // the Jacobian as k * r6 with r6 = r4 * r2 (:101-106)
        const float r6 = r4 * r2;
        n_poly = 1.0f + k1 * r2 + k2 * r4 + k3 * r6;
        dn_poly = 1.0f + k4 * r2 + k5 * r4 + k6 * r6;
    }
    const float np_poly = k1 + 2.0f * k2 * r2 + 3.0f * k3 * r4;
    const float dnp_poly = k4 + 2.0f * k5 * r2 + 3.0f * k6 * r4;
    const float inv_dn = 1.0f / dn_poly;
    const float inv_dn2 = inv_dn * inv_dn;
    o.rr = n_poly * inv_dn;
    o.rrp = (np_poly * dn_poly - n_poly * dnp_poly) * inv_dn2;
    o.rrpp = 0.0f;
    if (SECOND) {
        const float npp_poly = 2.0f * k2 + 6.0f * k3 * r2;
        const float dnpp_poly = 2.0f * k5 + 6.0f * k6 * r2;
        const float inv_dn3 = inv_dn2 * inv_dn;
        o.rrpp = (npp_poly * dn_poly * dn_poly - 2.0f * np_poly * dn_poly * dnp_poly - n_poly * dnpp_poly * dn_poly +
                  2.0f * n_poly * dnp_poly * dnp_poly) * inv_dn3;
    }
    o.d00 = o.rr + 2.0f * x * x * o.rrp + 2.0f * p1 * y + 6.0f * p2 * x;
    o.d01 = 2.0f * x * y * o.rrp + 2.0f * p1 * x + 2.0f * p2 * y;
    o.d11 = o.rr + 2.0f * y * y * o.rrp + 6.0f * p1 * y + 2.0f * p2 * x;
    return o;
}

// :69-149 — J = diag(fx, fy) * D * M at the clamped normalised point
BH_DEV Mat2x3 jacobian_rt8(Vec3A point, const ViewUniforms& u, const float* dd) {
    const float z = point.z;
    const float inv_z = 1.0f / z;
    const float inv_z2 = inv_z * inv_z;
    const float x_n = clampf(point.x * inv_z, u.lim_neg_x, u.lim_pos_x);
    const float y_n = clampf(point.y * inv_z, u.lim_neg_y, u.lim_pos_y);
    const float xc = x_n * z;
    const float yc = y_n * z;
    const Rt8Local l = rt8_local<false>(x_n, y_n, dd);
    Mat2x3 j;
    j.c0 = Vec2{u.fx * l.d00 * inv_z, u.fy * l.d01 * inv_z};
    j.c1 = Vec2{u.fx * l.d01 * inv_z, u.fy * l.d11 * inv_z};
    j.c2 = Vec2{-u.fx * (l.d00 * xc + l.d01 * yc) * inv_z2, -u.fy * (l.d01 * xc + l.d11 * yc) * inv_z2};
    return j;
}

// :151-377 — VJP through the clamp "surrogate point" (xc, yc, Z)
BH_DEV Vec3A projection_vjp_rt8(Vec3A mean_c, Sym3 cov_c, const ViewUniforms& u, Sym2 v_cov2d, Vec2 v_mean2d, const float* dd) {
    const float fx = u.fx, fy = u.fy;
    const float p1 = dd[6], p2 = dd[7];
    const float mx = mean_c.x, my = mean_c.y, mz = mean_c.z;
    const float inv_z = 1.0f / mz;
    const float mx_rz_raw = mx * inv_z;
    const float my_rz_raw = my * inv_z;
    const float mx_rz = clampf(mx_rz_raw, u.lim_neg_x, u.lim_pos_x);
    const float my_rz = clampf(my_rz_raw, u.lim_neg_y, u.lim_pos_y);
    const bool in_x = mx_rz_raw <= u.lim_pos_x && mx_rz_raw >= u.lim_neg_x;
    const bool in_y = my_rz_raw <= u.lim_pos_y && my_rz_raw >= u.lim_neg_y;
    const float xc = mx_rz * mz;
    const float yc = my_rz * mz;
    const float inv_z2 = inv_z * inv_z;
    const float inv_z3 = inv_z2 * inv_z;
    const float x = xc * inv_z;
    const float y = yc * inv_z;
    const Rt8Local l = rt8_local<true>(x, y, dd);
    const float rrp = l.rrp, rrpp = l.rrpp;
    const float d00 = l.d00, d01 = l.d01, d10 = l.d01, d11 = l.d11;
    const float rx = 2.0f * x * rrp;
    const float ry = 2.0f * y * rrp;
    const float rpx = 2.0f * x * rrpp;
    const float rpy = 2.0f * y * rrpp;
    // surrogate Jacobian
    const float js00 = fx * d00 * inv_z;
    const float js01 = fx * d01 * inv_z;
    const float js02 = -fx * (d00 * xc + d01 * yc) * inv_z2;
    const float js10 = fy * d10 * inv_z;
    const float js11 = fy * d11 * inv_z;
    const float js12 = -fy * (d10 * xc + d11 * yc) * inv_z2;
    // effective Jacobian: routed through the clamp
    const float je00 = in_x ? js00 : 0.0f;
    const float je10 = in_x ? js10 : 0.0f;
    const float je01 = in_y ? js01 : 0.0f;
    const float je11 = in_y ? js11 : 0.0f;
    const float je02 = (in_x ? 0.0f : mx_rz * js00) + (in_y ? 0.0f : my_rz * js01) + js02;
    const float je12 = (in_x ? 0.0f : mx_rz * js10) + (in_y ? 0.0f : my_rz * js11) + js12;
    float v_mx = je00 * v_mean2d.x + je10 * v_mean2d.y;
    float v_my = je01 * v_mean2d.x + je11 * v_mean2d.y;
    float v_mz = je02 * v_mean2d.x + je12 * v_mean2d.y;
    Mat2x3 je;
    je.c0 = Vec2{je00, je10};
    je.c1 = Vec2{je01, je11};
    je.c2 = Vec2{je02, je12};
    const VJ6 ve = vjp_two_vcov_j_cov(v_cov2d, je, cov_c);
    const float vs_u0 = in_x ? ve.u0 : mx_rz * ve.u2;
    const float vs_v0 = in_x ? ve.v0 : mx_rz * ve.v2;
    const float vs_u1 = in_y ? ve.u1 : my_rz * ve.u2;
    const float vs_v1 = in_y ? ve.v1 : my_rz * ve.v2;
    const float vs_u2 = ve.u2;
    const float vs_v2 = ve.v2;
    // derivatives of D in the normalised coordinates
    const float dd00_dx = rx + 4.0f * x * rrp + 2.0f * x * x * rpx + 6.0f * p2;
    const float dd00_dy = ry + 2.0f * x * x * rpy + 2.0f * p1;
    const float dd01_dx = 2.0f * y * rrp + 2.0f * x * y * rpx + 2.0f * p1;
    const float dd01_dy = 2.0f * x * rrp + 2.0f * x * y * rpy + 2.0f * p2;
    const float dd10_dx = dd01_dx;
    const float dd10_dy = dd01_dy;
    const float dd11_dx = rx + 2.0f * y * y * rpx + 2.0f * p2;
    const float dd11_dy = ry + 4.0f * y * rrp + 2.0f * y * y * rpy + 6.0f * p1;
    // chain to (xc, yc, Z)
    const float dd00_dxc = dd00_dx * inv_z;
    const float dd00_dyc = dd00_dy * inv_z;
    const float dd00_dz = -(xc * dd00_dx + yc * dd00_dy) * inv_z2;
    const float dd01_dxc = dd01_dx * inv_z;
    const float dd01_dyc = dd01_dy * inv_z;
    const float dd01_dz = -(xc * dd01_dx + yc * dd01_dy) * inv_z2;
    const float dd10_dxc = dd10_dx * inv_z;
    const float dd10_dyc = dd10_dy * inv_z;
    const float dd10_dz = -(xc * dd10_dx + yc * dd10_dy) * inv_z2;
    const float dd11_dxc = dd11_dx * inv_z;
    const float dd11_dyc = dd11_dy * inv_z;
    const float dd11_dz = -(xc * dd11_dx + yc * dd11_dy) * inv_z2;
    const float djs00_dxc = fx * dd00_dxc * inv_z;
    const float djs00_dyc = fx * dd00_dyc * inv_z;
    const float djs00_dz = fx * (dd00_dz * inv_z - d00 * inv_z2);
    const float djs01_dxc = fx * dd01_dxc * inv_z;
    const float djs01_dyc = fx * dd01_dyc * inv_z;
    const float djs01_dz = fx * (dd01_dz * inv_z - d01 * inv_z2);
    const float djs10_dxc = fy * dd10_dxc * inv_z;
    const float djs10_dyc = fy * dd10_dyc * inv_z;
    const float djs10_dz = fy * (dd10_dz * inv_z - d10 * inv_z2);
    const float djs11_dxc = fy * dd11_dxc * inv_z;
    const float djs11_dyc = fy * dd11_dyc * inv_z;
    const float djs11_dz = fy * (dd11_dz * inv_z - d11 * inv_z2);
    const float djs02_dxc = -fx * (dd00_dxc * xc + d00 + dd01_dxc * yc) * inv_z2;
    const float djs02_dyc = -fx * (dd00_dyc * xc + dd01_dyc * yc + d01) * inv_z2;
    const float djs02_dz = -fx * ((dd00_dz * xc + dd01_dz * yc) * inv_z2 - 2.0f * (d00 * xc + d01 * yc) * inv_z3);
    const float djs12_dxc = -fy * (dd10_dxc * xc + d10 + dd11_dxc * yc) * inv_z2;
    const float djs12_dyc = -fy * (dd10_dyc * xc + dd11_dyc * yc + d11) * inv_z2;
    const float djs12_dz = -fy * ((dd10_dz * xc + dd11_dz * yc) * inv_z2 - 2.0f * (d10 * xc + d11 * yc) * inv_z3);
    const float c_xc = vs_u0 * djs00_dxc + vs_u1 * djs01_dxc + vs_u2 * djs02_dxc + vs_v0 * djs10_dxc + vs_v1 * djs11_dxc + vs_v2 * djs12_dxc;
    const float c_yc = vs_u0 * djs00_dyc + vs_u1 * djs01_dyc + vs_u2 * djs02_dyc + vs_v0 * djs10_dyc + vs_v1 * djs11_dyc + vs_v2 * djs12_dyc;
    const float c_z = vs_u0 * djs00_dz + vs_u1 * djs01_dz + vs_u2 * djs02_dz + vs_v0 * djs10_dz + vs_v1 * djs11_dz + vs_v2 * djs12_dz;
    if (in_x) v_mx += c_xc;
    if (in_y) v_my += c_yc;
    v_mz += c_z;
    if (!in_x) v_mz += mx_rz * c_xc;
    if (!in_y) v_mz += my_rz * c_yc;
    return Vec3A{v_mx, v_my, v_mz};
}

// ---- thin-prism fisheye (thin_prism_fisheye.rs); dd = k1..k4 p1 p2 sx1 sy1 -------------------
struct TpPolys { float nu, nv, dnu_dx, dnu_dy, dnv_dx, dnv_dy; };
// :34-58
BH_DEV TpPolys thin_prism_polys(float x, float y, const float* dd) {
    const float p1 = dd[4], p2 = dd[5], sx1 = dd[6], sy1 = dd[7];
    const float x2 = x * x, y2 = y * y, xy = x * y;
    const float r2 = x2 + y2;
    TpPolys t;
    t.nu = 2.0f * p1 * xy + p2 * (3.0f * x2 + y2) + sx1 * r2;
    t.nv = 2.0f * p2 * xy + p1 * (x2 + 3.0f * y2) + sy1 * r2;
    t.dnu_dx = 2.0f * (p1 * y + (3.0f * p2 + sx1) * x);
    t.dnu_dy = 2.0f * (p1 * x + (p2 + sx1) * y);
    t.dnv_dx = 2.0f * (p2 * y + (p1 + sy1) * x);
    t.dnv_dy = 2.0f * (p2 * x + (3.0f * p1 + sy1) * y);
    return t;
}
// :60-78
BH_DEV void project_tpf(Vec3A point, const ViewUniforms& u, const float* dd, float& ou, float& ov) {
    float ku, kv;
    project_kb4(point, u, dd, ku, kv);
    const float inv_z = 1.0f / point.z;
    const float inv_z2 = inv_z * inv_z;
    const TpPolys t = thin_prism_polys(point.x, point.y, dd);
    ou = ku + u.fx * t.nu * inv_z2;
    ov = kv + u.fy * t.nv * inv_z2;
}
// :80-112
BH_DEV Mat2x3 jacobian_tpf(Vec3A point, const ViewUniforms& u, const float* dd) {
    const Mat2x3 kj = jacobian_kb4(point, u, dd);
    const float inv_z = 1.0f / point.z;
    const float inv_z2 = inv_z * inv_z;
    const float inv_z3 = inv_z2 * inv_z;
    const TpPolys t = thin_prism_polys(point.x, point.y, dd);
    const float add_du_dx = u.fx * t.dnu_dx * inv_z2;
    const float add_du_dy = u.fx * t.dnu_dy * inv_z2;
    const float add_du_dz = -2.0f * u.fx * t.nu * inv_z3;
    const float add_dv_dx = u.fy * t.dnv_dx * inv_z2;
    const float add_dv_dy = u.fy * t.dnv_dy * inv_z2;
    const float add_dv_dz = -2.0f * u.fy * t.nv * inv_z3;
    Mat2x3 j;
    j.c0 = Vec2{kj.c0.x + add_du_dx, kj.c0.y + add_dv_dx};
    j.c1 = Vec2{kj.c1.x + add_du_dy, kj.c1.y + add_dv_dy};
    j.c2 = Vec2{kj.c2.x + add_du_dz, kj.c2.y + add_dv_dz};
    return j;
}
// :114-203 — the KB4 VJP on the FULL Jacobian plus the Hessians of N_u/z^2, N_v/z^2
BH_DEV Vec3A projection_vjp_tpf(const Mat2x3& jac, Vec3A mean_c, Sym3 cov_c, const ViewUniforms& u, Sym2 v_cov2d, Vec2 v_mean2d,
                                const float* dd) {
    const Vec3A kb4_grad = projection_vjp_kb4(jac, mean_c, cov_c, u, v_cov2d, v_mean2d, dd);
    const float fx = u.fx, fy = u.fy;
    const float p1 = dd[4], p2 = dd[5], sx1 = dd[6], sy1 = dd[7];
    const float inv_z = 1.0f / mean_c.z;
    const float inv_z2 = inv_z * inv_z;
    const float inv_z3 = inv_z2 * inv_z;
    const float inv_z4 = inv_z2 * inv_z2;
    const TpPolys t = thin_prism_polys(mean_c.x, mean_c.y, dd);
    const float h_u_00 = (6.0f * p2 + 2.0f * sx1) * inv_z2;
    const float h_u_01 = (2.0f * p1) * inv_z2;
    const float h_u_11 = (2.0f * p2 + 2.0f * sx1) * inv_z2;
    const float h_u_02 = -2.0f * t.dnu_dx * inv_z3;
    const float h_u_12 = -2.0f * t.dnu_dy * inv_z3;
    const float h_u_22 = 6.0f * t.nu * inv_z4;
    const float h_v_00 = (2.0f * p1 + 2.0f * sy1) * inv_z2;
    const float h_v_01 = (2.0f * p2) * inv_z2;
    const float h_v_11 = (6.0f * p1 + 2.0f * sy1) * inv_z2;
    const float h_v_02 = -2.0f * t.dnv_dx * inv_z3;
    const float h_v_12 = -2.0f * t.dnv_dy * inv_z3;
    const float h_v_22 = 6.0f * t.nv * inv_z4;
    const VJ6 vj = vjp_two_vcov_j_cov(v_cov2d, jac, cov_c);
    const float v_mx = fx * (vj.u0 * h_u_00 + vj.u1 * h_u_01 + vj.u2 * h_u_02) + fy * (vj.v0 * h_v_00 + vj.v1 * h_v_01 + vj.v2 * h_v_02);
    const float v_my = fx * (vj.u0 * h_u_01 + vj.u1 * h_u_11 + vj.u2 * h_u_12) + fy * (vj.v0 * h_v_01 + vj.v1 * h_v_11 + vj.v2 * h_v_12);
    const float v_mz = fx * (vj.u0 * h_u_02 + vj.u1 * h_u_12 + vj.u2 * h_u_22) + fy * (vj.v0 * h_v_02 + vj.v1 * h_v_12 + vj.v2 * h_v_22);
    return Vec3A{kb4_grad.x + v_mx, kb4_grad.y + v_my, kb4_grad.z + v_mz};
}

// ---- dispatch (camera_model/mod.rs:40-125) ---------------------------------------------------
// project_forward.rs:47-61: the model's visibility gate on the camera-space mean
template <bool PINHOLE>
BH_DEV bool in_front_of_camera(Vec3A mean_c, const ViewUniforms& u) {
    if (PINHOLE) return !(mean_c.z < 0.01f);
    const float r = __builtin_sqrtf(mean_c.x * mean_c.x + mean_c.y * mean_c.y);
    return !(bh_atan2f(r, mean_c.z) > u.half_fov);
}
template <bool PINHOLE>
BH_DEV void project_point(Vec3A p, const ViewUniforms& u, float& ox, float& oy) {
    if (PINHOLE) { project_pinhole(p, u, ox, oy); return; }
    if (u.model == CAM_KB4) project_kb4(p, u, u.dist, ox, oy);
    else if (u.model == CAM_RT8) project_rt8(p, u, u.dist, ox, oy);
    else project_tpf(p, u, u.dist, ox, oy);
}
template <bool PINHOLE>
BH_DEV Mat2x3 project_jacobian(Vec3A p, const ViewUniforms& u) {
    if (PINHOLE) return jacobian_pinhole(p, u);
    if (u.model == CAM_KB4) return jacobian_kb4(p, u, u.dist);
    if (u.model == CAM_RT8) return jacobian_rt8(p, u, u.dist);
    return jacobian_tpf(p, u, u.dist);
}

// helpers.rs:145-175
template <bool PINHOLE>
BH_DEV Sym2 calc_cov2d(Vec3A scl, Quat quat, Vec3A mean_c, const ViewUniforms& u) {
    const Mat3 ns = mul_diag(mul_mat3(view_rotation(u), quat_to_mat3(quat)), scl);
    const Mat2x3 jac = project_jacobian<PINHOLE>(mean_c, u);
    const Mat2x3 v = mul_mat3(jac, ns);
    const Sym2 raw = gram_matrix(v);
    const float lim = 1.0e18f;
    const float max_abs = sym2_max_abs(raw);
    const float scale_down = max_abs > lim ? lim / max_abs : 1.0f;
    return sym2_scale(raw, scale_down);
}

}  // namespace bh
