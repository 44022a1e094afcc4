// device_sh.h — real spherical harmonics, degree 0..4 (forward + VJPs).
// Follows brush-render/src/kernels/sh.rs:47-355 (bases per Sloan, JCGT 2013).
// Coefficients are [C,3] floats, packed, per splat.
#pragma once
#include "device_math.h"

namespace bh {

struct ShBasis {
    float b[25];
};

// The 25 basis polynomials evaluated at unit `v`, in the reference's order and
// operation sequence (sh.rs:55-131); entry 0 is SH_C0.
template <int DEG>
BH_DEV void sh_basis(Vec3A v, float* b) {
    b[0] = 0.2820948f;
    if (DEG >= 1) {
        const float f0a = 0.4886025f;
        b[1] = -f0a * v.y;
        b[2] = f0a * v.z;
        b[3] = -f0a * v.x;
    }
    if (DEG >= 2) {
        const float z2 = v.z * v.z;
        const float f0b = -1.0925485f * v.z;
        const float f1a = 0.54627424f;
        const float fc1 = v.x * v.x - v.y * v.y;
        const float fs1 = 2.0f * v.x * v.y;
        b[4] = f1a * fs1;
        b[5] = f0b * v.y;
        b[6] = 0.9461747f * z2 - 0.31539157f;
        b[7] = f0b * v.x;
        b[8] = f1a * fc1;
        if (DEG >= 3) {
            const float f0c = -2.285229f * z2 + 0.4570458f;
            const float f1b = 1.4453057f * v.z;
            const float f2a = -0.5900436f;
            const float fc2 = v.x * fc1 - v.y * fs1;
            const float fs2 = v.x * fs1 + v.y * fc1;
            const float p12 = v.z * (1.8658817f * z2 - 1.119529f);
            b[9] = f2a * fs2;
            b[10] = f1b * fs1;
            b[11] = f0c * v.y;
            b[12] = p12;
            b[13] = f0c * v.x;
            b[14] = f1b * fc1;
            b[15] = f2a * fc2;
            if (DEG >= 4) {
                const float f0d = v.z * (-4.683326f * z2 + 2.0071396f);
                const float f1c = 3.3116114f * z2 - 0.47308735f;
                const float f2b = -1.7701308f * v.z;
                const float f3a = 0.62583575f;
                const float fc3 = v.x * fc2 - v.y * fs2;
                const float fs3 = v.x * fs2 + v.y * fc2;
                b[16] = f3a * fs3;
                b[17] = f2b * fs2;
                b[18] = f1c * fs1;
                b[19] = f0d * v.y;
                b[20] = 1.9843135f * v.z * p12 - 1.0062306f * b[6];
                b[21] = f0d * v.x;
                b[22] = f1c * fc1;
                b[23] = f2b * fc2;
                b[24] = f3a * fc3;
            }
        }
    }
}

// sh.rs:47-136: colour = sum_k coeff_k * basis_k, accumulated in index order.
template <int DEG>
BH_DEV Vec3A sh_coeffs_to_color(const float* __restrict__ c, Vec3A v) {
    constexpr int C = (DEG + 1) * (DEG + 1);
    float b[25];
    sh_basis<DEG>(v, b);
    Vec3A color = scale(Vec3A{c[0], c[1], c[2]}, b[0]);
#pragma unroll
    for (int k = 1; k < C; ++k) color = add(color, scale(Vec3A{c[3 * k], c[3 * k + 1], c[3 * k + 2]}, b[k]));
    return color;
}

// the same with the DC coefficient already in registers (K1 fetches it with the splat's other inputs)
template <int DEG>
BH_DEV Vec3A sh_coeffs_to_color_dc(const float* __restrict__ c, Vec3A v, const float (&dc)[3]) {
    constexpr int C = (DEG + 1) * (DEG + 1);
    float b[25];
    sh_basis<DEG>(v, b);
    Vec3A color = scale(Vec3A{dc[0], dc[1], dc[2]}, b[0]);
#pragma unroll
    for (int k = 1; k < C; ++k) color = add(color, scale(Vec3A{c[3 * k], c[3 * k + 1], c[3 * k + 2]}, b[k]));
    return color;
}

// sh.rs:277-355: v_coeff_k = vc * basis_k.
template <int DEG>
BH_DEV void sh_coeffs_to_color_vjp(float* __restrict__ vcoef, Vec3A v, Vec3A vc) {
    constexpr int C = (DEG + 1) * (DEG + 1);
    float b[25];
    sh_basis<DEG>(v, b);
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const Vec3A g = scale(vc, b[k]);
        vcoef[3 * k] = g.x;
        vcoef[3 * k + 1] = g.y;
        vcoef[3 * k + 2] = g.z;
    }
}

// sh.rs:143-271: dL/dv through the basis polynomials.
template <int DEG>
BH_DEV Vec3A sh_color_viewdir_vjp(const float* __restrict__ c, Vec3A v, Vec3A vc) {
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
    auto s = [&](int k) { return dot(Vec3A{c[3 * k], c[3 * k + 1], c[3 * k + 2]}, vc); };
    if (DEG >= 1) {
        const float f0a = 0.4886025f;
        {
            const float s_n1 = s(1), s_z0 = s(2), s_p1 = s(3);
            gx += -f0a * s_p1;
            gy += -f0a * s_n1;
            gz += f0a * s_z0;
        }
        if (DEG >= 2) {
            const float z = v.z, x = v.x, y = v.y;
            const float c2 = -1.0925485f;
            const float f1a = 0.54627424f;
            {
                const float s_n2 = s(4), s_n1 = s(5), s_z0 = s(6), s_p1 = s(7), s_p2 = s(8);
                gx += 2.0f * f1a * y * s_n2 + c2 * z * s_p1 + 2.0f * f1a * x * s_p2;
                gy += 2.0f * f1a * x * s_n2 + c2 * z * s_n1 - 2.0f * f1a * y * s_p2;
                gz += c2 * y * s_n1 + 2.0f * 0.9461747f * z * s_z0 + c2 * x * s_p1;
            }
            if (DEG >= 3) {
                const float z2 = z * z, x2 = x * x, y2 = y * y;
                const float f2a = -0.5900436f;
                const float c1b = 1.4453057f;
                const float f1b = c1b * z;
                const float c0c = -2.285229f;
                const float f0c = c0c * z2 + 0.4570458f;
                const float f0c_dz = 2.0f * c0c * z;
                {
                    const float s_n3 = s(9), s_n2 = s(10), s_n1 = s(11), s_z0 = s(12), s_p1 = s(13), s_p2 = s(14), s_p3 = s(15);
                    const float d12_z = 3.0f * 1.8658817f * z2 - 1.119529f;
                    gx += f2a * 6.0f * x * y * s_n3 + 2.0f * f1b * y * s_n2 + f0c * s_p1 + 2.0f * f1b * x * s_p2 + f2a * 3.0f * (x2 - y2) * s_p3;
                    gy += f2a * 3.0f * (x2 - y2) * s_n3 + 2.0f * f1b * x * s_n2 + f0c * s_n1 + (-2.0f) * f1b * y * s_p2 + f2a * (-6.0f) * x * y * s_p3;
                    gz += 2.0f * c1b * x * y * s_n2 + f0c_dz * y * s_n1 + d12_z * s_z0 + f0c_dz * x * s_p1 + c1b * (x2 - y2) * s_p2;
                }
                if (DEG >= 4) {
                    const float fc1 = x2 - y2;
                    const float fs1 = 2.0f * x * y;
                    const float fc2 = x * fc1 - y * fs1;
                    const float fs2 = x * fs1 + y * fc1;
                    const float f0d = z * (-4.683326f * z2 + 2.0071396f);
                    const float f0d_dz = -14.049978f * z2 + 2.0071396f;
                    const float f1c = 3.3116114f * z2 - 0.47308735f;
                    const float f1c_dz = 2.0f * 3.3116114f * z;
                    const float f2b_dz_const = -1.7701308f;
                    const float f2b = f2b_dz_const * z;
                    const float f3a = 0.62583575f;
                    const float p_sh12 = z * (1.8658817f * z2 - 1.119529f);
                    const float dp_sh12_dz = 3.0f * 1.8658817f * z2 - 1.119529f;
                    const float dp_sh6_dz = 2.0f * 0.9461747f * z;
                    const float dp_sh20_dz = 1.9843135f * (p_sh12 + z * dp_sh12_dz) - 1.0062306f * dp_sh6_dz;
                    const float s_n4 = s(16), s_n3 = s(17), s_n2 = s(18), s_n1 = s(19), s_z0 = s(20);
                    const float s_p1 = s(21), s_p2 = s(22), s_p3 = s(23), s_p4 = s(24);
                    gx += f3a * 4.0f * fs2 * s_n4 + f2b * 3.0f * fs1 * s_n3 + f1c * 2.0f * y * s_n2 + f0d * s_p1 + f1c * 2.0f * x * s_p2 + f2b * 3.0f * fc1 * s_p3 + f3a * 4.0f * fc2 * s_p4;
                    gy += f3a * 4.0f * fc2 * s_n4 + f2b * 3.0f * fc1 * s_n3 + f1c * 2.0f * x * s_n2 + f0d * s_n1 + f1c * (-2.0f) * y * s_p2 + f2b * (-3.0f) * fs1 * s_p3 + f3a * (-4.0f) * fs2 * s_p4;
                    gz += f2b_dz_const * fs2 * s_n3 + f1c_dz * fs1 * s_n2 + f0d_dz * y * s_n1 + dp_sh20_dz * s_z0 + f0d_dz * x * s_p1 + f1c_dz * fc1 * s_p2 + f2b_dz_const * fc2 * s_p3;
                }
            }
        }
    }
    return Vec3A{gx, gy, gz};
}

}  // namespace bh
