"""Oracle restatement (numpy) of view_to_packed_data / pack_rgba, brush-dataset/src/scene.rs:97-136.
TEST INFRASTRUCTURE ONLY."""
import numpy as np


def view_to_packed_data(img_u8, transparent_alpha=True):
    """img_u8 [H,W,3] or [H,W,4] uint8 -> (packed [H,W] uint32 little-endian r g b a, has_alpha).
    RGB views get a = 255; RGBA views with AlphaMode::Transparent are premultiplied in byte space,
    mul(c) = (c * a + 127) / 255 (integer); AlphaMode::Masked keeps the colours."""
    a = np.asarray(img_u8)
    assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] in (3, 4)
    has_alpha = a.shape[2] == 4
    v = a.astype(np.uint32)
    if not has_alpha:
        return (v[..., 0] | (v[..., 1] << 8) | (v[..., 2] << 16) | np.uint32(255 << 24)).astype(np.uint32), False
    al = v[..., 3]
    if transparent_alpha:
        rgb = [(v[..., c] * al + 127) // 255 for c in range(3)]
    else:
        rgb = [v[..., c] for c in range(3)]
    return (rgb[0] | (rgb[1] << 8) | (rgb[2] << 16) | (al << 24)).astype(np.uint32), True


def nerfstudio_frame_to_camera(transform_matrix, intrinsics):
    """One `frames[]` entry of a nerfstudio / Blender transforms.json -> the Camera arguments Brush builds from it
    (brush-dataset/src/formats/nerfstudio.rs:156-258, formats/mod.rs:123-131).  transform_matrix [4,4] rows as written in the
    file (camera-to-world, OpenGL axes: +X right, +Y up, +Z back); intrinsics = (fl_x, fl_y, cx, cy, w, h, camera_angle_x,
    camera_angle_y).  Returns dict(pos, rot_xyzw, fov_x, fov_y, center_uv, img_w, img_h) in Brush's convention (+Y down, +Z fwd).
    f32 arithmetic as glam does it (Mat4::to_scale_rotation_translation + Quat::from_rotation_axes)."""
    m = np.asarray(transform_matrix, np.float32).reshape(4, 4)
    fl_x, fl_y, cx, cy, w, h, ang_x, ang_y = [float(v) for v in intrinsics]
    f = np.float32
    # from_cols_slice(flattened rows).transpose() == the matrix as written; column k = m[:, k]
    x_axis, y_axis, z_axis, w_axis = m[:3, 0].copy(), -m[:3, 1], -m[:3, 2], m[:3, 3].copy()   # opengl_c2w_to_pose: y, z columns negated
    det = float(np.linalg.det(np.stack([x_axis, y_axis, z_axis], axis=1).astype(np.float64)))
    sx = f(np.sqrt(f(np.dot(x_axis, x_axis)))) * f(1.0 if det >= 0 else -1.0)
    sy, sz = f(np.sqrt(f(np.dot(y_axis, y_axis)))), f(np.sqrt(f(np.dot(z_axis, z_axis))))
    ax, ay, az = x_axis / sx, y_axis / sy, z_axis / sz
    m00, m01, m02 = ax
    m10, m11, m12 = ay
    m20, m21, m22 = az
    half = f(0.5)
    if m22 <= 0:
        dif10, omm22 = m11 - m00, f(1.0) - m22
        if dif10 <= 0:
            four = omm22 - dif10
            inv = half / f(np.sqrt(four))
            q = (four * inv, (m01 + m10) * inv, (m02 + m20) * inv, (m12 - m21) * inv)
        else:
            four = omm22 + dif10
            inv = half / f(np.sqrt(four))
            q = ((m01 + m10) * inv, four * inv, (m12 + m21) * inv, (m20 - m02) * inv)
    else:
        sum10, opm22 = m11 + m00, f(1.0) + m22
        if sum10 <= 0:
            four = opm22 - sum10
            inv = half / f(np.sqrt(four))
            q = ((m02 + m20) * inv, (m12 + m21) * inv, four * inv, (m01 - m10) * inv)
        else:
            four = opm22 + sum10
            inv = half / f(np.sqrt(four))
            q = ((m12 - m21) * inv, (m20 - m02) * inv, (m01 - m10) * inv, four * inv)
    # nerfstudio.rs:218-243: the frame's camera_angle_* win over its focal lengths
    return dict(pos=tuple(float(v) for v in w_axis), rot_xyzw=tuple(float(v) for v in q), fov_x=ang_x, fov_y=ang_y,
                center_uv=(float(f(cx / w)), float(f(cy / h))), img_w=int(w), img_h=int(h))
