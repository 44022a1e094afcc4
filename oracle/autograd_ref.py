"""Independent float64 brute-force renderer with torch autograd — TEST INFRASTRUCTURE ONLY.

A second pin for the backward ("parity" item 1).  The C++ oracle (oracle/brush_oracle.cpp) restates
the reference's hand-written backward kernels line by line, and its gradients were so far validated only by its
own central finite differences.  This file shares NOTHING with it:

  * written from the rendering equations (what the reference computes), not from the
    reference's backward code: there is no hand-written VJP here at all — torch.autograd differentiates the
    forward below;
  * no tiling, no sorting network, no per-tile lists, no early-out batches: every pixel visits every splat in
    depth order (a tile list is a conservative superset of the splats that can reach a pixel with
    alpha >= 1/255, so the image is the same function);
  * float64 throughout.

So a misreading shared by the oracle's backward and the HIP kernels (both follow
bwd/kernels/rasterize_backwards.rs and project_backwards.rs) cannot hide here: if d(image)/d(params) of this
file agrees with the oracle's v_transforms / v_coeffs / v_raw_opac, the reference's backward computes the
derivative of the reference's forward, which is all a backward has to do.

The forward follows, in the reference's order of operations:
  kernels/project_forward.rs:44-111 + helpers.rs:145-195 (projection, cov2d, blur, conic, opacity),
  kernels/camera_model/pinhole.rs:25-57 (pinhole projection + clamped Jacobian),
  kernels/sh.rs:47-136 (real SH up to degree 3, Sloan 2013 constants),
  kernels/rasterize.rs:129-166 (front-to-back blend, alpha clamp 0.999, cutoff 1/255, T <= 1e-4 stop).
Default and Mip-Splatting mode (helpers.rs:180-195), hard and smooth alpha cutoff (helpers.rs:23-34).  Small scenes only
(O(pixels x splats)).

Lens models (kernels/camera_model/{kannala_brandt_4,radial_tangential_8,thin_prism_fisheye}.rs): only the PROJECTION
FUNCTIONS are written down here (project_kb4 :19-58, project_rt8 :23-62, project_tpf :64-82); the 2x3 Jacobian that carries the
covariance to the image is obtained by autograd from them (create_graph=True), so the reference's analytic Jacobians AND its
hand-written second-order terms (calculate_projection_vjp_*) are both checked against something that contains neither.
The radial-tangential model evaluates its Jacobian at the clamped normalised point like the pinhole one (…_rt8 :64-96); the two
fisheye models do not clamp (kannala_brandt_4.rs:60-62).  Intrinsics and the visibility cone (fx, fy, cx, cy, clamp limits,
half_max_render_fov) are taken from the oracle's camera set-up, which the ABI tests pin separately.
"""
import math

import numpy as np
import torch

SH_C0 = 0.2820947917738781
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def _quat_to_mat(q):
    """q [N,4] (w,x,y,z), unit -> rotation matrices [N,3,3]."""
    w, x, y, z = q.unbind(-1)
    r0 = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1)
    r1 = torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1)
    r2 = torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)
    return torch.stack([r0, r1, r2], -2)


def _sh_color(coeffs, d):
    """coeffs [N,C,3], d [N,3] unit view directions -> [N,3] (before the +0.5)."""
    n_coef = coeffs.shape[1]
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    c = SH_C0 * coeffs[:, 0]
    if n_coef > 1:
        c = c + SH_C1 * (-y * coeffs[:, 1] + z * coeffs[:, 2] - x * coeffs[:, 3])
    if n_coef > 4:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        c = c + (SH_C2[0] * xy * coeffs[:, 4] + SH_C2[1] * yz * coeffs[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * coeffs[:, 6]
                 + SH_C2[3] * xz * coeffs[:, 7] + SH_C2[4] * (xx - yy) * coeffs[:, 8])
        if n_coef > 9:
            c = c + (SH_C3[0] * y * (3.0 * xx - yy) * coeffs[:, 9] + SH_C3[1] * xy * z * coeffs[:, 10]
                     + SH_C3[2] * y * (4.0 * zz - xx - yy) * coeffs[:, 11]
                     + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * coeffs[:, 12]
                     + SH_C3[4] * x * (4.0 * zz - xx - yy) * coeffs[:, 13] + SH_C3[5] * z * (xx - yy) * coeffs[:, 14]
                     + SH_C3[6] * x * (xx - 3.0 * yy) * coeffs[:, 15])
    return c


def _project(model, dist, p, fx, fy, cx, cy):
    """p [N,3] camera-space points -> (u, v) [N] pixel coordinates, float64 torch ops only."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    if model == "pinhole":
        return fx * x / z + cx, fy * y / z + cy
    if model == "rt8":
        k1, k2, k3, k4, k5, k6, p1, p2 = [float(v) for v in dist]
        xn, yn = x / z, y / z
        r2 = xn * xn + yn * yn
        r4, r6 = r2 * r2, r2 * r2 * r2
        d = (1.0 + k1 * r2 + k2 * r4 + k3 * r6) / (1.0 + k4 * r2 + k5 * r4 + k6 * r6)
        xd = xn * d + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn)
        yd = yn * d + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn)
        return fx * xd + cx, fy * yd + cy
    k1, k2, k3, k4 = [float(v) for v in dist[:4]]
    r = torch.sqrt(x * x + y * y)
    th = torch.atan2(r, z)
    t2 = th * th
    d = th * (1.0 + k1 * t2 + k2 * t2 * t2 + k3 * t2 * t2 * t2 + k4 * t2 * t2 * t2 * t2)
    u, v = fx * d * x / r + cx, fy * d * y / r + cy
    if model == "kb4":
        return u, v
    assert model == "tpf"
    p1, p2, sx1, sy1 = [float(v) for v in dist[4:8]]
    r2 = x * x + y * y
    nu = 2.0 * p1 * x * y + p2 * (3.0 * x * x + y * y) + sx1 * r2
    nv = 2.0 * p2 * x * y + p1 * (x * x + 3.0 * y * y) + sy1 * r2
    return u + fx * nu / (z * z), v + fy * nv / (z * z)


def camera_matrices(pos, rot_xyzw, fov_x, fov_y, center_uv, w, h):
    """camera.rs:63-101,200-254 in float64: world->camera rotation [3,3] and translation [3], intrinsics, clamp limits."""
    x, y, z, ww = [float(v) for v in rot_xyzw]
    nrm = math.sqrt(x * x + y * y + z * z + ww * ww)
    x, y, z, ww = x / nrm, y / nrm, z / nrm, ww / nrm
    c2w = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - ww * z), 2 * (x * z + ww * y)],
                    [2 * (x * y + ww * z), 1 - 2 * (x * x + z * z), 2 * (y * z - ww * x)],
                    [2 * (x * z - ww * y), 2 * (y * z + ww * x), 1 - 2 * (x * x + y * y)]], np.float64)
    r = c2w.T
    t = -r @ np.asarray(pos, np.float64)
    fx = (w / 2.0) / math.tan(fov_x / 2.0)
    fy = (h / 2.0) / math.tan(fov_y / 2.0)
    cx, cy = center_uv[0] * w, center_uv[1] * h
    lim = ((1.15 * w - cx) / fx, (1.15 * h - cy) / fy, (-0.15 * w - cx) / fx, (-0.15 * h - cy) / fy)
    return r, t, (fx, fy, cx, cy), lim


def render(transforms, sh, raw_opac, cam, w, h, bg=(0.0, 0.0, 0.0), intrinsics=None, mip=False, smooth=False, comp_is_constant=True):
    """transforms [N,10] (mean, quat wxyz un-normalised, log-scale), sh [N,C,3], raw_opac [N]: float64 torch tensors
    (requires_grad as wanted); cam: dict(pos, rot_xyzw, fov_x, fov_y, center_uv[, model, dist]).  Returns the [h,w,4] image.
    intrinsics (lens models): dict(fx, fy, cx, cy, lim=(pos_x, pos_y, neg_x, neg_y), half_max_render_fov) from the camera set-up."""
    dt = torch.float64
    r_np, t_np, (fx, fy, cx, cy), lim = camera_matrices(cam["pos"], cam["rot_xyzw"], cam["fov_x"], cam["fov_y"], cam["center_uv"], w, h)
    model = cam.get("model", "pinhole")
    if model != "pinhole":
        fx, fy, cx, cy = intrinsics["fx"], intrinsics["fy"], intrinsics["cx"], intrinsics["cy"]
        lim = intrinsics["lim"]
    rc, tc = torch.tensor(r_np, dtype=dt), torch.tensor(t_np, dtype=dt)
    mean, quat, log_s = transforms[:, 0:3], transforms[:, 3:7], transforms[:, 7:10]
    n = transforms.shape[0]
    mean_c = mean @ rc.T + tc
    zc = mean_c[:, 2]
    if model == "pinhole":
        keep = (zc >= 0.01) & (zc <= 1e10)   # project_forward.rs:47-51 (the scenes used here keep every splat in front)
    else:                                    # :52-61: inside the render cone
        theta = torch.atan2(torch.sqrt(mean_c[:, 0] ** 2 + mean_c[:, 1] ** 2), zc)
        keep = (theta <= intrinsics["half_max_render_fov"]) & (zc <= 1e10)
    q = quat / quat.norm(dim=1, keepdim=True)
    m = _quat_to_mat(q) * torch.exp(log_s)[:, None, :]             # R(q) diag(s)
    cov_c = rc @ (m @ m.transpose(1, 2)) @ rc.T
    xz = torch.clamp(mean_c[:, 0] / zc, lim[2], lim[0])             # pinhole.rs:33-57: the Jacobian uses clamped x/z, y/z
    yz = torch.clamp(mean_c[:, 1] / zc, lim[3], lim[1])
    zero = torch.zeros_like(zc)
    if model == "pinhole":
        jac = torch.stack([torch.stack([fx / zc, zero, -fx / zc * xz], -1), torch.stack([zero, fy / zc, -fy / zc * yz], -1)], -2)
    else:
        # Jacobian of the projection by autograd, at the (rt8: clamped) camera-space point; create_graph keeps it differentiable
        q = torch.stack([xz * zc, yz * zc, zc], -1) if model == "rt8" else mean_c
        ju, jv = _project(model, cam["dist"], q, fx, fy, cx, cy)
        ru = torch.autograd.grad(ju.sum(), q, create_graph=True)[0]
        rv = torch.autograd.grad(jv.sum(), q, create_graph=True)[0]
        jac = torch.stack([ru, rv], -2)
    cov2 = jac @ cov_c @ jac.transpose(1, 2)
    # helpers.rs:180-195: + 0.3 I (comp = 1), or Mip-Splatting: + 0.1 I and opacity * sqrt(det raw / det blurred)
    blur = 0.1 if mip else 0.3
    det_raw = torch.clamp(cov2[:, 0, 0] * cov2[:, 1, 1] - cov2[:, 0, 1] * cov2[:, 0, 1], min=0.0)
    a, b, c = cov2[:, 0, 0] + blur, cov2[:, 0, 1], cov2[:, 1, 1] + blur
    det = a * c - b * b
    c00, c01, c11 = c / det, -b / det, a / det                          # conic = inverse
    mx, my = _project(model, cam.get("dist", ()), mean_c, fx, fy, cx, cy)
    alpha0 = torch.sigmoid(raw_opac)
    if mip:
        # The reference's backward treats the compensation factor as a CONSTANT of the geometry: project_backwards.rs:181-183
        # multiplies v_raw_opac by filter_comp, and :190-196 builds v_cov2d from the conic alone — there is no
        # d(filter_comp)/d(cov2d) term.  comp_is_constant=True reproduces that (a stop-gradient); False is the true derivative
        # of the forward, which the reference's gradients therefore are NOT in Mip mode (tests/test_oracle_autograd_pin.py).
        comp = torch.sqrt(det_raw / det)
        alpha0 = alpha0 * (comp.detach() if comp_is_constant else comp)
    cam_pos = torch.tensor(np.asarray(cam["pos"], np.float64), dtype=dt)
    vd = mean - cam_pos
    vd = vd / vd.norm(dim=1, keepdim=True)
    color = torch.clamp(_sh_color(sh, vd) + 0.5, -100.0, 100.0)         # project_visible.rs:56-71
    color = torch.clamp(color, min=0.0)                                 # rasterize.rs:147-149
    keep = keep & (alpha0 >= 1.0 / 255.0)

    py, px = torch.meshgrid(torch.arange(h, dtype=dt) + 0.5, torch.arange(w, dtype=dt) + 0.5, indexing="ij")
    T = torch.ones((h, w), dtype=dt)
    rgb = torch.zeros((h, w, 3), dtype=dt)
    done = torch.zeros((h, w), dtype=torch.bool)
    order = torch.argsort(zc.detach(), stable=True)
    for i in order.tolist():
        if not bool(keep[i]):
            continue
        dx, dy = px - mx[i], py - my[i]
        sigma = 0.5 * (c00[i] * dx * dx + c11[i] * dy * dy) + c01[i] * dx * dy
        alpha = torch.clamp(alpha0[i] * torch.exp(-sigma), max=0.999)
        if smooth:   # helpers.rs:23-34: smoothstep over [1/255 - 5e-4, 1/255 + 5e-4] instead of the step at 1/255
            tt = torch.clamp((alpha - (1.0 / 255.0 - 0.5e-3)) / 1.0e-3, 0.0, 1.0)
            w_cut = tt * tt * (3.0 - 2.0 * tt)
            ok = (sigma >= 0) & (w_cut > 0) & ~done
            alpha = alpha * w_cut
        else:
            ok = (sigma >= 0) & (alpha >= 1.0 / 255.0) & ~done
        next_t = T * (1.0 - alpha)
        sat = ok & (next_t <= 1e-4)                 # rasterize.rs:155-160: the pixel is done WITHOUT adding this splat
        contrib = ok & ~sat
        vis = torch.where(contrib, alpha * T, torch.zeros_like(T))
        rgb = rgb + vis[..., None] * color[i]
        T = torch.where(contrib, next_t, T)
        done = done | sat
    bgt = torch.tensor(bg, dtype=dt)
    return torch.cat([rgb + T[..., None] * bgt, (1.0 - T)[..., None]], dim=-1)


def gradients(scene, cam, w, h, weights, bg=(0.0, 0.0, 0.0), intrinsics=None, mip=False, smooth=False, comp_is_constant=True):
    """d( sum(weights * image) ) / d(transforms, sh, raw_opac) by autograd; numpy float64 in and out."""
    tr = torch.tensor(np.asarray(scene["transforms"], np.float64), requires_grad=True)
    sh = torch.tensor(np.asarray(scene["sh"], np.float64), requires_grad=True)
    op = torch.tensor(np.asarray(scene["raw_opac"], np.float64), requires_grad=True)
    img = render(tr, sh, op, cam, w, h, bg, intrinsics, mip, smooth, comp_is_constant)
    loss = (img * torch.tensor(np.asarray(weights, np.float64))).sum()
    loss.backward()
    return img.detach().numpy(), tr.grad.numpy(), sh.grad.numpy(), op.grad.numpy()
