"""Oracle composition of SplatTrainer::step (brush-train/src/train.rs:176-429) out of the
C oracle's pieces.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline)."""
import numpy as np

# ---------------------------------------------------------------------------
# oracle composition of SplatTrainer::step (brush-train/src/train.rs:176-429)
# ---------------------------------------------------------------------------
class OracleTrainer:
    """CPU restatement of step(): oracle forward -> L1+SSIM loss (mean) -> loss backward ->
    render backward -> RefineRecord stats -> AdamScaled x3 -> optional mean noise."""

    def __init__(self, bo, cfg, median_scene_scale=1.0):
        self.bo, self.cfg, self.median = bo, cfg, float(median_scene_scale)
        self.step_count = 0
        self.state = None

    def step(self, scene, cam, gt_packed, background, has_alpha=False, alpha_is_mask=False, noise=None,
             extra_grads=None, world=1, dry_run=False, min_scale=None):
        bo, c = self.bo, self.cfg
        tr, sh, op = scene["transforms"], scene["sh"], scene["raw_opac"]
        n, C = tr.shape[0], sh.shape[1]
        if self.state is None:
            z = lambda *s: np.zeros(s, np.float32)  # noqa: E731
            self.state = dict(m1_t=z(n, 10), m2_t=z(n, 10), m1_sh=z(n, C * 3), m2_sh=z(n), m1_o=z(n, 1), m2_o=z(n, 1),
                              refine=z(n), vis=z(n), screen=z(n))
        self.step_count += 1
        h, w = cam.img_h, cam.img_w
        flags = bo.FLAG_BWD_INFO | (bo.FLAG_MIP if c.render_mip else 0)
        # Mip-Splatting 3D filter: the renderer sees fold_min_scale(params) (bwd/burn_glue.rs:260-270)
        r_tr, r_op = (tr, op) if min_scale is None else bo.fold_min_scale(tr, op, min_scale)
        R = bo.Render().forward(cam, r_tr, sh, r_op, bg=background, flags=flags)
        img = R.image()
        ssim_on = c.ssim_weight > 0
        l1_w, ssim_w = (1.0 - c.ssim_weight, -c.ssim_weight) if ssim_on else (1.0, 0.0)
        alpha_match = has_alpha and not alpha_is_mask and c.match_alpha_weight > 0
        ch = 4 if alpha_match else 3
        bgc = tuple(background) if (has_alpha and any(b != 0 for b in background)) else None
        pred = np.ascontiguousarray(img[..., :ch].transpose(2, 0, 1))
        lm = bo.image_loss_forward(pred, gt_packed, l1_w, ssim_w, bg=bgc, mask=alpha_is_mask)
        hw = h * w
        dl_rgb = np.float32(1.0) / np.float32(hw * 3)
        dl = np.full((ch, h, w), dl_rgb, np.float32)
        loss = float(lm[:3].astype(np.float64).sum() * dl_rgb)
        if alpha_match:
            dl_a = np.float32(c.match_alpha_weight) / np.float32(hw)
            dl[3] = dl_a
            loss += float(lm[3].astype(np.float64).sum() * dl_a)
        g = bo.image_loss_backward(pred, gt_packed, dl, l1_w, ssim_w, bg=bgc, mask=alpha_is_mask)
        v_out = np.zeros((h, w, 4), np.float32)
        v_out[..., :ch] = g.transpose(1, 2, 0)
        R.backward(v_out)
        g_tr = R.get("v_transforms").reshape(n, 10).copy()
        g_sh = R.get("v_coeffs").reshape(n, C * 3).copy()
        g_op = R.get("v_raw_opac").reshape(n, 1).copy()
        if min_scale is not None:  # chain through the fold
            g_tr, g_o1 = bo.fold_min_scale_backward(tr, op, min_scale, g_tr, g_op.reshape(n))
            g_op = g_o1.reshape(n, 1)
        refine, vis, radius = R.get("v_refine").copy(), R.get("visible").copy(), R.get("max_radius").copy()
        if dry_run:  # this rank's raw gradients only (data-parallel tests); no state change
            self.step_count -= 1
            return dict(g_tr=g_tr, g_sh=g_sh, g_op=g_op, refine=refine, vis=vis, radius=radius)
        if extra_grads is not None:  # data-parallel over cameras: gradients and visible flags are summed over
            # ranks (vis_weight counts views), the gradients then scaled by 1/world; refine weight / radius feed running maxima
            for eg in extra_grads:
                g_tr += eg["g_tr"]; g_sh += eg["g_sh"]; g_op += eg["g_op"]
                refine = np.maximum(refine, eg["refine"]); vis = vis + eg["vis"]; radius = np.maximum(radius, eg["radius"])
        if world != 1:
            s = np.float32(1.0 / world)
            g_tr *= s; g_sh *= s; g_op *= s
        st = self.state
        bo.lib().bo_gather_stats(bo._fp(st["refine"]), bo._fp(st["vis"]), bo._fp(st["screen"]), bo._fp(refine), bo._fp(vis), bo._fp(radius), n)
        decay = (c.lr_mean_end / c.lr_mean) ** (1.0 / c.total_train_iters)
        lr_mean = c.lr_mean * decay ** (self.step_count - 1) * self.median
        lrs = np.array([lr_mean] * 3 + [c.lr_rotation] * 4 + [c.lr_scale] * 3, np.float32)
        t = self.step_count
        bo.adam_step(tr, g_tr, st["m1_t"], st["m2_t"], 1.0, t, col_scale=lrs)
        rest = np.float32(1.0) / np.float32(c.lr_coeffs_sh_scale)
        sh_scale = np.array([1.0 if k // 3 == 0 else rest for k in range(3 * C)], np.float32)
        sh2 = sh.reshape(n, C * 3)
        bo.adam_step(sh2, g_sh, st["m1_sh"], st["m2_sh"], np.float32(c.lr_coeffs_dc), t, col_scale=sh_scale, reduce_m2=True)
        op2 = op.reshape(n, 1)
        bo.adam_step(op2, g_op, st["m1_o"], st["m2_o"], np.float32(c.lr_opac), t)
        if noise is not None and c.mean_noise_weight > 0:
            gate_op = op if min_scale is None else bo.fold_min_scale(tr, op, min_scale)[1]  # splats.opacities(), train.rs:389
            sig = 1.0 / (1.0 + np.exp(-gate_op.astype(np.float64)))
            wgt = np.clip((1.0 - sig) ** 150, 0, 1) * np.minimum(vis, 1.0)
            wm = (wgt * (np.float32(lr_mean) * c.mean_noise_weight)).astype(np.float32)
            tr[:, :3] += np.clip(noise * wm[:, None], -self.median, self.median).astype(np.float32)
        grads = dict(g_tr=g_tr, g_sh=g_sh, g_op=g_op, refine=refine, vis=vis, radius=radius)
        return dict(loss=loss, num_visible=R.num_visible, num_intersections=R.num_intersections, lr_mean=lr_mean, grads=grads, img=img)


