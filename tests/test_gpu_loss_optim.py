"""Image loss (fwd/bwd), AdamScaled and refine statistics through the C ABI vs the oracle."""
import numpy as np
import pytest
import torch

import util

pytestmark = pytest.mark.gpu


def _gt(rng, h, w):
    return util.packed_from_rgba(rng.integers(0, 256, (h, w, 4), dtype=np.uint32))


@pytest.mark.parametrize("h,w,ch", [(40, 52, 3), (16, 16, 3), (1, 1, 3), (67, 131, 4), (270, 480, 3)])
@pytest.mark.parametrize("bg,mask", [(None, False), ((0.3, 0.5, 0.2), False), (None, True)])
def test_image_loss_forward_backward_vs_oracle(dev, oracle_lib, h, w, ch, bg, mask):
    """brush-loss/src/lib.rs:181-661: same tap-pair accumulation order -> compared at 1e-6."""
    import brush_amd as ba
    rng = np.random.default_rng(h * w + ch)
    gt = _gt(rng, h, w)
    pred = rng.uniform(-0.1, 1.1, (h, w, ch)).astype(np.float32)
    dl = rng.uniform(-1, 1, (h, w, ch)).astype(np.float32)
    gt_t = torch.from_numpy(gt.view(np.int32)).to(dev)
    lm = ba.image_loss(torch.from_numpy(pred).to(dev), gt_t, 0.8, -0.2, composite_bg=bg, mask=mask).cpu().numpy()
    g = ba.image_loss_backward(torch.from_numpy(pred).to(dev), gt_t, torch.from_numpy(dl).to(dev), 0.8, -0.2, composite_bg=bg, mask=mask).cpu().numpy()
    pc = np.ascontiguousarray(pred.transpose(2, 0, 1))
    rlm = oracle_lib.image_loss_forward(pc, gt, 0.8, -0.2, bg=bg, mask=mask).transpose(1, 2, 0)
    rg = oracle_lib.image_loss_backward(pc, gt, dl.transpose(2, 0, 1), 0.8, -0.2, bg=bg, mask=mask).transpose(1, 2, 0)
    assert np.abs(lm - rlm).max() <= 2e-6
    assert np.abs(g - rg).max() <= 2e-6 * max(1.0, np.abs(rg).max())


@pytest.mark.parametrize("h,w", [(40, 52), (16, 16), (1, 1), (67, 131), (270, 480)])
@pytest.mark.parametrize("bg,mask,alpha_w", [(None, False, 0.0), ((0.3, 0.5, 0.2), False, 0.1), (None, True, 0.0)])
def test_fused_loss_matches_oracle_and_standalone(dev, oracle_lib, h, w, bg, mask, alpha_w):
    """bh_image_loss_value_and_grad (what bh_train_step runs) == mean(loss map) and its gradient
    from the oracle's stand-alone forward/backward (train.rs:227-260)."""
    import brush_amd as ba
    rng = np.random.default_rng(h * 7 + w)
    gt = _gt(rng, h, w)
    img = rng.uniform(-0.1, 1.1, (h, w, 4)).astype(np.float32)
    ch = 4 if alpha_w > 0 else 3
    gt_t = torch.from_numpy(gt.view(np.int32)).to(dev)
    loss, v_out = ba.image_loss_value_and_grad(torch.from_numpy(img).to(dev), gt_t, 0.8, -0.2, composite_bg=bg, mask=mask, alpha_weight=alpha_w)
    pc = np.ascontiguousarray(img[..., :ch].transpose(2, 0, 1))
    rlm = oracle_lib.image_loss_forward(pc, gt, 0.8, -0.2, bg=bg, mask=mask).astype(np.float64)
    ref_loss = rlm[:3].mean() + (alpha_w * rlm[3].mean() if ch == 4 else 0.0)
    dl = np.empty((ch, h, w), np.float32)
    dl[:3] = 1.0 / (h * w * 3)
    if ch == 4:
        dl[3] = alpha_w / (h * w)
    rg = oracle_lib.image_loss_backward(pc, gt, dl, 0.8, -0.2, bg=bg, mask=mask).transpose(1, 2, 0)
    assert abs(float(loss.item()) - ref_loss) <= 2e-6 * max(1.0, abs(ref_loss))
    g = v_out.cpu().numpy()
    assert np.abs(g[..., :ch] - rg).max() <= 2e-6 * max(np.abs(rg).max(), 1e-12)
    if ch == 3:
        assert not g[..., 3].any()
    # and against the stand-alone HIP kernels (same tap order)
    g2 = ba.image_loss_backward(torch.from_numpy(img[..., :ch].copy()).to(dev), gt_t, torch.from_numpy(np.ascontiguousarray(dl.transpose(1, 2, 0))).to(dev),
                                0.8, -0.2, composite_bg=bg, mask=mask).cpu().numpy()
    assert np.abs(g[..., :ch] - g2).max() <= 2e-6 * max(np.abs(g2).max(), 1e-12)   # fused kernels contract to FMA / use v_rcp


def test_ssim_of_identical_images_is_one(dev):
    """brush-loss/tests/reference.rs:57"""
    import brush_amd as ba
    rng = np.random.default_rng(0)
    rgba = rng.integers(0, 256, (48, 64, 4), dtype=np.uint32)
    gt = torch.from_numpy(util.packed_from_rgba(rgba).view(np.int32)).to(dev)
    pred = torch.from_numpy((rgba[..., :3] / 255.0).astype(np.float32)).to(dev)
    lm = ba.image_loss(pred, gt, 0.0, 1.0)
    assert float((lm - 1.0).abs().max()) < 1e-5


@pytest.mark.parametrize("rows,row_len,reduce_m2", [(1000, 10, False), (777, 48, True), (5000, 3, True), (4097, 1, False)])
def test_adam_step_bit_exact_vs_oracle(dev, oracle_lib, rows, row_len, reduce_m2):
    """adam_scaled.rs:75-147: identical operation sequence -> bit-exact over 4 steps."""
    import brush_amd as ba
    rng = np.random.default_rng(rows)
    p = rng.normal(size=(rows, row_len)).astype(np.float32)
    m1 = np.zeros_like(p)
    m2 = np.zeros(rows if reduce_m2 else (rows, row_len), np.float32)
    scale = rng.uniform(0.1, 1.0, row_len).astype(np.float32)
    tp, tm1, tm2, ts = (torch.from_numpy(a.copy()).to(dev) for a in (p, m1, m2, scale))
    for t in range(1, 5):
        g = (rng.normal(size=(rows, row_len)) * 10.0 ** rng.integers(-6, 1)).astype(np.float32)
        oracle_lib.adam_step(p, g, m1, m2, 0.01, t, col_scale=scale, reduce_m2=reduce_m2)
        ba.adam_step(tp, torch.from_numpy(g).to(dev), tm1, tm2, 0.01, t, col_scale=ts, reduce_m2=reduce_m2)
        assert np.array_equal(tp.cpu().numpy(), p), "step %d" % t
        assert np.array_equal(tm1.cpu().numpy(), m1) and np.array_equal(tm2.cpu().numpy(), m2)


def test_gather_stats(dev, oracle_lib):
    """stats.rs:40-50"""
    import ctypes as C
    import brush_amd as ba
    rng = np.random.default_rng(3)
    n = 10001
    a = [rng.uniform(0, 1, n).astype(np.float32) for _ in range(6)]
    t = [torch.from_numpy(x.copy()).to(dev) for x in a]
    ctx = ba.get_context(dev)
    ctx.check(ctx.lib.bh_gather_stats(ctx._h, *[C.c_void_p(x.data_ptr()) for x in t], n))
    oracle_lib.lib().bo_gather_stats(*[oracle_lib._fp(x) for x in a], n)
    for x, y in zip(t[:3], a[:3]):
        assert np.array_equal(x.cpu().numpy(), y)


def test_gather_stats_is_the_running_max_sum_max(dev):
    """RefineRecord::gather_stats (brush-train/src/stats.rs:40-50) through the Python mirror: exact (max / add / max per splat)."""
    import brush_amd as ba
    rng = np.random.default_rng(11)
    n = 100003
    a, b, c = rng.random(n, dtype=np.float32), rng.integers(0, 5, n).astype(np.float32), rng.uniform(0, 50, n).astype(np.float32)
    rw, vis, rad = rng.random(n, dtype=np.float32), rng.integers(0, 2, n).astype(np.float32), rng.uniform(0, 60, n).astype(np.float32)
    ta, tb, tc = (torch.from_numpy(x.copy()).to(dev) for x in (a, b, c))
    for _ in range(2):
        ba.gather_stats(ta, tb, tc, torch.from_numpy(rw).to(dev), torch.from_numpy(vis).to(dev), torch.from_numpy(rad).to(dev))
        a, b, c = np.maximum(a, rw), b + vis, np.maximum(c, rad)
    ba.get_context(dev).sync()
    assert np.array_equal(ta.cpu().numpy(), a) and np.array_equal(tb.cpu().numpy(), b) and np.array_equal(tc.cpu().numpy(), c)
