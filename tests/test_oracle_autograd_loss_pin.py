"""Second, independent pin of the loss backward: the oracle's restatement of the reference's hand-written backward kernel
(brush-loss/src/lib.rs:371-661) against torch.autograd through a float64 forward written from the SSIM definition
(oracle/autograd_loss_ref.py: conv2d blurs, no backward code).  Until now the loss backward was pinned by the oracle's own
finite differences only (tests/test_oracle_properties.py)."""
import numpy as np
import pytest

from oracle import autograd_loss_ref, bo


def _case(seed, c, h, w, smooth=True):
    rng = np.random.default_rng(seed)
    if smooth:   # correlated images: SSIM away from its clamps, like a render against its photo
        base = rng.uniform(0.1, 0.9, (4, h // 4 + 2, w // 4 + 2))
        up = np.kron(base, np.ones((4, 4)))[:, :h, :w]
        gt = np.clip(up + rng.normal(0, 0.02, up.shape), 0, 1)
        pred = np.clip(up + rng.normal(0, 0.08, up.shape), 0, 1)[:c]
    else:
        gt = rng.uniform(0, 1, (4, h, w))
        pred = rng.uniform(0, 1, (c, h, w))
    g8 = np.round(gt * 255).astype(np.uint32)
    packed = g8[0] | (g8[1] << 8) | (g8[2] << 16) | (g8[3] << 24)
    dl = rng.uniform(0.2, 1.0, (c, h, w)) / (c * h * w)
    return pred.astype(np.float32), packed, dl.astype(np.float32)


@pytest.mark.parametrize("seed,c,h,w,bg,mask,smooth", [
    (1, 3, 40, 56, None, False, True), (2, 4, 33, 47, None, False, True), (3, 3, 24, 24, (0.2, 0.5, 0.1), False, True),
    (4, 4, 37, 29, (0.9, 0.1, 0.3), True, True), (5, 3, 16, 70, None, True, False), (6, 4, 50, 18, (0.0, 0.0, 0.0), False, False),
    (7, 3, 11, 11, None, False, True), (8, 3, 5, 9, None, False, False)])
def test_loss_forward_and_backward_match_autograd(seed, c, h, w, bg, mask, smooth):
    pred, packed, dl = _case(seed, c, h, w, smooth)
    l1_w, ssim_w = 0.8, -0.2   # the train step's weights: (1 - w) * L1 + w * (1 - SSIM) up to the constant (train.rs:227-260)
    want_map, want_grad = autograd_loss_ref.backward(pred, packed, dl, l1_w, ssim_w, bg, mask)
    got_map = bo.image_loss_forward(pred, packed, l1_w, ssim_w, bg, mask).astype(np.float64)
    got_grad = bo.image_loss_backward(pred, packed, dl, l1_w, ssim_w, bg, mask).astype(np.float64)
    assert np.abs(got_map - want_map).max() <= 1e-5, "loss map"   # f32 cancellation in E[x^2] - mu^2 against C2 = 9e-4: ~3e-6 measured
    d, ref = np.abs(got_grad - want_grad).max(), np.abs(want_grad).max()
    assert ref > 0 and d <= 2e-5 * ref + 1e-10, (d, ref)
