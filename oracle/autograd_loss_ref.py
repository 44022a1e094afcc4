"""Independent float64 restatement of the image loss with torch autograd — TEST INFRASTRUCTURE ONLY.

Second pin for the loss backward (brush-loss/src/lib.rs:371-661, a hand-written kernel that re-derives the SSIM partials on an
apron): the oracle restates that kernel; this file contains no backward at all.  The forward is written from the definition
the reference's forward kernel implements (lib.rs:181-359): per colour channel
    loss = l1_w * |x - y| + ssim_w * clamp(SSIM(x, y), -1, 1)          (times gt alpha when masking)
with SSIM over an 11-tap Gaussian window (sigma 1.5, normalised, lib.rs:55-68), zero padding (lib.rs:110-176), the variances
clamped at zero, C1 = 0.01^2, C2 = 0.03^2; y = gt colour (+ (1 - gt alpha) * background when compositing); the alpha channel
(c == 3) is |x_a - gt_a| without blur (lib.rs:200-212).  torch.nn.functional.conv2d does the blurs."""
import numpy as np
import torch
import torch.nn.functional as F

C1, C2 = 0.01 ** 2, 0.03 ** 2


def _window():
    x = torch.arange(11, dtype=torch.float64) - 5.0
    g = torch.exp(-x * x / (2.0 * 1.5 * 1.5))
    return g / g.sum()


def _blur(img):
    """img [C,H,W] -> separable 11-tap Gaussian, zero padded."""
    g = _window()
    c = img.shape[0]
    x = img[None]
    x = F.conv2d(x, g.view(1, 1, 1, 11).repeat(c, 1, 1, 1), padding=(0, 5), groups=c)
    x = F.conv2d(x, g.view(1, 1, 11, 1).repeat(c, 1, 1, 1), padding=(5, 0), groups=c)
    return x[0]


def unpack_gt(gt_packed):
    g = np.asarray(gt_packed, np.uint32)
    return np.stack([((g >> (8 * k)) & 0xFF).astype(np.float64) / 255.0 for k in range(4)], 0)   # [4,H,W] r g b a


def loss_map(pred, gt_packed, l1_w, ssim_w, bg=None, mask=False):
    """pred [C,H,W] float64 torch tensor (C = 3 or 4) -> loss map [C,H,W]."""
    gt = torch.tensor(unpack_gt(gt_packed))
    ga = gt[3]
    y = gt[:3]
    if bg is not None:
        y = y + (1.0 - ga)[None] * torch.tensor(np.asarray(bg, np.float64)).view(3, 1, 1)
    x = pred[:3]
    mu1, mu2 = _blur(x), _blur(y)
    s1 = torch.clamp(_blur(x * x) - mu1 * mu1, min=0.0)
    s2 = torch.clamp(_blur(y * y) - mu2 * mu2, min=0.0)
    s12 = _blur(x * y) - mu1 * mu2
    ssim = torch.clamp(((2.0 * mu1 * mu2 + C1) * (2.0 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2)), -1.0, 1.0)
    out = l1_w * torch.abs(x - y) + ssim_w * ssim
    if mask:
        out = out * ga[None]
    if pred.shape[0] == 4:
        a = torch.abs(pred[3] - ga)
        if mask:
            a = a * ga
        out = torch.cat([out, a[None]], 0)
    return out


def backward(pred_chw, gt_packed, dl_dmap, l1_w, ssim_w, bg=None, mask=False):
    """-> (loss map, d(sum(dl_dmap * map)) / d pred) as float64 numpy arrays."""
    p = torch.tensor(np.asarray(pred_chw, np.float64), requires_grad=True)
    m = loss_map(p, gt_packed, l1_w, ssim_w, bg, mask)
    (m * torch.tensor(np.asarray(dl_dmap, np.float64))).sum().backward()
    return m.detach().numpy(), p.grad.numpy()
