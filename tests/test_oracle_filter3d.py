"""Mip-Splatting 3D filter: fold_min_scale (gaussian_splats.rs:86-111), its VJP and
compute_min_scale (train.rs:102-125).  The reference has no golden numbers for these (they are burn
tensor expressions); the oracle restatement is pinned against an independent float64 numpy
restatement of the same formulas and its hand-derived VJP against central differences of it."""
import numpy as np

from oracle import bo


def _scene(n, seed=0):
    rng = np.random.default_rng(seed)
    tr = rng.uniform(-1, 1, (n, 10)).astype(np.float32)
    tr[:, 7:] = rng.uniform(-6, -1, (n, 3))
    return tr, rng.uniform(-3, 4, n).astype(np.float32), rng.uniform(0.0005, 0.05, n).astype(np.float32)


def _fold64(tr, op, f):
    a = np.exp(2 * tr[:, 7:].astype(np.float64))
    b = a + (f.astype(np.float64) ** 2)[:, None]
    coef = np.sqrt(a.prod(1) / b.prod(1))
    o = np.clip(1 / (1 + np.exp(-op.astype(np.float64))) * coef, 1e-6, 1 - 1e-6)
    return 0.5 * np.log(b), np.log(o / (1 - o))


def test_fold_matches_float64_restatement():
    tr, op, f = _scene(500)
    ft, fo = bo.fold_min_scale(tr, op, f)
    nl, ro = _fold64(tr, op, f)
    assert np.array_equal(ft[:, :7], tr[:, :7])          # means / rotations untouched
    assert np.abs(ft[:, 7:] - nl).max() < 2e-6
    assert np.abs(fo - ro).max() < 5e-6
    assert (ft[:, 7:] >= tr[:, 7:] - 1e-6).all()          # the floor only ever inflates
    assert (fo <= op + 1e-5).all()                        # and only ever removes opacity
    assert np.abs(np.exp(ft[:, 7:]) - np.sqrt(np.exp(2 * tr[:, 7:].astype(np.float64)) + (f ** 2)[:, None])).max() < 1e-6


def test_zero_floor_is_identity_up_to_the_clamp():
    tr, op, _ = _scene(200, 1)
    ft, fo = bo.fold_min_scale(tr, op, np.zeros(200, np.float32))
    assert np.abs(ft - tr).max() < 1e-6 and np.abs(fo - op).max() < 2e-5


def test_fold_is_idempotent_under_bake():
    """bake_min_scale (gaussian_splats.rs:245-256): folding in place then rendering without a floor
    is the same splat; a second fold with f = 0 does not move it."""
    tr, op, f = _scene(300, 2)
    ft, fo = bo.fold_min_scale(tr, op, f)
    ft2, fo2 = bo.fold_min_scale(ft, fo, np.zeros(300, np.float32))
    assert np.abs(ft2 - ft).max() < 1e-6 and np.abs(fo2 - fo).max() < 2e-5


def test_fold_backward_matches_central_differences():
    n = 300
    tr, op, f = _scene(n, 3)
    rng = np.random.default_rng(4)
    gl, gr = rng.normal(size=(n, 3)), rng.normal(size=n)
    vt = rng.normal(size=(n, 10)).astype(np.float32)
    vt[:, 7:] = gl
    bt, bop = bo.fold_min_scale_backward(tr, op, f, vt, gr.astype(np.float32))
    assert np.array_equal(bt[:, :7], vt[:, :7])           # the other columns pass through
    eps = 1e-6

    def loss(t64, o64):
        nl, ro = _fold64(t64, o64, f)
        return (nl * gl).sum(1) + ro * gr
    t64, o64 = tr.astype(np.float64), op.astype(np.float64)
    for k in range(3):
        tp, tm = t64.copy(), t64.copy()
        tp[:, 7 + k] += eps
        tm[:, 7 + k] -= eps
        num = (loss(tp, o64) - loss(tm, o64)) / (2 * eps)
        assert np.abs(bt[:, 7 + k] - num).max() <= 2e-5 * max(1.0, np.abs(num).max())
    num = (loss(t64, o64 + eps) - loss(t64, o64 - eps)) / (2 * eps)
    assert np.abs(bop - num).max() <= 2e-5 * max(1.0, np.abs(num).max())


def test_clamped_opacity_gets_no_gradient():
    tr = np.zeros((2, 10), np.float32)
    tr[:, 7:] = -9.0                                     # tiny splats, large floor -> coef ~ 0 -> clamp at 1e-6
    op = np.array([0.0, 30.0], np.float32)
    f = np.array([5.0, 0.0], np.float32)                 # second: sigmoid(30) = 1 -> clamp at 1 - 1e-6
    vt = np.ones((2, 10), np.float32)
    bt, bop = bo.fold_min_scale_backward(tr, op, f, vt, np.ones(2, np.float32))
    assert bop[0] == 0.0 and bop[1] == 0.0
    assert np.isfinite(bt).all()


def test_compute_min_scale_matches_numpy():
    tr, _, _ = _scene(400, 5)
    cams = np.array([[0, 0, -3, 800.0], [2, 1, -2, 650.0], [-1, 0.5, 4, 0.0]], np.float32)  # focal 0 -> max(focal, 1e-6)
    got = bo.compute_min_scale(tr, cams, 0.1)
    d = np.linalg.norm(tr[:, None, :3].astype(np.float64) - cams[None, :, :3], axis=2) / np.maximum(cams[:, 3], 1e-6)[None]
    want = np.sqrt(0.1) * d.min(1)
    assert np.abs(got - want).max() <= 1e-6 * want.max()
    # ~0.32 px std-dev at the nearest camera (train.rs:39-44)
    near = d.argmin(1)
    px = got * cams[near, 3] / np.linalg.norm(tr[:, :3] - cams[near, :3], axis=1)
    assert np.allclose(px[near != 2], np.sqrt(0.1), rtol=1e-4)


def test_fold_min_scale_backward_matches_torch_autograd():
    """The reference gets this backward from burn's autodiff over the tensor expression (gaussian_splats.rs:86-111); the oracle's
    (and the library's) hand-derived chain rule is pinned against torch.autograd over the same expression in float64."""
    import torch
    rng = np.random.default_rng(11)
    n = 500
    tr = rng.normal(size=(n, 10)).astype(np.float32)
    tr[:, 7:] = rng.uniform(-6.0, -1.0, (n, 3)).astype(np.float32)
    op = rng.normal(0.0, 2.5, n).astype(np.float32)
    op[:5] = 14.0    # sigmoid * coef lands on the upper clamp only if coef ~ 1: keep a few splats near it
    f = rng.uniform(0.0, 0.08, n).astype(np.float32)
    f[:3] = 0.0
    v_t = rng.normal(size=(n, 10)).astype(np.float32)
    v_o = rng.normal(size=n).astype(np.float32)
    t64 = torch.tensor(tr.astype(np.float64), requires_grad=True)
    o64 = torch.tensor(op.astype(np.float64), requires_grad=True)
    f64 = torch.tensor(f.astype(np.float64))
    s2 = torch.exp(2.0 * t64[:, 7:10])
    s2f = s2 + (f64 * f64)[:, None]
    new_t = torch.cat([t64[:, :7], 0.5 * torch.log(s2f)], 1)
    coef = torch.sqrt(s2.prod(1) / s2f.prod(1))
    o = torch.clamp(torch.sigmoid(o64) * coef, 1e-6, 1.0 - 1e-6)
    new_o = torch.log(o / (1.0 - o))
    ((new_t * torch.tensor(v_t.astype(np.float64))).sum() + (new_o * torch.tensor(v_o.astype(np.float64))).sum()).backward()
    ft, fo = bo.fold_min_scale(tr, op, f)
    # (opacities compared as probabilities: at the 1 - 1e-6 clamp the logit amplifies one f32 ulp of the clamp bound to 1e-2)
    assert np.allclose(ft, new_t.detach().numpy(), rtol=2e-6, atol=2e-6) and np.allclose(1.0 / (1.0 + np.exp(-fo.astype(np.float64))), o.detach().numpy(), rtol=0, atol=2e-7)
    gt, go = bo.fold_min_scale_backward(tr, op, f, v_t, v_o)
    wt, wo = t64.grad.numpy(), o64.grad.numpy()
    assert np.abs(gt - wt).max() <= 2e-5 * np.abs(wt).max() and np.abs(go - wo).max() <= 2e-5 * np.abs(wo).max()
