"""The stochastic half of SplatTrainer::step on the device: the mean noise of
brush-train/src/train.rs:389-416 drawn by the library's counter-based generator (Philox-4x32-10 + Box-Muller,
brush_amd/csrc/device_rng.h) instead of injected samples.  The reference's stream (burn's GPU PRNG) is not
reproducible, so the contract is the DISTRIBUTION, the gate, and determinism in (seed, step, splat):
  * the generator against a numpy restatement of its specification,
  * N(0,1) statistics,
  * a seeded step == the same step with bh_normal_samples' tensor injected (bit-identical: fused and stand-alone
    noise paths agree, with and without the 3D-filter floor),
  * two trainers with the same seed stay bit-identical (what data-parallel replicas rely on), different seeds differ,
  * only visible, low-opacity splats move.
"""
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
import util

pytestmark = pytest.mark.gpu

M32 = 0xFFFFFFFF


def philox_np(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 on numpy uint64 arrays (Salmon et al. 2011)."""
    c0, c1, c2, c3 = [np.asarray(x, np.uint64) for x in (c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & np.uint64(M32)
        n1 = p1 & np.uint64(M32)
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & np.uint64(M32)
        n3 = p0 & np.uint64(M32)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & np.uint64(M32)
        k1 = (k1 + np.uint64(0xBB67AE85)) & np.uint64(M32)
    return c0, c1, c2, c3


def normal3_np(seed, step, n):
    i = np.arange(n, dtype=np.uint64)
    z = np.zeros(n, np.uint64)
    r = philox_np(i, z + np.uint64(step), z, z + np.uint64(0x4D4E0001), seed & M32, (seed >> 32) & M32)
    u = [((x >> np.uint64(9)).astype(np.float64) + 0.5) / 8388608.0 for x in r]
    ra, rb = np.sqrt(-2.0 * np.log(u[0])), np.sqrt(-2.0 * np.log(u[2]))
    return np.stack([ra * np.cos(2 * np.pi * u[1]), ra * np.sin(2 * np.pi * u[1]), rb * np.cos(2 * np.pi * u[3])], axis=1)


def _trainer(ba, seed, **kw):
    return ba.SplatTrainer(ba.TrainConfig(**kw), median_scene_scale=3.0, seed=seed)


def test_normal_samples_match_the_specification(dev):
    import brush_amd as ba
    tr = _trainer(ba, 0x1234ABCD5678)
    n = 200_000
    got = tr.normal_samples(n, 17, dev).cpu().numpy().astype(np.float64)
    want = normal3_np(0x1234ABCD5678, 17, n)
    # v_sin/v_cos (revolutions argument) and the f32 log polynomial vs float64: absolute 2e-5 on values of |x| <= 6
    assert np.abs(got - want).max() < 2e-5


def test_normal_samples_statistics(dev):
    import brush_amd as ba
    tr = _trainer(ba, 99)
    x = tr.normal_samples(1_000_000, 3, dev).cpu().numpy().astype(np.float64)
    assert np.isfinite(x).all()
    m, s = x.mean(0), x.std(0)
    assert np.abs(m).max() < 4e-3 and np.abs(s - 1.0).max() < 4e-3
    assert np.abs((x ** 3).mean(0)).max() < 2e-2                      # skewness 0
    assert np.abs((x ** 4).mean(0) - 3.0).max() < 5e-2                # kurtosis 3
    c = np.corrcoef(x.T)
    assert np.abs(c - np.eye(3)).max() < 4e-3                         # the three columns are independent
    y = tr.normal_samples(1_000_000, 4, dev).cpu().numpy().astype(np.float64)
    assert abs(float(np.corrcoef(x[:, 0], y[:, 0])[0, 1])) < 4e-3     # ... and so are consecutive steps
    assert abs(float(np.corrcoef(x[:-1, 0], x[1:, 0])[0, 1])) < 4e-3  # ... and neighbouring splats
    frac = (np.abs(x) > 3.0).mean()
    assert abs(frac - 0.0026998) < 3e-4                               # tails


def _problem(dev, ba, n=6000, w=160, h=96, deg=1, seed=0xD0):
    sc = synth.make_scene(n, seed, sh_degree=deg, log_scale_range=(math.log(0.02), math.log(0.2)),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w))
    # half of the splats nearly transparent: (1 - sigmoid)^150 is ~0 for the opaque half
    sc["raw_opac"][::2] = np.float32(-4.0)
    cp = synth.default_camera_params(w, h)
    gt = synth.synthetic_gt_packed(w, h)
    cam = util.hip_camera(ba, cp)
    batch = ba.SceneBatch(torch.from_numpy(gt.view(np.int32)).to(dev), cam)
    return sc, batch


def _rows_agree(x, y, tol):
    return np.abs(x.astype(np.float64) - y.astype(np.float64)).max(axis=1) <= tol


@pytest.mark.parametrize("floor", [False, True])
def test_seeded_step_equals_injected_samples(dev, floor):
    """A seeded step (noise drawn inside the update launch; with a 3D-filter floor: by the stand-alone noise kernel) against
    the same step with bh_normal_samples' tensor and bh_sample_background's colour injected.  Two runs of a step are not
    bit-identical (the backward sums with float atomics, and Adam turns a last-bit change of a gradient into a last-bit change
    of the update, a sign flip of a ~0 gradient into 2 lr), so: >= 99.5 % of the rows agree to 1e-3 lr_mean — the noise itself is
    of the order of lr_mean, a wrong sample, gate or scale would move every noised row by that much."""
    import brush_amd as ba
    sc, batch = _problem(dev, ba)
    n = sc["transforms"].shape[0]
    seed = 0xC0FFEE
    runs = []
    for injected in (False, True):
        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
        if floor:
            spl.with_min_scale(torch.full((n,), 0.01, device=dev))
        tr = _trainer(ba, seed)
        if injected:
            tr.step(batch, spl, background=tr.sample_background(), noise_samples=tr.normal_samples(n, 1, dev))
        else:
            tr.step(batch, spl)
        st = tr.stats()
        runs.append((spl.transforms.cpu().numpy(), spl.raw_opacities.cpu().numpy(), st.loss, st.lr_mean))
    lr = runs[0][3]
    noised = np.abs(runs[0][0][:, :3] - sc["transforms"][:, :3]).max(axis=1) > 1.5 * lr   # moved by more than the Adam step: noise
    assert noised.sum() > 200
    ok = _rows_agree(runs[0][0][:, :3], runs[1][0][:, :3], 1e-3 * lr)
    assert ok.mean() >= 0.995 and ok[noised].mean() >= 0.99, (ok.mean(), ok[noised].mean())
    assert abs(runs[0][2] - runs[1][2]) <= 1e-6 * abs(runs[1][2])


def test_same_seed_same_trajectory_and_the_gate(dev):
    import brush_amd as ba
    sc, batch = _problem(dev, ba)
    outs = {}
    for name, seed in (("a", 5), ("b", 5), ("c", 6), ("none", None)):
        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
        tr = ba.SplatTrainer(ba.TrainConfig(background_noise_strength=0.0), median_scene_scale=3.0, seed=seed)
        tr.step(batch, spl)
        st = tr.stats()
        outs[name] = (spl.transforms.cpu().numpy(), tr, st)
    lr = outs["a"][2].lr_mean
    same = _rows_agree(outs["a"][0], outs["b"][0], 1e-3 * lr)
    assert same.mean() >= 0.995, "same seed -> same noise (what data-parallel replicas rely on)"
    diff_seed = _rows_agree(outs["a"][0][:, :3], outs["c"][0][:, :3], 1e-3 * lr)
    # the noise is what separates a seeded step from an unseeded one (background jitter is off here)
    moved = ~_rows_agree(outs["a"][0][:, :3], outs["none"][0][:, :3], 1e-3 * lr)
    assert moved.sum() > 200 and diff_seed[moved].mean() < 0.05, "a different seed moves the noised rows elsewhere"
    assert _rows_agree(outs["a"][0][:, 3:], outs["none"][0][:, 3:], 1e-3 * 2e-3).mean() >= 0.995, "only the means are noised"
    vis = outs["none"][1].state["vis_weight"].cpu().numpy() > 0
    assert np.array_equal(outs["a"][0][~vis], sc["transforms"][~vis]), "invisible splats never move (no gradient, no noise)"
    # opaque splats: (1 - sigmoid(raw))^150 ~ 0 -> no visible noise; transparent visible ones: noise ~ lr_mean * 50 * w * N(0,1)
    opaque = (1.0 / (1.0 + np.exp(-sc["raw_opac"].astype(np.float64))) > 0.3) & vis
    assert (~moved[opaque]).mean() >= 0.99
    thin = (sc["raw_opac"] == np.float32(-4.0)) & vis
    assert moved[thin].mean() > 0.9
    d = np.abs(outs["a"][0][:, :3].astype(np.float64) - outs["none"][0][:, :3]).max(axis=1)
    assert d.max() <= 3.0 and d[moved].mean() < lr * 50 * 4


def test_loss_survives_refine_and_bounds_readbacks(dev):
    """bh_refine_plan / bh_splat_bounds reuse the pinned scalar block; the loss word must not be theirs."""
    import brush_amd as ba
    sc, batch = _problem(dev, ba)
    spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
    tr = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0)
    tr.step(batch, spl)
    want = tr.stats().loss
    assert want > 0.0
    spl2 = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
    tr2 = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0)
    tr2.step(batch, spl2)
    tr2.refine(100, spl2, seed=1)      # step -> refine -> stats(): the public loop on a refine iteration
    assert tr2.stats().loss == want


def test_failed_step_does_not_advance_the_step_count(dev):
    """an error return (here: the exchange hook fails) applied no update; Adam's t must not move."""
    import ctypes as C
    import brush_amd as ba
    from brush_amd import _ffi
    sc, batch = _problem(dev, ba)
    spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
    tr = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0)
    tr.step(batch, spl)
    assert tr.step_count == 1
    before = spl.transforms.clone()
    tr.pg = object()                                   # route through the hook path ...
    tr._world = 1
    tr._hook = _ffi.GRAD_HOOK(lambda user, p, cnt: 1)  # ... with a hook that reports failure
    with pytest.raises(ba.BrushHipError):
        tr.step(batch, spl)
    assert tr.step_count == 1
    torch.cuda.synchronize()
    assert torch.equal(spl.transforms, before)
