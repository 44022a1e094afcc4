"""SplatTrainer::refine on the device (bh_refine_plan / bh_refine_apply / bh_splat_bounds) vs the
oracle restatement (oracle/refine.py).  The reference draws its split candidates from an unseeded RNG
(multinomial.rs:2), so WHICH splats are sampled is not contractual: the test feeds the indices the
HIP plan chose to the oracle and checks (i) the counting rules that bound each selection stage,
(ii) the sampling invariants (no zero-weight pick, no duplicates, seed-determinism, weight bias),
(iii) every output tensor after prune + split + decay, and (iv) the percentile bounds."""
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
from oracle import refine as orf

pytestmark = pytest.mark.gpu


def _setup(dev, n=20000, deg=1, seed=0xF1):
    import brush_amd as ba
    sc = synth.make_scene(n, seed, sh_degree=deg, log_scale_range=(math.log(0.02), math.log(0.3)))
    rng = np.random.default_rng(seed)
    # make every prune reason and every split reason occur
    sc["raw_opac"][rng.choice(n, 900, replace=False)] = -7.0
    sc["transforms"][rng.choice(n, 40, replace=False), 8] = 12.0
    sc["transforms"][rng.choice(n, 30, replace=False), 1] = 5e4
    sc["sh"][rng.choice(n, 25, replace=False), 0, 2] = np.inf
    sc["transforms"][rng.choice(n, 10, replace=False), 4] = np.nan
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    cfg = ba.TrainConfig()
    tr = ba.SplatTrainer(cfg, median_scene_scale=3.0)
    tr._init_state(spl)
    st = tr.state
    host = {}
    for k, lo, hi in (("m1_t", -1, 1), ("m2_t", 0, 1), ("m1_sh", -1, 1), ("m2_sh", 0, 1), ("m1_o", -1, 1), ("m2_o", 0, 1)):
        host[k] = rng.uniform(lo, hi, tuple(st[k].shape)).astype(np.float32)
        st[k].copy_(torch.from_numpy(host[k]))
    host["refine_weight_norm"] = (rng.uniform(0, 0.01, n) * (rng.uniform(size=n) < 0.5)).astype(np.float32)
    host["vis_weight"] = np.floor(rng.uniform(0, 3, n)).astype(np.float32)
    host["max_screen_size"] = (rng.uniform(0, 0.3, n) + 0.6 * (rng.uniform(size=n) < 0.01)).astype(np.float32)
    for k in ("refine_weight_norm", "vis_weight", "max_screen_size"):
        st[k].copy_(torch.from_numpy(host[k]))
    tr.set_bounds((0.0, 0.0, 6.0), (4.0, 2.5, 5.0))
    tr.step_count = 400
    return ba, sc, spl, tr, host


@pytest.mark.parametrize("deg,iter_,max_splats", [(1, 400, 10_000_000), (0, 400, 20_150), (2, 20000, 10_000_000)])
def test_refine_matches_oracle_given_the_same_choices(dev, deg, iter_, max_splats):
    ba, sc, spl, tr, host = _setup(dev, deg=deg)
    tr.config.max_splats = max_splats
    n = spl.num_splats()
    bounds = tr.bounds
    new, rs = tr.refine(iter_, spl, seed=1234)
    plan = {k: v.cpu().numpy().astype(np.int64) for k, v in tr.last_refine_plan.items()}
    keep, split = plan["keep"].astype(bool), plan["split"].astype(bool)
    # ---- (i) prune decision and counting rules
    mask, bad = orf.prune_mask(sc["transforms"], sc["sh"], sc["raw_opac"], bounds[0], bounds[1])
    assert np.array_equal(keep, ~mask)
    assert rs.num_pruned == int(mask.sum()) and rs.num_pruned_non_finite == int(bad.sum())
    rcfg = dict(split_at_screen_size=tr.config.split_at_screen_size, growth_grad_threshold=tr.config.growth_grad_threshold,
                growth_select_fraction=tr.config.growth_select_fraction, iter=iter_, total_train_iters=tr.config.total_train_iters,
                opac_decay=tr.config.opac_decay)
    bc = orf.budget_counts(n, keep, host["vis_weight"], host["refine_weight_norm"], host["max_screen_size"], rcfg)
    assert not (split & ~keep).any()
    vis = host["vis_weight"] > 0
    w1 = keep & vis            # opacity x visibility weights (opacity >= 1/255 for every kept splat)
    k1 = min(bc["pruned"], int(w1.sum()))
    assert rs.num_resampled == k1
    budget = max(0, max_splats - (bc["n_keep"] + k1))
    stage1 = _stage1_only(dev, deg, iter_, max_splats)
    assert int(stage1.sum()) == k1 and not (stage1 & ~split).any()
    assert rs.num_split_oversized == min(budget, int((bc["oversized"] & ~stage1).sum()))
    growing = iter_ < min(tr.config.growth_stop_iter, tr.config.total_train_iters)
    if growing:
        headroom = max(0, max_splats - (bc["n_keep"] + k1 + rs.num_split_oversized))
        k3 = min(max(0, bc["grow_count"] - bc["pruned"]), headroom, int(bc["above"].sum()))
        assert rs.num_split_high_grad <= k3
        assert rs.num_added >= max(k1, rs.num_split_oversized) and rs.num_added <= k1 + rs.num_split_oversized + k3
    else:
        assert rs.num_split_high_grad == 0
    assert rs.num_added == int(split.sum()) and rs.total_splats == bc["n_keep"] + rs.num_added == new.num_splats()
    assert rs.total_splats <= max(max_splats, bc["n_keep"] + k1)
    # every split splat is legitimate: sampled from a positive weight, oversized, or above the gradient threshold
    assert not (split & ~(w1 | bc["oversized"] | bc["above"])).any()
    # ---- (iii) tensors after prune + split + decay, children in ascending parent order
    state = dict(transforms=sc["transforms"], sh=sc["sh"], raw_opac=sc["raw_opac"], **{k: host[k] for k in ("m1_t", "m2_t", "m1_sh", "m2_sh", "m1_o", "m2_o")})
    ref = orf.apply(state, keep, split, rcfg, host["max_screen_size"])
    got = dict(transforms=new.transforms, sh=new.sh_coeffs, raw_opac=new.raw_opacities, **{k: tr.state[k] for k in ("m1_t", "m2_t", "m1_sh", "m2_sh", "m1_o", "m2_o")})
    for k, r in ref.items():
        g = got[k].cpu().numpy()
        assert g.shape == r.shape, k
        assert np.allclose(g, r, rtol=2e-5, atol=2e-5), (k, float(np.abs(g - r).max()))
    for k in ("refine_weight_norm", "vis_weight", "max_screen_size"):
        assert not tr.state[k].any()      # fresh RefineRecord
    # ---- (iv) bounds of the new splats
    c, e = orf.bounds_from_pos(0.8, ref["transforms"][:, :3])
    assert np.allclose(tr.bounds[0], c, atol=1e-6) and np.allclose(tr.bounds[1], e, atol=1e-6)
    assert abs(tr.median_scene_scale - sorted(e)[1] * 2.0) <= 1e-6


def _stage1_only(dev, deg, iter_, max_splats):
    """The resample stage in isolation: same scene / state / seed, the other two stages disabled
    (their keys are independent streams, so the stage-1 picks are the same set)."""
    _, _, spl, tr, _ = _setup(dev, deg=deg)
    tr.config.max_splats = max_splats
    tr.config.split_at_screen_size = 0.0
    tr.config.growth_stop_iter = 0
    tr.refine(iter_, spl, seed=1234)
    return tr.last_refine_plan["split"].cpu().numpy().astype(bool)


def test_sampling_invariants_and_seed_determinism(dev):
    ba, sc, spl, tr, host = _setup(dev, deg=0)
    tr.config.split_at_screen_size = 0.0        # isolate the two sampling stages
    snap = {k: v.clone() for k, v in tr.state.items()}
    picks = []
    for seed in (7, 7, 8):
        tr.state = {k: v.clone() for k, v in snap.items()}
        tr.step_count = 400
        tr.set_bounds((0.0, 0.0, 6.0), (4.0, 2.5, 5.0))
        _, rs = tr.refine(400, spl, seed=seed)
        picks.append(tr.last_refine_plan["split"].cpu().numpy().astype(bool))
    assert np.array_equal(picks[0], picks[1]), "same seed -> same decision (data-parallel replicas stay identical)"
    assert not np.array_equal(picks[0], picks[2])
    keep = tr.last_refine_plan["keep"].cpu().numpy().astype(bool)
    vis = host["vis_weight"] > 0
    above = vis & (host["refine_weight_norm"] > tr.config.growth_grad_threshold)
    for p in picks:
        assert not (p & ~keep).any() and not (p & ~vis).any()          # zero-weight splats are never drawn
    # weight bias of the gradient stage: picked splats carry larger refine weights than the candidates on average
    only_grad = picks[0] & above
    assert host["refine_weight_norm"][only_grad].mean() > host["refine_weight_norm"][keep & above].mean()


def test_refine_then_training_continues(dev):
    """crates/brush-bench-test/tests/integration.rs:186-235 pattern: steps, refine, steps; splats > 0, finite."""
    import brush_amd as ba
    sc = synth.make_scene(5000, 0xF3, sh_degree=1, log_scale_range=(math.log(0.02), math.log(0.2)))
    w, h = 160, 128
    cp = synth.default_camera_params(w, h)
    cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h).view(np.int32)).to(dev)
    cfg = ba.TrainConfig(refine_every=5, growth_grad_threshold=1e-5)
    tr = ba.SplatTrainer(cfg, median_scene_scale=3.0)
    batch = ba.SceneBatch(gt, cam)
    counts = []
    for it in range(1, 16):
        tr.step(batch, spl)
        if it % cfg.refine_every == 0:
            spl, rs = tr.refine(it, spl)
            counts.append(rs.total_splats)
            assert rs.total_splats == spl.num_splats() > 0
    assert max(counts) > 5000 - 1, "growth happened"
    assert np.isfinite(tr.stats().loss)
    for t in (spl.transforms, spl.sh_coeffs, spl.raw_opacities):
        assert bool(torch.isfinite(t).all())


def test_splat_bounds_edge_cases(dev):
    """splat_init.rs:249-280"""
    import brush_amd as ba
    nan = ba.Splats(np.full((10, 10), np.nan, np.float32), np.zeros((10, 1, 3), np.float32), np.zeros(10, np.float32), device=dev)
    c, e = ba.splat_bounds(nan)
    assert c == (0.0, 0.0, 0.0) and e == (1.0, 1.0, 1.0)
    t = np.zeros((100, 10), np.float32)
    t[:, :3] = np.nan
    t[1::2, :3] = np.arange(1, 100, 2, dtype=np.float32)[:, None]
    c, e = ba.splat_bounds(ba.Splats(t, np.zeros((100, 1, 3), np.float32), np.zeros(100, np.float32), device=dev))
    rc, re = orf.bounds_from_pos(0.8, t[:, :3])
    assert np.allclose(c, rc) and np.allclose(e, re)
