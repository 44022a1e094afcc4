"""oracle/refine.py (numpy restatement of SplatTrainer::refine's deterministic part) against the
closed-form properties the reference's split construction states (train.rs:705-735) and its bounds
tests (splat_init.rs:249-280)."""
import numpy as np

from oracle import refine as orf


def _state(n, coeffs=1, seed=0):
    rng = np.random.default_rng(seed)
    tr = np.concatenate([rng.normal(size=(n, 3)), rng.normal(size=(n, 4)), rng.uniform(-4, -1, (n, 3))], axis=1).astype(np.float32)
    st = dict(transforms=tr, sh=rng.normal(size=(n, coeffs, 3)).astype(np.float32), raw_opac=rng.normal(size=n).astype(np.float32))
    for k, shape in (("m1_t", (n, 10)), ("m2_t", (n, 10)), ("m1_sh", (n, coeffs, 3)), ("m2_sh", (n,)), ("m1_o", (n,)), ("m2_o", (n,))):
        st[k] = rng.normal(size=shape).astype(np.float32)
    return st


CFG = dict(split_at_screen_size=0.5, growth_grad_threshold=0.0025, growth_select_fraction=0.25, iter=1000, total_train_iters=30000, opac_decay=0.004)


def test_split_preserves_centroid_and_shrinks_the_long_axis():
    st = _state(50)
    keep = np.ones(50, bool)
    split = np.zeros(50, bool)
    split[[3, 17, 40]] = True
    cfg = dict(CFG, opac_decay=0.0)
    out = orf.apply(st, keep, split, cfg, np.zeros(50, np.float32))
    assert out["transforms"].shape[0] == 53
    for k, p in enumerate([3, 17, 40]):
        parent, child = out["transforms"][p], out["transforms"][50 + k]
        # anti-correlated offsets: centroid preserved (train.rs:754-758)
        assert np.allclose((parent[:3] + child[:3]) / 2, st["transforms"][p, :3], atol=1e-6)
        # the largest axis shrinks by 1/sqrt(2); no axis grows
        d = parent[7:10] - st["transforms"][p, 7:10]
        assert np.isclose(d.min(), np.log(orf.FRAC_1_SQRT_2), atol=1e-5) and (d <= 1e-6).all()
        assert np.allclose(parent[7:10], child[7:10], atol=1e-6)
        assert np.isclose(np.linalg.norm(child[3:7]), 1.0, atol=1e-5)
        # children/parents restart Adam
        for key in ("m1_t", "m2_t", "m1_sh", "m2_sh", "m1_o", "m2_o"):
            assert not out[key][p].any() and not out[key][50 + k].any()
    # untouched rows keep their moments
    assert np.array_equal(out["m1_t"][5], st["m1_t"][5])


def test_oversized_split_lands_at_the_screen_cap():
    st = _state(4)
    screen = np.array([0.0, 2.0, 0.0, 0.0], np.float32)   # 4x the cap -> k_max = 0.25 on the long axis
    split = np.array([False, True, False, False])
    out = orf.apply(st, np.ones(4, bool), split, dict(CFG, opac_decay=0.0), screen)
    d = out["transforms"][1, 7:10] - st["transforms"][1, 7:10]
    assert np.isclose(d.min(), np.log(0.25), atol=1e-5)


def test_prune_mask_reasons_and_stable_gather():
    st = _state(10)
    st["raw_opac"][1] = -9.0                      # opacity < 1/255
    st["transforms"][2, 8] = 9.0                  # scale > 100 x extent
    st["transforms"][3, 0] = 1e4                  # out of bounds
    st["sh"][4, 0, 1] = np.nan                    # non-finite
    mask, bad = orf.prune_mask(st["transforms"], st["sh"], st["raw_opac"], (0, 0, 0), (1.0, 0.5, 0.2))
    assert mask.tolist() == [False, True, True, True, True] + [False] * 5 and bad.sum() == 1
    out = orf.apply(st, ~mask, np.zeros(10, bool), dict(CFG, opac_decay=0.0), np.zeros(10, np.float32))
    assert np.array_equal(out["transforms"], st["transforms"][[0, 5, 6, 7, 8, 9]])


def test_opacity_decay_schedule():
    st = _state(5)
    out = orf.apply(st, np.ones(5, bool), np.zeros(5, bool), dict(CFG, iter=15000), np.zeros(5, np.float32))
    want = orf.sigmoid(st["raw_opac"]) - np.float32(0.004 * 0.5)
    assert np.allclose(orf.sigmoid(out["raw_opac"]), np.clip(want, 1e-12, 1), atol=2e-6)


def test_bounds_from_pos_cases():
    """splat_init.rs:249-280"""
    c, e = orf.bounds_from_pos(0.8, np.full((10, 3), np.nan, np.float32))
    assert np.isfinite(c).all() and np.isfinite(e).all()
    m = np.full((100, 3), np.nan, np.float32)
    m[1::2] = np.arange(1, 100, 2, dtype=np.float32)[:, None]
    c, e = orf.bounds_from_pos(0.8, m)
    assert np.isfinite(c).all() and (np.array(e) > 0).all()
    m = np.linspace(-1, 1, 1001, dtype=np.float32)[:, None].repeat(3, 1)
    c, e = orf.bounds_from_pos(0.8, m)
    assert np.allclose(c, 0, atol=2e-3) and np.allclose(e, 0.8, atol=3e-3)
