"""Mip-Splatting 3D filter on the MI355X: bh_fold_min_scale / _backward / bh_compute_min_scale vs the
oracle, renders of a Splats with a floor, and a train step with a floor.
The fold is f32 in the oracle's operation order with the shared exp/ln polynomials -> bit-exact;
the VJP multiplies the same terms -> compared to 1e-6 relative."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
import util
from test_gpu_render import assert_stagewise_exact, IMG_TOL

pytestmark = pytest.mark.gpu


def _scene(n, seed=0):
    rng = np.random.default_rng(seed)
    tr = rng.uniform(-1, 1, (n, 10)).astype(np.float32)
    tr[:, 7:] = rng.uniform(-6, -1, (n, 3))
    return tr, rng.uniform(-3, 4, n).astype(np.float32), rng.uniform(0.0005, 0.05, n).astype(np.float32)


@pytest.mark.parametrize("n", [1, 255, 256, 257, 100_003])
def test_fold_exact_vs_oracle_and_bake_in_place(dev, oracle_lib, n):
    import brush_amd as ba
    tr, op, f = _scene(n, n)
    spl = ba.Splats(tr, np.zeros((n, 1, 3), np.float32), op, device=dev, min_scale=f)
    ft, fo = spl.folded()
    rt, ro = oracle_lib.fold_min_scale(tr, op, f)
    assert np.array_equal(ft.cpu().numpy(), rt) and np.array_equal(fo.cpu().numpy(), ro)
    assert torch.equal(spl.scales(), torch.exp(torch.from_numpy(rt[:, 7:]).to(dev)))
    spl.bake_min_scale()            # out == in
    assert spl.min_scale is None
    assert np.array_equal(spl.transforms.cpu().numpy(), rt) and np.array_equal(spl.raw_opacities.cpu().numpy(), ro)


def test_fold_backward_vs_oracle(dev, oracle_lib):
    import brush_amd as ba
    n = 50_001
    tr, op, f = _scene(n, 9)
    rng = np.random.default_rng(1)
    vt, vo = rng.normal(size=(n, 10)).astype(np.float32), rng.normal(size=n).astype(np.float32)
    ctx = ba.get_context(dev)
    d = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    t, o, ff, gvt, gvo = d(tr), d(op), d(f), d(vt.copy()), d(vo.copy())
    ctx.check(ctx.lib.bh_fold_min_scale_backward(ctx._h, t.data_ptr(), o.data_ptr(), ff.data_ptr(), n, gvt.data_ptr(), gvo.data_ptr()))
    rt, ro = oracle_lib.fold_min_scale_backward(tr, op, f, vt, vo)
    assert np.array_equal(gvt.cpu().numpy()[:, :7], vt[:, :7])
    assert util.rel_linf(gvt.cpu().numpy()[:, 7:], rt[:, 7:]) <= 1e-6
    assert util.rel_linf(gvo.cpu().numpy(), ro) <= 1e-6


@pytest.mark.parametrize("k", [1, 3, 64, 65, 150])
def test_compute_min_scale_vs_oracle(dev, oracle_lib, k):
    import brush_amd as ba
    n = 20_000
    tr, _, _ = _scene(n, 2)
    rng = np.random.default_rng(k)
    cams = np.concatenate([rng.uniform(-4, 4, (k, 3)), rng.uniform(300, 1500, (k, 1))], 1).astype(np.float32)
    trainer = ba.SplatTrainer(ba.TrainConfig())
    trainer.set_view_cams([(c[:3], c[3]) for c in cams])
    spl = ba.Splats(tr, np.zeros((n, 1, 3), np.float32), np.zeros(n, np.float32), device=dev)
    got = trainer.compute_min_scale(spl).cpu().numpy()
    want = oracle_lib.compute_min_scale(tr, cams, 0.1)
    assert util.rel_linf(got, want) <= 1e-6
    assert ba.SplatTrainer(ba.TrainConfig()).compute_min_scale(spl) is None  # no cameras -> no floor (train.rs:107-109)


def test_render_with_floor_equals_render_of_folded_splats(dev, oracle_lib):
    """gaussian_splats.rs:379-386 / bwd/burn_glue.rs:260-270: forward exact, gradients chained through the fold."""
    import brush_amd as ba
    scene, w, h = synth.config_scene("10k_256", 1)
    cp = synth.default_camera_params(w, h)
    n = scene["transforms"].shape[0]
    cams = np.array([[0, 0, 0, 221.7], [1, 0, -1, 300.0]], np.float32)
    f = oracle_lib.compute_min_scale(scene["transforms"], cams, 0.1) * 8.0   # exaggerate so the floor bites
    spl = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev, min_scale=f)
    rng = np.random.default_rng(0)
    v_out = (rng.uniform(-1, 1, (h, w, 4)) / (h * w)).astype(np.float32)
    res = ba.render_splats_bwd(spl, util.hip_camera(ba, cp), (w, h), (0.1, 0.2, 0.3), torch.from_numpy(v_out).to(dev))
    ft, fo = oracle_lib.fold_min_scale(scene["transforms"], scene["raw_opac"], f)
    ref = oracle_lib.Render().forward(oracle_lib.camera(**cp), ft, scene["sh"], fo, bg=(0.1, 0.2, 0.3), flags=oracle_lib.FLAG_BWD_INFO)
    plain = oracle_lib.Render().forward(oracle_lib.camera(**cp), scene["transforms"], scene["sh"], scene["raw_opac"], bg=(0.1, 0.2, 0.3), flags=oracle_lib.FLAG_BWD_INFO)
    assert np.abs(ref.image() - plain.image()).max() > 1e-3, "the floor did not change the render: test is vacuous"
    assert_stagewise_exact(res["aux"], ref)
    assert np.abs(res["img"].cpu().numpy() - ref.image()).max() <= IMG_TOL
    ref.backward(v_out)
    gt, go = oracle_lib.fold_min_scale_backward(scene["transforms"], scene["raw_opac"], f, ref.get("v_transforms").reshape(n, 10), ref.get("v_raw_opac"))
    g = res["v_transforms"].cpu().numpy()
    for sl in (slice(0, 3), slice(3, 7), slice(7, 10)):
        assert util.rel_linf(g[:, sl], gt[:, sl]) <= 1e-4
    assert util.rel_linf(res["v_raw_opacities"].cpu().numpy(), go) <= 1e-4


def test_train_steps_with_floor_match_oracle_trainer(dev, oracle_lib):
    import brush_amd as ba
    n, w, h = 4000, 160, 96
    sc = synth.make_scene(n, 0xF3, sh_degree=1, log_scale_range=(math.log(0.02), math.log(0.2)),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w))
    cp = synth.default_camera_params(w, h)
    gt = synth.synthetic_gt_packed(w, h)
    cfg = ba.TrainConfig()
    trainer = ba.SplatTrainer(cfg, median_scene_scale=3.0)
    trainer.set_view_cams([((0.0, 0.0, 0.0), 138.0)])
    spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
    f = trainer.compute_min_scale(spl) * 6.0
    spl.with_min_scale(f)
    fh = f.cpu().numpy()
    otr = util.OracleTrainer(oracle_lib, cfg, median_scene_scale=3.0)
    osc = {k: v.copy() for k, v in sc.items()}
    ocam = oracle_lib.camera(**cp)
    rng = np.random.default_rng(7)
    batch = ba.SceneBatch(torch.from_numpy(gt.view(np.int32)).to(dev), util.hip_camera(ba, cp))
    for step in range(3):
        noise = rng.normal(size=(n, 3)).astype(np.float32)
        trainer.step(batch, spl, noise_samples=torch.from_numpy(noise).to(dev))
        st = trainer.stats()
        ref = otr.step(osc, ocam, gt, (0.0, 0.0, 0.0), noise=noise, min_scale=fh)
        assert st.num_visible == ref["num_visible"] and st.num_intersections == ref["num_intersections"]
        assert abs(st.loss - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
        tr = spl.transforms.cpu().numpy()
        util.assert_adam_close(tr[:, 7:10], osc["transforms"][:, 7:10], cfg.lr_scale, step + 1, "scale")
        util.assert_adam_close(tr[:, 3:7], osc["transforms"][:, 3:7], cfg.lr_rotation, step + 1, "rotation")
        util.assert_adam_close(tr[:, 0:3], osc["transforms"][:, 0:3], ref["lr_mean"], step + 1, "mean", extra_abs=1e-7)
        util.assert_adam_close(spl.raw_opacities.cpu().numpy(), osc["raw_opac"], cfg.lr_opac, step + 1, "opacity")
    assert spl.min_scale is not None   # the floor stays attached between refines


def test_refine_bakes_and_recomputes_the_floor(dev):
    """train.rs:433-437 and :636-648."""
    import brush_amd as ba
    n, w, h = 3000, 128, 128
    sc = synth.make_scene(n, 0xF4, sh_degree=0, log_scale_range=(math.log(0.02), math.log(0.2)))
    cp = synth.default_camera_params(w, h)
    gt = synth.synthetic_gt_packed(w, h)
    trainer = ba.SplatTrainer(ba.TrainConfig(total_train_iters=1000), median_scene_scale=3.0)
    trainer.set_view_cams([((0.0, 0.0, 0.0), 110.0), ((0.5, 0.0, 0.0), 110.0)])
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    spl.with_min_scale(trainer.compute_min_scale(spl))
    batch = ba.SceneBatch(torch.from_numpy(gt.view(np.int32)).to(dev), util.hip_camera(ba, cp))
    for _ in range(3):
        trainer.step(batch, spl)
    before = spl.clone()
    expect_t, expect_o = before.folded()
    new, stats = trainer.refine(200, spl, seed=5)
    assert spl.min_scale is None                                   # baked in place before the plan
    assert torch.equal(spl.transforms, expect_t) and torch.equal(spl.raw_opacities, expect_o)
    assert new.min_scale is not None and new.min_scale.numel() == stats.total_splats == new.num_splats()
    assert torch.equal(new.min_scale, trainer.compute_min_scale(new))
    late, _ = trainer.refine(950, new, seed=6)                     # past MIN_SCALE_FREEZE_FRAC: stays baked
    assert late.min_scale is None
