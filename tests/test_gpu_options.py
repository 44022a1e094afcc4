"""bh_set_option (the library reads no environment variable) and the paths its keys select: every alternative path must give the
default path's results.  Also BhTrainConfig.growth_stop_iter: from that step on the blend backward leaves the refine weight out
(its one consumer stops reading it there, crates/brush-train/src/train.rs:589-614) — every other output unchanged."""
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
import util

pytestmark = pytest.mark.gpu


def _scene(n=6000, w=192, h=128, sh_degree=1, seed=0x51):
    sc = synth.make_scene(n, seed, sh_degree=sh_degree, log_scale_range=(math.log(0.02), math.log(0.2)),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w))
    return sc, synth.default_camera_params(w, h), synth.synthetic_gt_packed(w, h)


def test_set_option_validates_keys_and_values(dev):
    import brush_amd as ba
    ctx = ba.Context(dev)
    keys = ctx.options()
    assert {"cut_min_pairs", "event_waits", "readback_copy", "tile_sort", "k16_order", "grad_allreduce", "auto_exact_share", "band_mode", "k16_split", "k16_waves"} <= set(keys)
    assert all(len(v) > 10 for v in keys.values())   # one line of documentation each
    ctx.set_option("cut_min_pairs", 0)
    ctx.set_option("tile_sort", "lsd")
    ctx.set_option("cut_ctrl", "1.5:0.998:0.5:0.3333")
    for k, v in (("no_such_key", "1"), ("k16_order", "3"), ("tile_sort", "fast"), ("update_rows", "100"), ("event_waits", "yes"),
                 ("cut_ctrl", "0.5:2:0:9"), ("auto_exact_share", "1.5"), ("cut_min_pairs", "")):
        with pytest.raises(ba.BrushHipError):
            ctx.set_option(k, v)
    ctx.close()


def test_environment_translation_is_the_harness_not_the_library():
    """The Python mirror turns BH_OPTIONS / the legacy variable names into bh_set_option calls; nothing else reads them."""
    from brush_amd import host
    got = host.options_from_environment({"BH_EVENT_WAITS": "1", "BH_CUT_MIN_PAIRS": "0", "BH_TILE_SORT_LSD": "1", "BH_OPTIONS": "k16_order=2, no_lpt=1"})
    assert ("event_waits", "1") in got and ("cut_min_pairs", "0") in got and ("tile_sort", "lsd") in got
    assert got[-2:] == [("k16_order", "2"), ("no_lpt", "1")]


@pytest.mark.parametrize("options", [{"bwd_jobs": 0}, {"lpt_classes": "linear"}, {"spec_k5": 0}, {"event_waits": 1}, {"readback_copy": 1}, {"tile_sort": "lsd"}, {"tile_sort": "bucket"}, {"generic_depth_sort": 1}, {"dsort_splitters": 0},
                                     {"k16_order": 0}, {"k16_order": 2}, {"no_lpt": 1}, {"cut_sort_all": 1}, {"no_view_hash": 1},
                                     {"auto_exact_share": 0}, {"k5_exact_spw": 16}, {"k5_exact_spw": 64},
                                     {"band_mode": 0}, {"k16_waves": 5}, {"k16_split": 0}, {"k16_split": 1, "k16_split_min": 1, "k16_split_of_max": 0}])
def test_alternative_paths_give_the_default_results(dev, options):
    """Six alternating cut-list train steps (the host's mid-step waits, the sorts, the tile orders, the list builder's shapes) under
    each option against a context with the defaults: images bit-identical at every step, parameters equal up to atomic order."""
    import brush_amd as ba
    sc, cp, gt = _scene()
    n = sc["transforms"].shape[0]
    cams = []
    for dx in (0.0, 0.4):
        p = dict(cp)
        p["pos"] = (dx, 0.0, 0.0)
        cams.append(util.hip_camera(ba, p))
    gt_dev = torch.from_numpy(gt.view(np.int32)).to(dev)
    runs = []
    for opts in ({}, options):
        ctx = ba.Context(dev, options=dict({"cut_min_pairs": 0}, **opts))
        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
        trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0, ctx=ctx)
        rng = np.random.default_rng(3)
        imgs = []
        for step in range(6):
            bg = tuple(float(x) for x in rng.uniform(0, 0.3, 3))
            noise = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)).to(dev)
            trainer.step(ba.SceneBatch(gt_dev, cams[step % 2], view_id=step % 2 + 1), spl, background=bg, noise_samples=noise)
            img, aux = ba.render_splats(spl, cams[step % 2], (192, 128), (0.1, 0.2, 0.3), ba.RasterPass.Backward, ctx=ctx)
            imgs.append((img.cpu().numpy(), aux.num_visible, aux.num_intersections))
        runs.append((imgs, spl.transforms.cpu().numpy(), spl.raw_opacities.cpu().numpy(), trainer.stats().loss))
        ctx.close()
    (ia, ta, oa, la), (ib, tb, ob, lb) = runs
    # every run starts from identical parameters; after a step they differ by float-atomic order (an Adam step on a noise gradient is
    # +-lr whichever sign the noise takes: single pixels move by up to ~1e-3, the image as a whole does not)
    for (a, nva, nia), (b, nvb, nib) in zip(ia, ib):
        d = np.abs(a - b)
        assert float(d.max()) <= 5e-3 and float(d.mean()) <= 2e-6 and abs(nva - nvb) <= 3 and abs(nia - nib) <= max(8, nia // 2000)
    cfg = ba.TrainConfig()
    util.assert_adam_close(ta[:, 3:7], tb[:, 3:7], cfg.lr_rotation, 6, "rotation")
    util.assert_adam_close(ta[:, 7:10], tb[:, 7:10], cfg.lr_scale, 6, "scale")
    util.assert_adam_close(oa, ob, cfg.lr_opac, 6, "opacity")
    assert abs(la - lb) <= 1e-4 * max(1.0, abs(la))


@pytest.mark.parametrize("sh_degree,smooth_scene", [(0, False), (2, True)])
def test_backward_without_refine_weight_from_growth_stop_iter(dev, oracle_lib, sh_degree, smooth_scene):
    """growth_stop_iter = 3: steps 1-2 accumulate the refine weight like the reference, steps 3-5 leave it out.  Parameters, Adam
    moments, vis_weight and max_screen_size follow the oracle trainer throughout (which always computes it); refine_weight_norm
    stops at the oracle's value after step 2 (a running maximum: later steps of the oracle can only raise it)."""
    import brush_amd as ba
    sc, cp, gt = _scene(n=5000 if smooth_scene else 3000, sh_degree=sh_degree, seed=0x77 + sh_degree)
    n = sc["transforms"].shape[0]
    cfg = ba.TrainConfig(growth_stop_iter=3)
    ctx = ba.Context(dev, options={"cut_min_pairs": 0})
    trainer = ba.SplatTrainer(cfg, median_scene_scale=3.0, ctx=ctx)
    otr = util.OracleTrainer(oracle_lib, cfg, median_scene_scale=3.0)
    spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
    osc = {k: v.copy() for k, v in sc.items()}
    cam, ocam = util.hip_camera(ba, cp), oracle_lib.camera(**cp)
    gt_dev = torch.from_numpy(gt.view(np.int32)).to(dev)
    rng = np.random.default_rng(11)
    refine_after = {}
    for step in range(1, 6):
        bg = tuple(float(x) for x in rng.uniform(0, 0.3, 3))
        noise = rng.normal(size=(n, 3)).astype(np.float32)
        trainer.step(ba.SceneBatch(gt_dev, cam), spl, background=bg, noise_samples=torch.from_numpy(noise).to(dev))
        st = trainer.stats(ctx)
        ref = otr.step(osc, ocam, gt, bg, noise=noise)
        assert st.num_visible == ref["num_visible"] and abs(st.loss - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"]))
        tr = spl.transforms.cpu().numpy()
        util.assert_adam_close(tr[:, 3:7], osc["transforms"][:, 3:7], cfg.lr_rotation, step, "rotation")
        util.assert_adam_close(tr[:, 7:10], osc["transforms"][:, 7:10], cfg.lr_scale, step, "scale")
        util.assert_adam_close(tr[:, 0:3], osc["transforms"][:, 0:3], ref["lr_mean"], step, "mean", extra_abs=1e-7)
        util.assert_adam_close(spl.raw_opacities.cpu().numpy(), osc["raw_opac"], cfg.lr_opac, step, "opacity")
        util.assert_adam_close(spl.sh_coeffs.cpu().numpy(), osc["sh"], cfg.lr_coeffs_dc, step, "sh")
        refine_after[step] = (trainer.state["refine_weight_norm"].cpu().numpy().copy(), otr.state["refine"].copy())
    s = trainer.state
    assert np.mean(s["vis_weight"].cpu().numpy() != otr.state["vis"]) <= 2e-3
    assert util.rel_linf(s["max_screen_size"].cpu().numpy(), otr.state["screen"]) <= 1e-3
    # steps 1-2 computed it: equal to the oracle's record after step 2 ...
    assert util.rel_linf(refine_after[2][0], refine_after[2][1]) <= 1e-2 and float(refine_after[2][0].max()) > 0.0
    # ... and steps 3-5 left the record exactly where it was, while the oracle's kept growing
    assert np.array_equal(refine_after[5][0], refine_after[2][0])
    assert float((refine_after[5][1] - refine_after[2][1]).max()) > 0.0
    ctx.close()


def test_render_backward_always_computes_the_refine_weight(dev, oracle_lib):
    """bh_render_backward / _saved are the reference's operators: the refine weight is part of their result whatever a train step
    on the same context did before."""
    import brush_amd as ba
    sc, cp, gt = _scene(n=2500, sh_degree=0, seed=0x99)
    ctx = ba.Context(dev, options={"cut_min_pairs": 0})
    spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
    trainer = ba.SplatTrainer(ba.TrainConfig(growth_stop_iter=1), median_scene_scale=3.0, ctx=ctx)
    cam = util.hip_camera(ba, cp)
    trainer.step(ba.SceneBatch(torch.from_numpy(gt.view(np.int32)).to(dev), cam), spl)
    ctx.sync()
    assert float(trainer.state["refine_weight_norm"].abs().max()) == 0.0   # the step ran without it
    w, h = 192, 128
    v_out = torch.full((h, w, 4), 1.0 / (h * w), device=dev)
    res = ba.render_splats_bwd(spl, cam, (w, h), (0.0, 0.0, 0.0), v_out, ctx=ctx)
    ref = oracle_lib.Render().forward(oracle_lib.camera(**cp), spl.transforms.cpu().numpy(), spl.sh_coeffs.cpu().numpy(), spl.raw_opacities.cpu().numpy(),
                                      bg=(0.0, 0.0, 0.0), flags=oracle_lib.FLAG_BWD_INFO)
    ref.backward(v_out.cpu().numpy())
    got = res["v_refine_weight"].cpu().numpy()
    assert float(got.max()) > 0.0 and util.rel_linf(got, ref.get("v_refine")) <= 1e-4
    ctx.close()


def test_a_moving_viewer_camera_allocates_no_view_tables(dev):
    """forward-only frames without a view id are keyed by a hash of their camera — a free-moving viewer (or a pose-
    optimised camera) has a new hash every frame and must not mint a table per frame.  A table appears only when a camera comes
    back, at most 32 such tables exist, and training frames are unaffected."""
    import brush_amd as ba
    sc, cp, gt = _scene(n=4000, sh_degree=0, seed=0x42)
    ctx = ba.Context(dev, options={"cut_min_pairs": 0})
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)

    def cam_at(x):
        p = dict(cp)
        p["pos"] = (x, 0.0, 0.0)
        return util.hip_camera(ba, p)
    ref_img, _ = ba.render_splats(spl, cam_at(0.0), (192, 128), (0, 0, 0), ba.RasterPass.Forward, ctx=ctx)
    assert ctx.view_table_count() == 0
    for k in range(1, 200):   # a camera path: every frame a new camera
        ba.render_splats(spl, cam_at(0.001 * k), (192, 128), (0, 0, 0), ba.RasterPass.Forward, ctx=ctx, copy=False)
    assert ctx.view_table_count() == 0
    # the same camera again (within the ring of recently met cameras): now it gets its table, and the image is the same
    ba.render_splats(spl, cam_at(0.199), (192, 128), (0, 0, 0), ba.RasterPass.Forward, ctx=ctx, copy=False)
    assert ctx.view_table_count() == 1
    for rep in range(2):      # 100 fixed eval cameras rendered twice: capped
        for k in range(100):
            ba.render_splats(spl, cam_at(1.0 + 0.01 * k), (192, 128), (0, 0, 0), ba.RasterPass.Forward, ctx=ctx, copy=False)
    assert 1 <= ctx.view_table_count() <= 32
    img2, _ = ba.render_splats(spl, cam_at(0.0), (192, 128), (0, 0, 0), ba.RasterPass.Forward, ctx=ctx)
    assert torch.equal(ref_img, img2)
    # training frames keyed by their camera always get (and keep) their tables
    before = ctx.view_table_count()
    trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0, ctx=ctx)
    gt_dev = torch.from_numpy(gt.view(np.int32)).to(dev)
    for k in range(40):
        trainer.step(ba.SceneBatch(gt_dev, cam_at(5.0 + 0.01 * k)), spl)
    assert ctx.view_table_count() == before + 40
    ctx.close()


def test_views_render_complete_lists_when_cuts_save_nothing(dev):
    """a view whose last cut frame listed more than auto_exact_share (default 0.9) of its pairs — a scene that does not
    saturate early: where a converging training run ends up, bench.py train_loop — renders complete lists for its next 12 frames (no
    near count in K1, no second attempts) and then tries one cut frame again.  auto_exact_share = 0 keeps cutting."""
    import brush_amd as ba
    n, w, h = 8000, 192, 128
    # (large, mostly opaque splats: the tiles saturate after a fraction of their lists, so a cut frame lists well under 100 %)
    sc = synth.make_scene(n, 0x61, sh_degree=0, log_scale_range=(math.log(0.05), math.log(0.4)), opacity_range=(0.5, 0.95),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w))
    cam = util.hip_camera(ba, synth.default_camera_params(w, h))
    gt_dev = torch.from_numpy(synth.synthetic_gt_packed(w, h).view(np.int32)).to(dev)

    def run(share_option):
        ctx = ba.Context(dev, options={"cut_min_pairs": 0, "auto_exact_share": share_option})
        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
        trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0, ctx=ctx)
        seq = []
        for _ in range(30):
            trainer.step(ba.SceneBatch(gt_dev, cam, view_id=1), spl)
            seq.append(float(ctx.lib.bh_last_list_share(ctx._h)))
        ctx.close()
        return seq
    off = run("0")
    assert off[0] == 1.0 and all(0.05 < s < 1.0 for s in off[1:]), off          # frame 1 seeds the table, every later frame is cut
    lo = min(off[1:]) - 0.02                                                      # a threshold every cut frame of this scene exceeds
    on = run("%.4f" % lo)
    assert on[0] == 1.0 and lo < on[1] < 1.0                                      # frame 2 is cut and finds out that it saved too little
    assert on[2:14] == [1.0] * 12 and lo < on[14] < 1.0 and on[15:27] == [1.0] * 12, on   # 12 complete frames, one probe, 12 more


def test_speculative_list_builder_overflow_is_rendered_again(dev, oracle_lib):
    """The list builder (K5) is queued before the host has read the frame's counts, with the pair buffers' capacity as its bound: a
    frame whose pairs do not fit (here: a 20x larger scene on a context that has only seen a small one) must come out exactly as on
    a fresh context — every aux tensor bit for bit — and so must the frames after it."""
    import brush_amd as ba
    w, h = 256, 192
    cp = synth.default_camera_params(w, h)
    cam = util.hip_camera(ba, cp)
    tans = (math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w)
    small = synth.make_scene(3000, 0x31, sh_degree=0, log_scale_range=(math.log(0.02), math.log(0.2)), tan_half_fov=tans)
    big = synth.make_scene(60000, 0x32, sh_degree=0, log_scale_range=(math.log(0.02), math.log(0.25)), tan_half_fov=tans)

    def aux_arrays(ctx, sc):
        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
        img, aux = ba.render_splats(spl, cam, (w, h), (0.1, 0.2, 0.3), ba.RasterPass.Backward, ctx=ctx)
        ni = aux.num_intersections
        return dict(img=img.cpu().numpy(), nv=aux.num_visible, ni=ni, gids=util.u32(aux.compact_gid_from_isect)[:ni].copy(), tiles=util.u32(aux.tile_id_from_isect)[:ni].copy(),
                    offs=util.u32(aux.tile_offsets).copy(), proj=aux.projected_splats.cpu().numpy()[:aux.num_visible].copy(), gfc=util.u32(aux.global_from_compact_gid)[:aux.num_visible].copy())
    used = ba.Context(dev)
    for _ in range(3):   # the context's arena now holds the small scene's pairs, and K5 runs in front of the count readback
        a_small = aux_arrays(used, small)
    a_big = aux_arrays(used, big)          # 20x the pairs: the speculative launch overflows and is repeated
    a_big2 = aux_arrays(used, big)         # ... and now fits
    a_small2 = aux_arrays(used, small)
    fresh = ba.Context(dev, options={"spec_k5": 0})
    f_big, f_small = aux_arrays(fresh, big), aux_arrays(fresh, small)
    assert a_big["ni"] > 8 * a_small["ni"]
    for got, want in ((a_big, f_big), (a_big2, f_big), (a_small, f_small), (a_small2, f_small)):
        assert got["nv"] == want["nv"] and got["ni"] == want["ni"]
        for k in ("img", "gids", "tiles", "offs", "proj", "gfc"):
            assert np.array_equal(got[k], want[k]), k
    ref = oracle_lib.Render().forward(oracle_lib.camera(**cp), big["transforms"], big["sh"], big["raw_opac"], bg=(0.1, 0.2, 0.3), flags=oracle_lib.FLAG_BWD_INFO)
    assert a_big["ni"] == ref.num_intersections and np.array_equal(a_big["gids"], ref.get("compact_gid_from_isect"))
    used.close()
    fresh.close()


def test_backward_jobs_equal_the_whole_tile_backward_on_a_skewed_frame(dev, oracle_lib):
    """The blend backward works on checkpointed 128-entry segments of the tiles' lists (context.h BwdJobs): on a frame whose work is
    concentrated in a few tiles — large opaque-ish splats in the image centre, tiles with hundreds of blended entries next to empty
    ones, several segments per tile — the gradients equal the whole-tile backward's (option bwd_jobs = 0) and the oracle's."""
    import brush_amd as ba
    n, w, h = 30000, 320, 256
    sc = synth.make_scene(n, 0x71, sh_degree=1, log_scale_range=(math.log(0.03), math.log(0.25)), opacity_range=(0.02, 0.3), spread=0.45,
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w))
    cp = synth.default_camera_params(w, h)
    cam = util.hip_camera(ba, cp)
    rng = np.random.default_rng(4)
    v_out = torch.from_numpy((rng.normal(size=(h, w, 4)) * 1e-3).astype(np.float32)).to(dev)
    got = {}
    for jobs in (1, 0):
        ctx = ba.Context(dev, options={"bwd_jobs": jobs})
        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
        r = ba.render_splats_bwd(spl, cam, (w, h), (0.1, 0.2, 0.3), v_out, ctx=ctx)
        to = r["aux"].tile_offsets.to(torch.int64)
        work = (to[:, 1] - to[:, 0]).clamp(min=0)
        got[jobs] = {k: r[k].cpu().numpy() for k in ("v_transforms", "v_sh_coeffs", "v_raw_opacities", "v_refine_weight")}
        ctx.close()
    assert int(work.max()) > 4 * 128 and float(work.float().median()) < float(work.max()) / 8.0, (int(work.max()), float(work.float().median()))   # several segments per heavy tile, most tiles light
    ref = oracle_lib.Render().forward(oracle_lib.camera(**cp), sc["transforms"], sc["sh"], sc["raw_opac"], bg=(0.1, 0.2, 0.3), flags=oracle_lib.FLAG_BWD_INFO)
    ref.backward(v_out.cpu().numpy())
    want = {"v_transforms": ref.get("v_transforms").reshape(-1, 10), "v_sh_coeffs": ref.get("v_coeffs").reshape(got[1]["v_sh_coeffs"].shape),
            "v_raw_opacities": ref.get("v_raw_opac").reshape(-1), "v_refine_weight": ref.get("v_refine").reshape(-1)}
    for k, wv in want.items():
        m = max(float(np.abs(wv).max()), 1e-30)
        assert float(np.abs(got[1][k] - wv).max()) <= 1e-4 * m, k        # the north star's gradient tolerance, vs the oracle
        assert float(np.abs(got[1][k] - got[0][k]).max()) <= 2e-6 * m, k   # vs the whole-tile backward: rounding
