"""Split tiles (context.h SPLIT_MAX, rasterize.hip blend_tile<.., NQ = 1>): the forward blend hands the heaviest tiles of a view's
forecast to four quadrant waves.  Contract: kernels/rasterize.rs:27-190 — the image, the shrunk list ends, the visible flags and
(through the checkpoints the quadrant waves leave) the backward's gradients are what the one-wave-per-tile blend gives."""
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
import util

pytestmark = pytest.mark.gpu

W, H = 256, 192


def _scene(n, seed, spread, sh_degree=1, scales=(0.01, 0.08)):
    sc = synth.make_scene(n, seed, sh_degree=sh_degree, log_scale_range=(math.log(scales[0]), math.log(scales[1])),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * H / W), spread=spread)
    return sc, synth.default_camera_params(W, H)


def _frames(ba, dev, sc, cp, options, visits=3, pass_=None):
    """`visits` differentiable renders of ONE camera on a fresh context (the first seeds the view's per-tile work table, the later ones
    order — and split — their tiles by it); returns what the last one produced."""
    ctx = ba.Context(dev, options=dict({"cut_min_pairs": 0}, **options))
    spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
    cam = util.hip_camera(ba, cp)
    rng = np.random.default_rng(5)
    v_out = torch.from_numpy(rng.normal(size=(H, W, 4)).astype(np.float32)).to(dev)
    res = None
    for _ in range(visits):
        res = ba.render_splats_bwd(spl, cam, (W, H), (0.1, 0.2, 0.3), v_out, pass_ or ba.RasterPass.Backward, ctx=ctx)
    out = {"img": res["img"].cpu().numpy(), "offsets": res["aux"].tile_offsets.cpu().numpy().copy(),
           "visible": res["aux"].visible.cpu().numpy().copy(), "nv": res["aux"].num_visible, "ni": res["aux"].num_intersections,
           "v_t": res["v_transforms"].cpu().numpy(), "v_sh": res["v_sh_coeffs"].cpu().numpy(), "v_o": res["v_raw_opacities"].cpu().numpy(),
           "v_r": res["v_refine_weight"].cpu().numpy()}
    ctx.close()
    return out


@pytest.mark.parametrize("case", ["object_centric", "uniform_all_split", "low_opacity_long_lists"])
def test_split_tiles_give_the_whole_tile_results(dev, case):
    import brush_amd as ba
    if case == "object_centric":       # heavy tiles in the middle, empty ones around: the product's rule picks the split tiles
        sc, cp = _scene(60000, 0xA1, spread=0.35)
        split = {"k16_split": 150, "k16_split_min": 32, "k16_split_of_max": 30}
    elif case == "uniform_all_split":  # every tile with any work is split (64 per band at most)
        sc, cp = _scene(20000, 0xA2, spread=1.5)
        split = {"k16_split": 1, "k16_split_min": 1, "k16_split_of_max": 0}
    else:                              # faint splats: nothing saturates, lists of many segments (checkpoints of quadrants that stop at different entries)
        sc, cp = _scene(40000, 0xA3, spread=0.6, scales=(0.02, 0.12))
        sc["raw_opac"][:] = np.float32(-2.5)
        split = {"k16_split": 1, "k16_split_min": 1, "k16_split_of_max": 0}
    ref = _frames(ba, dev, sc, cp, {"k16_split": 0})
    got = _frames(ba, dev, sc, cp, split)
    assert got["nv"] == ref["nv"] and got["ni"] == ref["ni"]
    assert np.array_equal(got["img"], ref["img"])                # the same fold per pixel: bit for bit
    assert np.array_equal(got["offsets"], ref["offsets"])        # shrunk list ends (rasterize.rs:183-189)
    assert np.array_equal(got["visible"], ref["visible"])        # rasterize.rs:143-145
    for k in ("v_t", "v_sh", "v_o", "v_r"):                     # float atomics: equal up to the order of the sums
        scale = float(np.abs(ref[k]).max())
        assert scale > 0.0
        assert float(np.abs(got[k] - ref[k]).max()) <= 2e-5 * scale, k
    # ... and the whole-tile backward (no jobs, no checkpoints) of a split forward
    whole = _frames(ba, dev, sc, cp, dict(split, bwd_jobs=0))
    assert np.array_equal(whole["img"], ref["img"]) and np.array_equal(whole["offsets"], ref["offsets"])
    for k in ("v_t", "v_o"):
        assert float(np.abs(whole[k] - ref[k]).max()) <= 2e-5 * float(np.abs(ref[k]).max()), k


def test_split_tiles_are_actually_split(dev):
    """The rule fires on the object-centric frame (guards the test above against passing on an unsplit path): K1's split counts,
    read from the order table's tail through the test hook of the context's scratch."""
    import brush_amd as ba
    import ctypes as C
    from brush_amd import _ffi
    sc, cp = _scene(60000, 0xA1, spread=0.35)
    ctx = ba.Context(dev, lib=_ffi.load_test_hooks(), options={"cut_min_pairs": 0, "k16_split": 150, "k16_split_min": 32, "k16_split_of_max": 30})

    def split_counts():
        out = (C.c_uint32 * 8)()
        r = ctx.lib.bh_debug_split_counts(ctx._h, out)
        assert r in (0, 1)
        return list(out) if r == 1 else None
    spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
    cam = util.hip_camera(ba, cp)
    for _ in range(2):
        ba.render_splats(spl, cam, (W, H), (0.0, 0.0, 0.0), ba.RasterPass.Backward, ctx=ctx)
    counts = split_counts()
    assert counts is not None and len(counts) == 8 and 0 < sum(counts) <= 8 * 128
    ctx.set_option("k16_split", 0)
    ba.render_splats(spl, cam, (W, H), (0.0, 0.0, 0.0), ba.RasterPass.Backward, ctx=ctx)
    assert split_counts() is None
    ctx.close()


def test_split_tiles_in_cut_list_train_steps(dev):
    """Train steps with per-tile cuts (near pass = PHASE 1, parked tiles, second attempts) on split tiles against the unsplit path."""
    import brush_amd as ba
    sc, cp = _scene(30000, 0xA4, spread=0.5, sh_degree=0)
    n = sc["transforms"].shape[0]
    gt = synth.synthetic_gt_packed(W, H)
    gt_dev = torch.from_numpy(gt.view(np.int32)).to(dev)
    cams = []
    for dx in (0.0, 0.3):
        p = dict(cp)
        p["pos"] = (dx, 0.0, 0.0)
        cams.append(util.hip_camera(ba, p))
    runs = []
    for opts in ({"k16_split": 0}, {"k16_split": 1, "k16_split_min": 1, "k16_split_of_max": 0}):
        ctx = ba.Context(dev, options=dict({"cut_min_pairs": 0, "auto_exact_share": 0}, **opts))
        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
        trainer = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0, ctx=ctx)
        rng = np.random.default_rng(3)
        imgs = []
        for step in range(8):
            bg = tuple(float(x) for x in rng.uniform(0, 0.3, 3))
            noise = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)).to(dev)
            trainer.step(ba.SceneBatch(gt_dev, cams[step % 2], view_id=step % 2 + 1), spl, background=bg, noise_samples=noise)
            img, aux = ba.render_splats(spl, cams[step % 2], (W, H), (0.1, 0.2, 0.3), ba.RasterPass.Backward, ctx=ctx)
            imgs.append(img.cpu().numpy())
        runs.append((imgs, spl.transforms.cpu().numpy(), spl.raw_opacities.cpu().numpy(), trainer.stats().loss))
        ctx.close()
    (ia, ta, oa, la), (ib, tb, ob, lb) = runs
    for a, b in zip(ia, ib):
        d = np.abs(a - b)
        assert float(d.max()) <= 5e-3 and float(d.mean()) <= 2e-6
    cfg = ba.TrainConfig()
    util.assert_adam_close(ta[:, 3:7], tb[:, 3:7], cfg.lr_rotation, 8, "rotation")
    util.assert_adam_close(ta[:, 7:10], tb[:, 7:10], cfg.lr_scale, 8, "scale")
    util.assert_adam_close(oa, ob, cfg.lr_opac, 8, "opacity")
    assert abs(la - lb) <= 1e-4 * max(1.0, abs(la))
