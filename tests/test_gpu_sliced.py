"""Depth-sliced per-tile lists (BH_FLAG_SLICED_LISTS, the train step's default) against the exact path and the oracle.

Contract (include/brush_hip.h): out_img / out_img_packed bit for bit, visible[], max_radius, num_visible,
num_intersections, gradients and refine weights are those of the exact path; the list outputs are truncated but a
tile's blended splats — its near segment followed by its far segment — are exactly the exact path's shrunk list.
Reference semantics: kernels/rasterize.rs:116-189 (a tile stops once its pixels saturate), map_gaussians.rs:15-80."""
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
import util

pytestmark = pytest.mark.gpu


def _scene(n, w, h, seed, opacity=(0.05, 0.95), scales=(0.02, 0.2), sh_degree=0):
    sc = synth.make_scene(n, seed, sh_degree=sh_degree, log_scale_range=(math.log(scales[0]), math.log(scales[1])),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w), opacity_range=opacity)
    return sc, synth.default_camera_params(w, h)


def _reference_render(ba, ctx, *args, **kw):
    """A complete-list render that must not seed the per-view table of the camera under test (since round 5 complete-list frames
    take part in the per-view state: the forward blend's tile order): it runs under a view id of its own."""
    ba.set_view_id(0xEEEE, ctx)
    try:
        return ba.render_splats(*args, ctx=ctx, **kw)
    finally:
        ba.set_view_id(0, ctx)


def _blended_lists(aux):
    """per tile: the splats (GLOBAL ids) the blend kernels consumed, front to back — compact ids are translated, because a frame
    with per-tile cuts numbers only the splats that own a listed pair (a sub-sequence of the full depth order)"""
    gfc = util.u32(aux.global_from_compact_gid)
    cg = util.u32(aux.compact_gid_from_isect)
    assert cg.size == 0 or int(cg.max()) < gfc.size
    gids = gfc[cg] if cg.size else cg
    near = util.u32(aux.tile_offsets).reshape(-1, 2)
    far = util.u32(aux.tile_offsets_far).reshape(-1, 2) if aux.tile_offsets_far is not None else None
    out = []
    for t in range(near.shape[0]):
        seg = gids[near[t, 0]:max(near[t, 1], near[t, 0])]
        if far is not None and far[t, 1] > far[t, 0]:
            seg = np.concatenate([seg, gids[far[t, 0]:far[t, 1]]])
        out.append(seg)
    return out


def _assert_same_blended_list(exact, sliced, what):
    """A tile's replay list under slicing = the exact path's shrunk list, except that splats at the END of the near segment
    that touched no pixel may be missing (the near segment of a tile that goes on into the far slice is cut at ITS last useful
    splat): a subset in the same (depth) order that ends with the same last useful splat.  That nothing useful is missing is
    what the image / visible / gradient comparisons show."""
    assert len(sliced) <= len(exact), what
    if len(exact) == 0:
        return
    assert len(sliced) > 0 and sliced[-1] == exact[-1], what
    # a sub-sequence: every entry occurs in the exact list, at strictly increasing positions (a tile lists a splat once)
    order = np.argsort(exact, kind="stable")
    at = np.searchsorted(exact[order], sliced)
    assert np.all(at < len(exact)) and np.array_equal(exact[order][at], sliced), what
    assert np.all(np.diff(order[at]) > 0), what


def _assert_same_render(ba, spl, cam, size, bg, share, pass_=None, tile_rows=None):
    pass_ = pass_ or ba.RasterPass.Backward
    ctx = ba.get_context(spl.device)
    img_e, aux_e = ba.render_splats(spl, cam, size, bg, pass_, tile_rows=tile_rows)
    ba.host.set_list_slicing(share, ctx)
    try:
        img_s, aux_s = ba.render_splats(spl, cam, size, bg, pass_, tile_rows=tile_rows, sliced=True)
    finally:
        ba.host.set_list_slicing(0.0, ctx)
    if tile_rows is not None:   # a strip render writes only its own pixel rows
        r0, r1 = tile_rows[0] * 16, min(tile_rows[1] * 16, size[1])
        img_e, img_s = img_e[r0:r1], img_s[r0:r1]
    assert torch.equal(img_e, img_s), "image differs (share %g): max %g" % (share, float((img_e.float() - img_s.float()).abs().max()))
    assert aux_e.num_visible == aux_s.num_visible and aux_e.num_intersections == aux_s.num_intersections
    assert torch.equal(aux_e.max_radius, aux_s.max_radius)
    if aux_s.num_listed_splats == aux_s.num_visible:
        assert torch.equal(aux_e.global_from_compact_gid, aux_s.global_from_compact_gid)
    else:
        # a frame with per-tile cuts (the exact render above seeded this camera's table: complete-list frames take part in the
        # per-view state since round 5) numbers only the splats that own a listed pair: a sub-sequence of the full depth order
        ge, gs = util.u32(aux_e.global_from_compact_gid), util.u32(aux_s.global_from_compact_gid)
        pos = np.full(int(ge.max()) + 1 if ge.size else 1, -1, np.int64)
        pos[ge] = np.arange(ge.size)
        assert gs.size == 0 or (np.all(pos[gs] >= 0) and np.all(np.diff(pos[gs]) > 0))
    if pass_.bwd_info():
        assert torch.equal(aux_e.visible, aux_s.visible)
        le, ls = _blended_lists(aux_e), _blended_lists(aux_s)
        for t, (a, b) in enumerate(zip(le, ls)):
            _assert_same_blended_list(a, b, "tile %d (share %g)" % (t, share))
        # projected rows of every blended splat are the exact path's (a fixed share keeps the full depth order: same compact ids)
        ce, cs = util.u32(aux_e.compact_gid_from_isect), util.u32(aux_s.compact_gid_from_isect)
        near_e, near_s = util.u32(aux_e.tile_offsets).reshape(-1, 2), util.u32(aux_s.tile_offsets).reshape(-1, 2)
        used = np.unique(np.concatenate([ce[a:max(a, b)] for a, b in near_e])) if len(near_e) else np.zeros(0, np.int64)
        pe, ps = aux_e.projected_splats.cpu().numpy(), aux_s.projected_splats.cpu().numpy()
        if aux_s.num_listed_splats == aux_s.num_visible:
            assert np.array_equal(pe[used], ps[used])
        else:   # other compact numbering (per-tile cuts): compare the rows by splat id
            ge, gs = util.u32(aux_e.global_from_compact_gid), util.u32(aux_s.global_from_compact_gid)
            row_s = np.full(int(max(ge.max(), gs.max())) + 1, -1, np.int64)
            row_s[gs] = np.arange(gs.size)
            assert np.all(row_s[ge[used]] >= 0), "a blended splat is missing from the cut frame's compact arrays"
            assert np.array_equal(pe[used], ps[row_s[ge[used]]])
        del cs, near_s
    return aux_e, aux_s


@pytest.mark.parametrize("share", [0.0, 0.02, 0.1, 0.35, 0.8, 1.0])
@pytest.mark.parametrize("opacity", [(0.05, 0.95), (0.02, 0.1)])
def test_sliced_forward_equals_exact(dev, share, opacity):
    """saturating and non-saturating scenes, from a near slice that finishes nothing to one that holds everything"""
    import brush_amd as ba
    n, w, h = 30000, 320, 208
    sc, cp = _scene(n, w, h, 0x51, opacity)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    cam = util.hip_camera(ba, cp)
    aux_e, aux_s = _assert_same_render(ba, spl, cam, (w, h), (0.1, 0.3, 0.2), share)
    if 0.0 < share < 1.0:
        assert aux_s.list_budget < aux_s.num_intersections and aux_s.tile_offsets_far is not None
    if share == 1.0:
        assert aux_s.tile_offsets_far is None and aux_s.list_budget == aux_s.num_intersections
        assert torch.equal(aux_e.compact_gid_from_isect, aux_s.compact_gid_from_isect)


@pytest.mark.parametrize("share", [0.03, 0.3])
def test_sliced_forward_only_and_smooth_passes(dev, share):
    import brush_amd as ba
    n, w, h = 20000, 250, 170   # ragged edge tiles
    sc, cp = _scene(n, w, h, 0x52, sh_degree=1)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    cam = util.hip_camera(ba, cp)
    _assert_same_render(ba, spl, cam, (w, h), (0.0, 0.0, 0.0), share, ba.RasterPass.Forward)
    _assert_same_render(ba, spl, cam, (w, h), (0.3, 0.1, 0.5), share, ba.RasterPass.BackwardSmoothCutoff)


def test_sliced_strip_render(dev):
    """tile-row window (one frame over several ranks): the strip's tiles slice like a whole frame's"""
    import brush_amd as ba
    n, w, h = 20000, 256, 256
    sc, cp = _scene(n, w, h, 0x53)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    cam = util.hip_camera(ba, cp)
    for rows in ((0, 5), (5, 11), (11, 16)):
        _assert_same_render(ba, spl, cam, (w, h), (0.2, 0.2, 0.2), 0.1, tile_rows=rows)


@pytest.mark.parametrize("share,opacity", [(0.05, (0.05, 0.95)), (0.3, (0.05, 0.95)), (0.2, (0.02, 0.1))])
def test_sliced_backward_equals_exact_and_oracle(dev, oracle_lib, share, opacity):
    import brush_amd as ba
    n, w, h = 20000, 256, 160
    sc, cp = _scene(n, w, h, 0x54, opacity, sh_degree=2)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    cam = util.hip_camera(ba, cp)
    bg = (0.1, 0.2, 0.3)
    rng = np.random.default_rng(5)
    v_out = torch.from_numpy(rng.normal(size=(h, w, 4)).astype(np.float32)).to(dev)
    ctx = ba.get_context(dev)
    ex = ba.render_splats_bwd(spl, cam, (w, h), bg, v_out)
    ba.host.set_list_slicing(share, ctx)
    try:
        sl = ba.render_splats_bwd(spl, cam, (w, h), bg, v_out, sliced=True)
    finally:
        ba.host.set_list_slicing(0.0, ctx)
    assert sl["aux"].tile_offsets_far is not None
    assert torch.equal(ex["img"], sl["img"])
    ref = oracle_lib.Render().forward(oracle_lib.camera(**cp), sc["transforms"], sc["sh"], sc["raw_opac"], bg=bg, flags=oracle_lib.FLAG_BWD_INFO)
    ref.backward(v_out.cpu().numpy())
    for key, okey in (("v_transforms", "v_transforms"), ("v_sh_coeffs", "v_coeffs"), ("v_raw_opacities", "v_raw_opac"), ("v_refine_weight", "v_refine")):
        a, b = ex[key].cpu().numpy().reshape(-1), sl[key].cpu().numpy().reshape(-1)
        o = ref.get(okey).reshape(-1)
        scale = max(float(np.abs(o).max()), 1e-12)
        # the two HIP paths run the same replay; only the float-atomic order differs
        assert float(np.abs(a - b).max()) <= 2e-6 * scale, key
        assert float(np.abs(b - o).max()) <= 1e-4 * scale, key   # the gradients' stated tolerance vs the oracle


def test_sliced_lists_hold_the_exact_lists_prefix(dev):
    """the near slice's pairs are the first `budget` slots of the exact list: sorted by tile they are, per tile, a prefix
    of the exact tile list (in order)"""
    import brush_amd as ba
    n, w, h = 30000, 320, 208
    sc, cp = _scene(n, w, h, 0x55)
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    cam = util.hip_camera(ba, cp)
    ctx = ba.get_context(dev)
    _, aux_e = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Forward)   # forward-only: ends are not shrunk
    ba.host.set_list_slicing(0.25, ctx)
    try:
        _, aux_s = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Forward, sliced=True)
    finally:
        ba.host.set_list_slicing(0.0, ctx)
    ge, gs = util.u32(aux_e.compact_gid_from_isect), util.u32(aux_s.compact_gid_from_isect)
    oe, os_ = util.u32(aux_e.tile_offsets).reshape(-1, 2), util.u32(aux_s.tile_offsets).reshape(-1, 2)
    cum = util.u32(aux_e.cum_tiles_hit)
    n0 = int(np.searchsorted(cum, aux_s.list_budget, side="right"))   # splats whose slot range ends within the budget
    total = 0
    for t in range(oe.shape[0]):
        a, b = ge[oe[t, 0]:oe[t, 1]], gs[os_[t, 0]:os_[t, 1]]
        assert np.array_equal(a[:len(b)], b)
        assert np.all(b < n0) and (len(b) == len(a) or a[len(b)] >= n0)
        total += len(b)
    assert total == (int(cum[n0 - 1]) if n0 else 0) <= aux_s.list_budget


def test_automatic_cuts_follow_the_views_previous_frame(dev):
    """automatic mode = per-tile depth cuts from the view's last frame: a view's first frame is rendered with complete lists (and
    seeds the table); the second lists, per tile, what the first needed + a margin — a fraction of the pairs, no far pass; a scene
    whose tiles never saturate keeps complete lists (nothing to cut)"""
    import brush_amd as ba
    n, w, h = 60000, 320, 208
    ctx = ba.Context(dev)
    try:
        cam = util.hip_camera(ba, synth.default_camera_params(w, h))
        sc, _ = _scene(n, w, h, 0x56, scales=(0.03, 0.3))
        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
        lib = ctx.lib
        img_e, aux_e = _reference_render(ba, ctx, spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward)
        img0, aux0 = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, sliced=True)
        assert lib.bh_last_list_share(ctx._h) == 1.0 and aux0.tile_offsets_far is None        # no history: complete lists
        assert torch.equal(aux0.compact_gid_from_isect, aux_e.compact_gid_from_isect)
        img1, aux1 = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, sliced=True)
        share1 = lib.bh_last_list_share(ctx._h)
        assert torch.equal(img0, img1) and torch.equal(img_e, img1) and torch.equal(aux_e.visible, aux1.visible)
        assert aux1.tile_offsets_far is not None and share1 < 0.8, share1
        assert int(lib.bh_far_slices_queued(ctx._h)) == 0                                       # the forecast held: no far pass
        near, far = ba.last_list_counts(ctx)
        assert far == 0 and near == aux1.list_budget and abs(near - share1 * aux1.num_intersections) <= 1.0 + 1e-6 * aux1.num_intersections
        assert aux1.num_visible == aux_e.num_visible and 0 < aux1.num_listed_splats < aux1.num_visible   # only the splats that own a listed pair are numbered
        # every tile's near list = the exact tile list's prefix up to the cut, and holds everything the tile blended
        le, l1 = _blended_lists(aux_e), _blended_lists(aux1)
        for t, (a, b) in enumerate(zip(le, l1)):
            _assert_same_blended_list(a, b, "tile %d" % t)
        # what was listed: >= what was blended, and not much more than the margin allows on the whole
        blended = sum(len(x) for x in le)
        assert blended <= near <= 4 * blended + 256 * len(le), (blended, near)
        # a scene that does not saturate: nothing to cut, the lists stay complete
        sc2, _ = _scene(n, w, h, 0x56, opacity=(0.01, 0.03), scales=(0.005, 0.02))
        spl2 = ba.Splats(sc2["transforms"], sc2["sh"], sc2["raw_opac"], device=dev)
        ba.set_view_id(7, ctx)
        ref2, _ = ba.render_splats(spl2, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx)
        ba.render_splats(spl2, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, sliced=True)
        img3, aux3 = ba.render_splats(spl2, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, sliced=True)
        assert torch.equal(img3, ref2)
        assert lib.bh_last_list_share(ctx._h) > 0.9
    finally:
        ctx.close()


@pytest.mark.parametrize("sh_degree", [0, 3])
def test_train_step_sliced_equals_exact(dev, sh_degree):
    """the default (sliced) step and the exact_lists step from the same state: same loss, counts and update"""
    import brush_amd as ba
    n, w, h = 20000, 256, 160
    sc, cp = _scene(n, w, h, 0x57, sh_degree=sh_degree)
    gt = synth.synthetic_gt_packed(w, h)
    cam = util.hip_camera(ba, cp)
    res = {}
    for exact in (True, False):
        cfg = ba.TrainConfig(exact_lists=exact)
        tr = ba.SplatTrainer(cfg, median_scene_scale=3.0)
        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
        batch = ba.SceneBatch(torch.from_numpy(gt.view(np.int32)).to(dev), cam)
        losses = []
        for _ in range(3):
            tr.step(batch, spl, background=(0.1, 0.1, 0.1))
            losses.append(tr.stats().loss)
        res[exact] = (losses, spl.transforms.cpu().numpy(), spl.sh_coeffs.cpu().numpy(), spl.raw_opacities.cpu().numpy(),
                      tr.state["refine_weight_norm"].cpu().numpy(), tr.state["vis_weight"].cpu().numpy(), tr.stats())
    le, ls = res[True][0], res[False][0]
    assert res[True][6].num_visible == res[False][6].num_visible and res[True][6].num_intersections == res[False][6].num_intersections
    assert all(abs(a - b) <= 1e-6 * max(1.0, abs(a)) for a, b in zip(le, ls))
    cfg = ba.TrainConfig()
    util.assert_adam_close(res[True][1][:, 3:7], res[False][1][:, 3:7], cfg.lr_rotation, 3, "rotation")
    util.assert_adam_close(res[True][1][:, 7:10], res[False][1][:, 7:10], cfg.lr_scale, 3, "scale")
    util.assert_adam_close(res[True][3], res[False][3], cfg.lr_opac, 3, "opacity")
    util.assert_adam_close(res[True][2], res[False][2], cfg.lr_coeffs_dc, 3, "sh")
    assert np.mean(res[True][5] != res[False][5]) <= 2e-3


@pytest.mark.parametrize("mode", ["ids", "camera_hash", "shared_table"])
def test_alternating_views_with_and_without_view_ids(dev, mode):
    """a shallow and a deep view in turn.  With view ids every view has its own per-tile cuts: after each view's first frame the
    near lists are a fraction of the pairs and no far pass runs.  WITHOUT ids (the reference's SceneBatch carries none,
    brush-dataset/src/scene.rs:138-147) the table is keyed by the camera itself: the same shares, frame for frame.  Only with the A/B
    knob BH_NO_VIEW_HASH do the two views share one table (round 4's behaviour): the deep view's tiles are cut too early every time,
    a second attempt finishes them (same image), and repeated misses put the table on hold."""
    import os
    import brush_amd as ba
    n, w, h = 60000, 320, 208
    if mode == "shared_table":
        os.environ["BH_NO_VIEW_HASH"] = "1"
    try:
        ctx = ba.Context(dev)   # (knobs are read at bh_create)
    finally:
        os.environ.pop("BH_NO_VIEW_HASH", None)
    try:
        sc, cp = _scene(n, w, h, 0x59, scales=(0.03, 0.3))
        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
        near_cam = util.hip_camera(ba, cp)
        far = dict(cp)
        far["pos"] = (0.0, 0.0, -6.0)          # from further away the splats are smaller on screen: tiles saturate deeper in the list
        far_cam = util.hip_camera(ba, far)
        ref = {}
        for name, cam in (("near", near_cam), ("far", far_cam)):
            ref[name] = _reference_render(ba, ctx, spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward)
        queued, shares = [], []
        for i in range(16):
            name, cam = (("near", near_cam), ("far", far_cam))[i % 2]
            ba.set_view_id(1 + i % 2 if mode == "ids" else 0, ctx)
            img, aux = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, sliced=True)
            assert torch.equal(img, ref[name][0]) and torch.equal(aux.visible, ref[name][1].visible), (i, name)
            queued.append(int(ctx.lib.bh_far_slices_queued(ctx._h)))
            shares.append(float(ctx.lib.bh_last_list_share(ctx._h)))
        if mode != "shared_table":
            assert shares[0] == 1.0 and shares[1] == 1.0 and max(shares[2:]) < 0.9, shares
            assert queued[-1] == 0, queued
            _ALTERNATING_SHARES[mode] = shares
            if len(_ALTERNATING_SHARES) == 2:   # keyed by id or by camera: the same tables, so the same near lists
                assert _ALTERNATING_SHARES["ids"] == _ALTERNATING_SHARES["camera_hash"]
        else:
            assert queued[-1] >= 1, queued                    # the shared table mispredicts (the image is right all the same)
    finally:
        ctx.close()


_ALTERNATING_SHARES = {}


def test_blank_background_is_cut_like_any_other_frame(dev):
    """an object-centric frame: the outer tiles are empty and never saturate.  Per-tile cuts do not care: those tiles keep their
    (empty or short) lists whole, the object's tiles are cut where they saturated — no far pass, the exact path's image."""
    import brush_amd as ba
    n, w, h = 60000, 320, 208
    ctx = ba.Context(dev)
    try:
        sc = synth.make_scene(n, 0x5A, log_scale_range=(math.log(0.02), math.log(0.1)),
                              tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w), spread=0.45)
        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
        cam = util.hip_camera(ba, synth.default_camera_params(w, h))
        ref, aux_e = _reference_render(ba, ctx, spl, cam, (w, h), (0.2, 0.2, 0.2), ba.RasterPass.Backward)
        assert float(ref[..., 3].min()) == 0.0                        # blank border: those tiles never saturate
        shares = []
        for _ in range(6):
            img, aux = ba.render_splats(spl, cam, (w, h), (0.2, 0.2, 0.2), ba.RasterPass.Backward, ctx=ctx, sliced=True)
            assert torch.equal(img, ref) and torch.equal(aux.visible, aux_e.visible)
            shares.append(float(ctx.lib.bh_last_list_share(ctx._h)))
        assert shares[0] == 1.0 and max(shares[1:]) < 0.95, shares
        assert int(ctx.lib.bh_far_slices_queued(ctx._h)) == 0
    finally:
        ctx.close()


def test_a_view_that_changed_behind_its_cuts_is_rendered_again_with_complete_lists(dev):
    """the forecast fails on purpose: between two frames of the same view every splat turns nearly transparent, so tiles that
    saturated after a few splats now need their whole lists.  The near pass finds live tiles behind cut lists, and the frame is
    rendered a second time with complete lists (only the splats owning a near pair had been sorted: there is nothing to
    continue from) — image, visible flags, blended lists and gradients are the exact path's, and the table is right afterwards."""
    import brush_amd as ba
    n, w, h = 30000, 320, 208
    ctx = ba.Context(dev)
    try:
        sc, cp = _scene(n, w, h, 0x5B, scales=(0.03, 0.3), sh_degree=1)
        cam = util.hip_camera(ba, cp)
        bg = (0.1, 0.3, 0.2)
        ba.set_view_id(3, ctx)
        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
        ba.render_splats(spl, cam, (w, h), bg, ba.RasterPass.Backward, ctx=ctx, sliced=True)    # seeds the view's table
        thin = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"] - 4.0, device=dev)          # opacities / ~50: nothing saturates early any more
        rng = np.random.default_rng(9)
        v_out = torch.from_numpy(rng.normal(size=(h, w, 4)).astype(np.float32)).to(dev)
        F = ba.Context(dev)
        try:
            ex = ba.render_splats_bwd(thin, cam, (w, h), bg, v_out, ctx=F)
        finally:
            F.close()
        q0 = int(ctx.lib.bh_far_slices_queued(ctx._h))
        sl = ba.render_splats_bwd(thin, cam, (w, h), bg, v_out, ctx=ctx, sliced=True)
        assert int(ctx.lib.bh_far_slices_queued(ctx._h)) == q0 + 1 and sl["aux"].tile_offsets_far is None   # the second attempt's (exact) lists
        near, far = ba.last_list_counts(ctx)
        assert far == 0 and near == sl["aux"].num_intersections and sl["aux"].num_listed_splats == sl["aux"].num_visible
        assert torch.equal(ex["img"], sl["img"]) and torch.equal(ex["aux"].visible, sl["aux"].visible)
        for t, (a, b) in enumerate(zip(_blended_lists(ex["aux"]), _blended_lists(sl["aux"]))):
            _assert_same_blended_list(a, b, "tile %d" % t)
        _grads_close(sl, ex, "second attempt after a failed forecast")
        # ... and the next frame of the (now thin) view is forecast correctly: no far pass, same image
        img2, aux2 = ba.render_splats(thin, cam, (w, h), bg, ba.RasterPass.Backward, ctx=ctx, sliced=True)
        assert torch.equal(img2, ex["img"]) and int(ctx.lib.bh_far_slices_queued(ctx._h)) == q0 + 1
    finally:
        ctx.close()


def test_view_tables_survive_eviction_and_resolution_changes(dev):
    """the ctx keeps the 4096 most recently used view tables: a view that was evicted is simply seeded again; a view id that comes
    back at another resolution (another tile grid) gets a fresh table — the images are the exact path's throughout"""
    import brush_amd as ba
    n, w, h = 6000, 160, 112
    sc, cp = _scene(n, w, h, 0x5E, scales=(0.03, 0.3))
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    cam = util.hip_camera(ba, cp)
    import os
    # (the adaptive margin grows with the number of views between two visits — thousands here: pin it, this test is about the tables)
    os.environ["BH_CUT_MARGIN_FIXED"] = "1"
    try:
        ctx = ba.Context(dev)
    finally:
        del os.environ["BH_CUT_MARGIN_FIXED"]
    try:
        ref, _ = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Forward, ctx=ctx)
        ref2, _ = ba.render_splats(spl, cam, (w * 2, h * 2), (0, 0, 0), ba.RasterPass.Forward, ctx=ctx)
        for vid in range(1, 4301):           # more ids than the ctx keeps tables for
            ba.set_view_id(vid, ctx)
            img, _ = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Forward, ctx=ctx, sliced=True, copy=(vid % 500 == 0))
            if vid % 500 == 0:
                assert torch.equal(img, ref), vid
        shares = []
        for vid in (1, 4300, 4300, 1, 1):    # 1 was evicted long ago: seeded again, then cut; 4300 still has its table
            ba.set_view_id(vid, ctx)
            img, _ = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Forward, ctx=ctx, sliced=True)
            assert torch.equal(img, ref), vid
            shares.append(float(ctx.lib.bh_last_list_share(ctx._h)))
        assert shares[0] == 1.0 and shares[1] < 1.0 and shares[4] < 1.0, shares
        for size, want in (((w * 2, h * 2), ref2), ((w, h), ref), ((w * 2, h * 2), ref2), ((w * 2, h * 2), ref2)):   # one id, two tile grids
            ba.set_view_id(7, ctx)
            img, _ = ba.render_splats(spl, cam, size, (0, 0, 0), ba.RasterPass.Forward, ctx=ctx, sliced=True)
            assert torch.equal(img, want), size
    finally:
        ctx.close()


def test_small_frames_keep_complete_lists_by_default(dev):
    """bh_set_list_cut_threshold: a view whose last frame had fewer pairs than the threshold (default 1.5 M) is not cut"""
    import brush_amd as ba
    n, w, h = 30000, 320, 208
    sc, cp = _scene(n, w, h, 0x5D, scales=(0.03, 0.3))
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    cam = util.hip_camera(ba, cp)
    ctx = ba.Context(dev)
    try:
        ctx.check(ctx.lib.bh_set_list_cut_threshold(ctx._h, 1500000))
        shares = []
        for _ in range(3):
            _, aux = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, sliced=True)
            shares.append(float(ctx.lib.bh_last_list_share(ctx._h)))
        assert aux.num_intersections < 1500000 and shares == [1.0, 1.0, 1.0]
        ctx.check(ctx.lib.bh_set_list_cut_threshold(ctx._h, 0))
        ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, sliced=True)
        assert float(ctx.lib.bh_last_list_share(ctx._h)) < 1.0
    finally:
        ctx.close()


@pytest.mark.parametrize("sh_degree", [0, 2])
def test_train_steps_over_cycling_views_with_ids_equal_exact_lists(dev, sh_degree):
    """three views in turn, five rounds: the default step (per-tile cuts keyed by view id) and the exact_lists step follow the same
    trajectory; after each view's first visit the near lists are a fraction of the pairs"""
    import brush_amd as ba
    n, w, h = 20000, 256, 160
    sc, cp = _scene(n, w, h, 0x5C, sh_degree=sh_degree)
    cams = []
    for k in range(3):
        c = dict(cp)
        c["pos"] = (0.5 * k - 0.5, 0.0, -1.5 * k)
        cams.append(util.hip_camera(ba, c))
    gts = [torch.from_numpy(synth.synthetic_gt_packed(w, h, seed=3 + k).view(np.int32)).to(dev) for k in range(3)]
    res = {}
    for exact in (True, False):
        ctx = ba.Context(dev)
        try:
            cfg = ba.TrainConfig(exact_lists=exact)
            tr = ba.SplatTrainer(cfg, median_scene_scale=3.0, ctx=ctx)
            spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
            losses, shares = [], []
            for it in range(15):
                k = it % 3
                tr.step(ba.SceneBatch(gts[k], cams[k], view_id=k + 1), spl, background=(0.1, 0.1, 0.1))
                losses.append(tr.stats(ctx).loss)
                shares.append(float(ctx.lib.bh_last_list_share(ctx._h)))
            res[exact] = (losses, spl.transforms.cpu().numpy(), spl.sh_coeffs.cpu().numpy(), spl.raw_opacities.cpu().numpy(), shares,
                          int(ctx.lib.bh_far_slices_queued(ctx._h)))
        finally:
            ctx.close()
    assert all(abs(a - b) <= 1e-6 * max(1.0, abs(a)) for a, b in zip(res[True][0], res[False][0]))
    cfg = ba.TrainConfig()
    util.assert_adam_close(res[True][1][:, 3:7], res[False][1][:, 3:7], cfg.lr_rotation, 15, "rotation")
    util.assert_adam_close(res[True][1][:, 7:10], res[False][1][:, 7:10], cfg.lr_scale, 15, "scale")
    util.assert_adam_close(res[True][3], res[False][3], cfg.lr_opac, 15, "opacity")
    util.assert_adam_close(res[True][2], res[False][2], cfg.lr_coeffs_dc, 15, "sh")
    shares = res[False][4]
    assert shares[:3] == [1.0, 1.0, 1.0] and max(shares[3:]) < 1.0, shares
    assert res[False][5] <= 3, "far passes: %d" % res[False][5]


# ---------------------------------------------------------------------------------------------------------------
# the slicing state machine under random call sequences (include/brush_hip.h: "Results do not depend on the choice")
# ---------------------------------------------------------------------------------------------------------------
def _fuzz_scenes(dev):
    import brush_amd as ba
    w, h = 208, 160
    tans = (math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w)
    mk = lambda n, seed, **kw: synth.make_scene(n, seed, tan_half_fov=tans, **kw)  # noqa: E731
    raw = {
        "saturating": mk(9000, 0x61, log_scale_range=(math.log(0.03), math.log(0.3))),
        "non_saturating": mk(9000, 0x62, log_scale_range=(math.log(0.01), math.log(0.05)), opacity_range=(0.01, 0.05)),
        "blank_background": mk(9000, 0x63, log_scale_range=(math.log(0.02), math.log(0.15)), spread=0.45),
        "forty": mk(40, 0x64, log_scale_range=(math.log(0.05), math.log(0.4))),
        "sh2": mk(5000, 0x65, sh_degree=2, log_scale_range=(math.log(0.03), math.log(0.3))),
        "empty": dict(transforms=np.zeros((0, 10), np.float32), sh=np.zeros((0, 1, 3), np.float32), raw_opac=np.zeros((0,), np.float32)),
    }
    return raw, w, h


def _fuzz_camera(ba, rng, w, h):
    cp = synth.default_camera_params(w, h)
    cp["pos"] = (float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.3, 0.3)), float(rng.choice([0.0, 0.0, -3.0, 1.0])))
    cp["rot_xyzw"] = util.quat_from_axis_angle((0, 1, 0), float(rng.uniform(-0.2, 0.2)))
    return util.hip_camera(ba, cp)


def _grads_close(a, b, what):
    for k in ("v_transforms", "v_sh_coeffs", "v_raw_opacities", "v_refine_weight"):
        x, y = a[k].cpu().numpy().reshape(-1), b[k].cpu().numpy().reshape(-1)
        scale = max(float(np.abs(y).max()), 1e-20)
        # the float atomics' order only.  (3e-5, not 5e-6: two FRESH contexts given the same call differ by up to 2.5e-6 of the
        # largest gradient in one ill-conditioned element of op 110 — a sum of many pixel terms that nearly cancel — and a context
        # with history by up to 7e-6 in the same element, both with complete lists; scripts/micro/grad_noise_probe.py measures
        # 1e-7 .. 9e-7 for ordinary elements, with and without per-tile cuts.  A missing or doubled pair shows up at 1e-3 and more.)
        assert float(np.abs(x - y).max()) <= 3e-5 * scale, (what, k, float(np.abs(x - y).max()) / scale)


def test_random_call_sequences_on_one_ctx_equal_fresh_contexts(dev):
    """One long random walk over the public calls on ONE ctx — renders (exact / sliced x Forward / Backward / SmoothCutoff, whole
    frames and strips), backwards, train steps (with and without an exchange hook), bh_set_list_slicing, RETAINED forwards that are
    replayed (bh_render_backward_saved) two ops later and released, changing scenes
    (saturating, non-saturating, blank background, 40 splats, empty) and cameras: every result equals the same call on a FRESH
    ctx (bit for bit: images, counts, flags; gradients to the float atomics' order; updates to Adam's sign flips on noise).
    The host-side slicing state (far_job.pending, gate_learn, far_direct, need_hint, last_one_slice) may only change WHEN work
    is done, never WHAT comes out."""
    import brush_amd as ba
    from brush_amd import _ffi
    rng = np.random.default_rng(0xF022)
    raw, w, h = _fuzz_scenes(dev)
    names = list(raw)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h).view(np.int32)).to(dev)
    noop_hook = _ffi.GRAD_HOOK(lambda _u, _p, _c: 0)    # one rank: the sum over the ranks is the buffer itself
    A = ba.Context(dev)
    n_ops = 420
    kinds = {"render": 0, "backward": 0, "step": 0, "slicing": 0, "retain": 0}
    held = []   # retained render nodes of A (bh_render_retain): each is replayed LATER, after other calls ran, and must still give its own gradients

    def retire(node_rec):
        node, v_out, ref, what = node_rec
        got = node.backward(v_out)
        _grads_close(got, ref, ("retained node", what))
        assert torch.equal(node.img, ref["img"]), what
        node.release()

    try:
        for it in range(n_ops):
            op = rng.choice(["render", "render", "backward", "step", "slicing", "retain"])
            kinds[op] += 1
            if op == "slicing":
                share = float(rng.choice([0.0, 0.0, 0.02, 0.1, 0.4, 1.0]))
                ba.set_list_slicing(share, A)
                cur_share = share
                continue
            vid = int(rng.integers(0, 4))     # 0 = no id; the same id meets different scenes and cameras: forecasts fail, results must not
            ba.set_view_id(vid, A)
            name = names[int(rng.integers(len(names)))]
            if name == "empty" and op != "render":   # (the Python mirror cannot hand over zero-sized output tensors)
                name = "forty"
            sc = raw[name]
            cam = _fuzz_camera(ba, rng, w, h)
            bg = tuple(float(x) for x in rng.choice([0.0, 0.2, 0.7], 3))
            sliced = bool(rng.integers(2))
            F = ba.Context(dev)
            try:
                if 'cur_share' in locals():
                    ba.set_list_slicing(cur_share, F)       # the SHARE is an input; the history is what differs
                if op == "retain":
                    # a forward on A whose saved state must outlive whatever comes next (two render nodes alive; an eval render or a
                    # train step between a forward and its backward): reference gradients from a fresh ctx NOW, replay on A later
                    while len(held) >= 2:
                        retire(held.pop(0))
                    v_out = torch.from_numpy((rng.normal(size=(h, w, 4)) / (h * w)).astype(np.float32)).to(dev)
                    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
                    ref = ba.render_splats_bwd(spl, cam, (w, h), bg, v_out, ctx=F)
                    node = ba.render_splats_diff(spl, cam, (w, h), bg, ctx=A, retain=True, sliced=sliced)
                    held.append((node, v_out, ref, (it, name, sliced)))
                elif op == "render":
                    pass_ = [ba.RasterPass.Forward, ba.RasterPass.Backward, ba.RasterPass.BackwardSmoothCutoff][int(rng.integers(3))]
                    rows = None
                    if rng.integers(4) == 0:
                        a = int(rng.integers(0, (h + 15) // 16 - 1))
                        rows = (a, int(rng.integers(a + 1, (h + 15) // 16 + 1)))
                    outs = []
                    for ctx in (A, F):
                        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
                        img, aux = ba.render_splats(spl, cam, (w, h), bg, pass_, ctx=ctx, tile_rows=rows, sliced=sliced)
                        if rows is not None:
                            img = img[rows[0] * 16:min(rows[1] * 16, h)]
                        outs.append((img, aux))
                    (ia, xa), (ib, xb) = outs
                    what = (it, op, name, str(pass_), rows, sliced)
                    assert torch.equal(ia, ib), what
                    assert (xa.num_visible, xa.num_intersections) == (xb.num_visible, xb.num_intersections), what
                    assert torch.equal(xa.max_radius, xb.max_radius), what
                    # (the compact numbering of a frame with per-tile cuts covers only the splats that own a listed pair: a
                    #  sub-sequence of the other context's full depth order)
                    ga, gb = util.u32(xa.global_from_compact_gid), util.u32(xb.global_from_compact_gid)
                    small, big = (ga, gb) if ga.size <= gb.size else (gb, ga)
                    pos = np.full(int(big.max()) + 1 if big.size else 1, -1, np.int64)
                    pos[big] = np.arange(big.size)
                    assert small.size == 0 or (np.all(pos[small] >= 0) and np.all(np.diff(pos[small]) > 0)), what
                    if pass_.bwd_info():
                        assert torch.equal(xa.visible, xb.visible), what
                        for t, (la, lb) in enumerate(zip(_blended_lists(xa), _blended_lists(xb))):
                            # the two contexts may have cut the slices elsewhere: the same blended list up to useless tail entries
                            assert (len(la) == 0) == (len(lb) == 0) and (len(la) == 0 or la[-1] == lb[-1]), (what, t)
                elif op == "backward":
                    v_out = torch.from_numpy((rng.normal(size=(h, w, 4)) / (h * w)).astype(np.float32)).to(dev)
                    smooth = bool(rng.integers(4) == 0)
                    pass_ = ba.RasterPass.BackwardSmoothCutoff if smooth else ba.RasterPass.Backward
                    res = []
                    for ctx in (A, F):
                        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
                        res.append(ba.render_splats_bwd(spl, cam, (w, h), bg, v_out, pass_, ctx=ctx, sliced=sliced))
                    assert torch.equal(res[0]["img"], res[1]["img"]), (it, op, name)
                    _grads_close(res[0], res[1], (it, op, name, sliced, smooth))
                else:   # a train step from the same state on both contexts
                    with_hook = bool(rng.integers(3) == 0)
                    exact = bool(rng.integers(4) == 0)
                    finals = []
                    for ctx in (A, F):
                        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
                        tr = ba.SplatTrainer(ba.TrainConfig(exact_lists=exact), median_scene_scale=3.0, ctx=ctx, sparse_exchange=bool(it % 2))
                        if with_hook:
                            tr.pg, tr._hook, tr._world = "one rank", noop_hook, 1
                        tr.step(ba.SceneBatch(gt, cam, view_id=vid), spl, background=bg)
                        st = tr.stats(ctx)
                        finals.append((st, spl.transforms.cpu().numpy(), spl.sh_coeffs.cpu().numpy(), spl.raw_opacities.cpu().numpy(),
                                       tr.state["vis_weight"].cpu().numpy(), tr.state["max_screen_size"].cpu().numpy()))
                    (sa, ta, ha, oa, va, ra), (sb, tb, hb, ob, vb, rb) = finals
                    what = (it, op, name, with_hook, exact)
                    assert (sa.num_visible, sa.num_intersections) == (sb.num_visible, sb.num_intersections), what
                    assert abs(sa.loss - sb.loss) <= 1e-6 * max(1.0, abs(sb.loss)), what
                    assert np.array_equal(va, vb) and np.array_equal(ra, rb), what
                    cfg = ba.TrainConfig()
                    if sc["transforms"].shape[0] >= 5000:
                        util.assert_adam_close(ta[:, 3:7], tb[:, 3:7], cfg.lr_rotation, 1, what)
                        util.assert_adam_close(ta[:, 7:10], tb[:, 7:10], cfg.lr_scale, 1, what)
                        util.assert_adam_close(oa, ob, cfg.lr_opac, 1, what)
                        util.assert_adam_close(ha, hb, cfg.lr_coeffs_dc, 1, what)
                    else:   # 40 splats: one +-lr sign flip on a noise gradient is already 0.6 % of the entries - bound the size only
                        assert np.abs(ta[:, 3:7] - tb[:, 3:7]).max() <= 2.1 * cfg.lr_rotation and np.abs(oa - ob).max() <= 2.1 * cfg.lr_opac, what
            finally:
                F.close()
        while held:
            retire(held.pop(0))
        assert min(kinds.values()) >= 30, kinds
    finally:
        A.close()


def test_a_step_that_fails_behind_its_near_slice_leaves_the_ctx_usable(dev):
    """bh_train_step queues the near slice, defers the far-slice decision behind the loss kernels (far_job.pending) and THEN fails
    (an injected BH_ERR_OOM in front of the loss: BH_TEST_FAIL_LOSS_AT).  The step must not count (a failed step applied no
    update), and everything that follows on the ctx — the retried step, a render, a backward — must equal a fresh ctx's."""
    import os
    import brush_amd as ba
    raw, w, h = _fuzz_scenes(dev)
    sc = raw["saturating"]
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h).view(np.int32)).to(dev)
    cam = util.hip_camera(ba, synth.default_camera_params(w, h))
    bg = (0.1, 0.2, 0.3)
    from brush_amd import _ffi
    os.environ["BH_TEST_FAIL_LOSS_AT"] = "3"
    try:
        A = ba.Context(dev, lib=_ffi.load_test_hooks())   # (fault injection exists in the -DBH_TEST_HOOKS build only)
    finally:
        del os.environ["BH_TEST_FAIL_LOSS_AT"]
    F = ba.Context(dev)
    try:
        runs = {}
        for key, ctx in (("A", A), ("F", F)):
            spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
            tr = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0, ctx=ctx)
            losses, failed = [], 0
            while tr.step_count < 4:
                # steps 1-2: automatic share, every tile finishes in the near slice -> the host decides (far_job.pending) from now on;
                # from step 3 (the one that fails on A): a near slice that leaves most tiles unsaturated -> the pending decision is "yes"
                ba.set_list_slicing(0.0 if tr.step_count < 2 else 0.03, ctx)
                try:
                    tr.step(ba.SceneBatch(gt, cam), spl, background=bg)
                    losses.append(tr.stats(ctx).loss)
                except ba.BrushHipError as e:
                    assert "injected failure" in str(e)
                    failed += 1
                    assert tr.step_count == 2, "a failed step must not count"
            img, aux = ba.render_splats(spl, cam, (w, h), bg, ba.RasterPass.Backward, ctx=ctx, sliced=True)
            runs[key] = (losses, failed, spl.transforms.cpu().numpy(), spl.raw_opacities.cpu().numpy(), img, aux.visible)
        assert runs["A"][1] == 1 and runs["F"][1] == 0
        assert len(runs["A"][0]) == len(runs["F"][0]) == 4
        assert all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(runs["A"][0], runs["F"][0]))
        cfg = ba.TrainConfig()
        util.assert_adam_close(runs["A"][2][:, 3:7], runs["F"][2][:, 3:7], cfg.lr_rotation, 4, "rotation")
        util.assert_adam_close(runs["A"][3], runs["F"][3], cfg.lr_opac, 4, "opacity")
    finally:
        A.close()
        F.close()


def test_a_failed_step_drops_its_per_tile_cut_job_and_never_replays_freed_parameters(dev):
    """a per-tile-cut frame whose far decision is still pending when bh_train_step fails keeps the CALLER's
    parameter pointers in its job; collecting that job later (bh_sync, the next forward) replayed a whole forward from them — a
    device use-after-free once the caller has released the tensors (a refine does).  The failing step must drop the job: nothing is
    replayed (bh_far_slices_queued does not move), and the ctx then trains a DIFFERENT scene exactly like a fresh ctx."""
    import os
    import brush_amd as ba
    from brush_amd import _ffi
    n, w, h = 30000, 320, 208
    sc, cp = _scene(n, w, h, 0x5B, scales=(0.03, 0.3), sh_degree=0)
    cam = util.hip_camera(ba, cp)
    gt = torch.from_numpy(synth.synthetic_gt_packed(w, h).view(np.int32)).to(dev)
    bg = (0.1, 0.2, 0.3)
    os.environ["BH_TEST_FAIL_LOSS_AT"] = "3"
    try:
        A = ba.Context(dev, lib=_ffi.load_test_hooks())
    finally:
        del os.environ["BH_TEST_FAIL_LOSS_AT"]
    F = ba.Context(dev)
    try:
        tr = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0, ctx=A)
        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
        for _ in range(2):   # seed the view's table, then one cut frame
            tr.step(ba.SceneBatch(gt, cam, view_id=5), spl, background=bg)
        assert float(A.lib.bh_last_list_share(A._h)) < 1.0, "step 2 should have run with per-tile cut lists"
        # step 3: every splat turns nearly transparent (the forecast WILL fail: live tiles behind cut lists) and the step fails behind
        # its forward, with the decision pending
        thin = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"] - 4.0, device=dev)
        q0 = int(A.lib.bh_far_slices_queued(A._h))
        with pytest.raises(ba.BrushHipError, match="injected failure"):
            tr.step(ba.SceneBatch(gt, cam, view_id=5), thin, background=bg)
        assert tr.step_count == 2
        # the caller releases the parameters and poisons the memory they lived in (the caching allocator hands it out again)
        thin.transforms.fill_(float("nan")); thin.sh_coeffs.fill_(float("nan")); thin.raw_opacities.fill_(float("nan"))
        del thin
        A.sync()
        assert int(A.lib.bh_far_slices_queued(A._h)) == q0, "the dropped job must not be replayed"
        # a different scene (other n) on the same ctx == a fresh ctx
        sc2, _ = _scene(20000, w, h, 0x77, scales=(0.03, 0.3), sh_degree=0)
        outs = {}
        for key, ctx in (("A", A), ("F", F)):
            s2 = ba.Splats(sc2["transforms"].copy(), sc2["sh"].copy(), sc2["raw_opac"].copy(), device=dev)
            t2 = ba.SplatTrainer(ba.TrainConfig(), median_scene_scale=3.0, ctx=ctx)
            losses = []
            for k in range(3):
                t2.step(ba.SceneBatch(gt, cam, view_id=5), s2, background=bg)
                losses.append(t2.stats(ctx).loss)
            img, _ = ba.render_splats(s2, cam, (w, h), bg, ba.RasterPass.Backward, ctx=ctx, sliced=True)
            outs[key] = (losses, s2.transforms.cpu().numpy(), s2.raw_opacities.cpu().numpy(), img)
        assert all(np.isfinite(x) for x in outs["A"][0])
        assert all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(outs["A"][0], outs["F"][0]))
        cfg = ba.TrainConfig()
        util.assert_adam_close(outs["A"][1][:, 3:7], outs["F"][1][:, 3:7], cfg.lr_rotation, 3, "rotation")
        util.assert_adam_close(outs["A"][2], outs["F"][2], cfg.lr_opac, 3, "opacity")
    finally:
        A.close()
        F.close()


def test_the_cut_margin_adapts_to_how_long_a_view_is_away_and_to_failed_forecasts(dev):
    """api.hip cut_margin_pct: a view that comes back after many other views gets a deeper margin than one that alternates with a
    single other view (same scene, same camera: a larger near share), and a failed forecast deepens the margins of the frames that
    follow (x 1.5 per miss).  Images stay the exact path's throughout."""
    import brush_amd as ba
    n, w, h = 60000, 320, 208
    sc, cp = _scene(n, w, h, 0x59, scales=(0.03, 0.3))
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    cam = util.hip_camera(ba, cp)
    others = []
    for k in range(40):
        c = dict(cp)
        c["pos"] = (0.05 * (k + 1), 0.0, 0.0)
        others.append(util.hip_camera(ba, c))
    ctx = ba.Context(dev)
    try:
        ref, _ = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx)

        def frame(c, vid):
            ba.set_view_id(vid, ctx)
            img, _ = ba.render_splats(spl, c, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, sliced=True, copy=(c is cam))
            if c is cam:
                assert torch.equal(img, ref)
            return float(ctx.lib.bh_last_list_share(ctx._h))
        # alternating with ONE other view: gap 2
        for _ in range(3):
            s_close = frame(cam, 1)
            frame(others[0], 2)
        # ... then 40 other views between two visits: the table written at a visit carries the margin for the gap it has just seen
        for _ in range(3):
            for k, c in enumerate(others):
                frame(c, 10 + k)
            s_far = frame(cam, 1)
        assert s_close < 1.0 and s_far > s_close * 1.15, (s_close, s_far)
        # a failed forecast (the scene turns nearly transparent between two frames of a view) widens what follows
        thin = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"] - 4.0, device=dev)
        q0 = int(ctx.lib.bh_far_slices_queued(ctx._h))
        ba.set_view_id(1, ctx)
        ba.render_splats(thin, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, sliced=True)
        assert int(ctx.lib.bh_far_slices_queued(ctx._h)) == q0 + 1
    finally:
        ctx.close()
