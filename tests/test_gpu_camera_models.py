"""Camera models on the MI355X: the HIP project kernels with Kannala-Brandt 4,
radial-tangential 8 and thin-prism fisheye lenses vs the oracle, through the C ABI.

Same bar as the pinhole path: cull decisions, counts, depth order, tile assignment, projected
records and images bit-identical (the fixed atan2 polynomial is shared), gradients within
1e-4 * max|g|."""
import math

import numpy as np
import pytest
import torch

from brush_amd import synth
import util
from test_gpu_render import render_both, assert_stagewise_exact, IMG_TOL
from test_gpu_backward import run_both, assert_grads_match

pytestmark = pytest.mark.gpu


def _cam(base, lens, **over):
    model, dist = util.REF_LENSES[lens]
    p = dict(base)
    p.update(model=model, dist=dist)
    p.update(over)
    return p


@pytest.mark.parametrize("lens", ["kb4", "rt8", "tpf"])
@pytest.mark.parametrize("sh_degree,mip", [(0, False), (2, True)])
def test_10k_forward_exact_vs_oracle(dev, oracle_lib, lens, sh_degree, mip):
    import brush_amd as ba
    scene, w, h = synth.config_scene("10k_256", sh_degree)
    cp = _cam(synth.default_camera_params(w, h), lens)
    img, aux, ref = render_both(ba, oracle_lib, dev, scene, cp, w, h, bg=(0.1, 0.2, 0.3), mip=mip)
    assert aux.num_visible > 5000
    assert_stagewise_exact(aux, ref)
    assert np.abs(img.cpu().numpy() - ref.image()).max() <= IMG_TOL


@pytest.mark.parametrize("lens", ["kb4", "rt8", "tpf"])
def test_10k_backward_vs_oracle(dev, oracle_lib, lens):
    import brush_amd as ba
    scene, w, h = synth.config_scene("10k_256", 1)
    cp = _cam(synth.default_camera_params(w, h), lens)
    rng = np.random.default_rng(17)
    v_out = (rng.uniform(-1, 1, (h, w, 4)) / (h * w)).astype(np.float32)
    res, ref = run_both(ba, oracle_lib, dev, scene, cp, w, h, v_out, bg=(0.2, 0.1, 0.4))
    assert np.abs(ref.get("v_transforms")).max() > 0
    assert_grads_match(res, ref)


def test_wide_fisheye_sees_behind_the_image_plane(dev, oracle_lib):
    """A 200-degree KB4 lens: splats at z <= 0 in camera space are rendered (the pinhole gate
    z >= 0.01 does not apply, project_forward.rs:53-61); exact vs the oracle."""
    import brush_amd as ba
    rng = np.random.default_rng(3)
    n = 4000
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    means = (d * rng.uniform(2.0, 6.0, (n, 1))).astype(np.float32)
    tr = np.concatenate([means, rng.uniform(-1, 1, (n, 4)), rng.uniform(math.log(0.03), math.log(0.15), (n, 3))], 1).astype(np.float32)
    scene = dict(transforms=tr, sh=rng.uniform(-1, 1.7, (n, 1, 3)).astype(np.float32), raw_opac=rng.uniform(-1, 3, n).astype(np.float32))
    fov = math.radians(200.0)
    cp = dict(pos=(0.0, 0.0, 0.0), rot_xyzw=(0, 0, 0, 1), fov_x=fov, fov_y=fov, center_uv=(0.5, 0.5), model="kb4", dist=(-0.02, 0.003, -0.0002, 0.0))
    img, aux, ref = render_both(ba, oracle_lib, dev, scene, cp, 200, 200, bg=(0.0, 0.0, 0.0))
    gids = util.u32(aux.global_from_compact_gid)
    assert (means[gids, 2] <= 0.0).sum() > 50, "no splat behind the image plane was rendered"
    assert_stagewise_exact(aux, ref)
    assert np.abs(img.cpu().numpy() - ref.image()).max() <= IMG_TOL
    v_out = np.full((200, 200, 4), 1.0 / (200 * 200 * 4), np.float32)
    res, ref2 = run_both(ba, oracle_lib, dev, scene, cp, 200, 200, v_out)
    assert_grads_match(res, ref2)


def test_fuzz_models_small_scenes(dev, oracle_lib):
    """The reference's camera-model fuzz scenes (finite_diff.rs:723-800, 1170-1240), HIP vs oracle,
    with the smooth-cutoff pass the reference's gradient tests use."""
    import brush_amd as ba
    for seed in range(12):
        for cp, size in ((util.random_camera_with_model(seed), 32), (util.heavy_distortion_camera(seed), 48)):
            rng = util.Sm64((seed + 0xC0DEBEEF) & 0xFFFFFFFFFFFFFFFF)
            scene = util.random_scene(seed, rng.usize_in(3, 9))
            v_out = np.full((size, size, 4), 1.0 / (size * size * 4), np.float32)
            res, ref = run_both(ba, oracle_lib, dev, scene, cp, size, size, v_out, pass_=ba.RasterPass.BackwardSmoothCutoff)
            assert np.abs(res["img"].cpu().numpy() - ref.image()).max() <= IMG_TOL, (seed, cp["model"])
            assert_grads_match(res, ref)


def test_camera_setup_host_math_matches_oracle_for_every_model(oracle_lib):
    """bh_camera_setup_model vs the oracle's restatement of camera.rs (also runs in the CPU suite, test_abi)."""
    import brush_amd as ba
    for lens in util.REF_LENSES:
        p = _cam(dict(pos=(0.3, -0.2, 1.5), rot_xyzw=util.quat_from_axis_angle((0.3, -1.0, 0.2), 1.1), fov_x=1.1, fov_y=0.7, center_uv=(0.45, 0.55)), lens)
        a = util.hip_camera(ba, p).uniforms((640, 360))
        b = oracle_lib.camera(img_w=640, img_h=360, **p)
        for f, _ in b._fields_:
            va, vb = getattr(a, f), getattr(b, f)
            assert (list(va) == list(vb)) if hasattr(va, "__len__") else (va == vb), (lens, f)


@pytest.mark.parametrize("lens", ["kb4", "rt8"])
def test_train_step_with_lens_matches_oracle_trainer(dev, oracle_lib, lens):
    """bh_train_step with a distorted camera == the oracle trainer's step (loss, parameters)."""
    import brush_amd as ba
    scene, w, h = synth.config_scene("10k_256", 0)
    cp = _cam(synth.default_camera_params(w, h), lens)
    gt = synth.synthetic_gt_packed(w, h, seed=3)
    cfg = ba.TrainConfig(mean_noise_weight=0.0, background_noise_strength=0.0)
    spl = ba.Splats(scene["transforms"], scene["sh"], scene["raw_opac"], device=dev)
    tr = ba.SplatTrainer(cfg, median_scene_scale=5.0)
    batch = ba.SceneBatch(torch.from_numpy(gt.view(np.int32)).to(dev), util.hip_camera(ba, cp))
    otr = util.OracleTrainer(oracle_lib, cfg, median_scene_scale=5.0)
    sc = {k: v.copy() for k, v in scene.items()}
    p = {k: v for k, v in cp.items() if k not in ("img_w", "img_h")}
    ocam = oracle_lib.camera(img_w=w, img_h=h, **p)
    for step in range(2):
        tr.step(batch, spl)
        st = tr.stats()
        ost = otr.step(sc, ocam, gt, (0.0, 0.0, 0.0))
        assert st.num_visible == ost["num_visible"] and st.num_intersections == ost["num_intersections"]
        assert abs(st.loss - ost["loss"]) <= 1e-5 * max(1.0, abs(ost["loss"]))
        if step > 0:
            continue
        # Adam's first update is lr * g / (|g| + 1e-15) = +-lr: where the gradient itself is summation-order noise
        # (|g| below 1e-5 of the tensor's largest) the sign is arbitrary, elsewhere the parameters must agree closely.
        g = ost["grads"]
        t = spl.transforms.cpu().numpy()
        for sl, lr in ((slice(0, 3), ost["lr_mean"]), (slice(3, 7), cfg.lr_rotation), (slice(7, 10), cfg.lr_scale)):
            solid = np.abs(g["g_tr"][:, sl]) >= 1e-5 * np.abs(g["g_tr"][:, sl]).max()
            d = np.abs(t[:, sl] - sc["transforms"][:, sl])
            assert d[solid].max() <= 0.02 * lr + 1e-7 and d.max() <= 2.1 * lr + 1e-7
            assert solid.mean() > 0.05
        solid = np.abs(g["g_op"][:, 0]) >= 1e-5 * np.abs(g["g_op"]).max()
        d = np.abs(spl.raw_opacities.cpu().numpy() - sc["raw_opac"])
        assert d[solid].max() <= 0.02 * cfg.lr_opac and d.max() <= 2.1 * cfg.lr_opac
