"""Host image -> packed device batch: bh_uploader_* and the SceneLoader mirror.
Byte work: the packed rgba8 words must equal the oracle's restatement of view_to_packed_data
(brush-dataset/src/scene.rs:97-136) bit for bit, for every size incl. ragged tails."""
import math

import numpy as np
import pytest
import torch

from oracle import scene as oscene
from brush_amd import synth
import util

pytestmark = pytest.mark.gpu


def _img(h, w, c, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, c), dtype=np.uint8)


@pytest.mark.parametrize("h,w", [(1, 1), (1, 3), (5, 7), (16, 16), (33, 61), (270, 481), (1080, 1920)])
@pytest.mark.parametrize("c,premul", [(3, False), (4, True), (4, False)])
def test_packed_words_equal_oracle(dev, h, w, c, premul):
    import brush_amd as ba
    up = ba.BatchUploader(1080 * 1920, slots=2)
    img = _img(h, w, c, h * 31 + w + c)
    if c == 4:  # make sure the interesting alphas are present
        img[0, 0, 3], img[-1, -1, 3] = 0, 255
    slot = up.submit(img, premultiply=premul)
    packed, has_alpha = up.acquire(slot)
    want, wa = oscene.view_to_packed_data(img, transparent_alpha=premul)
    got = util.u32(packed)
    up.release(slot)
    assert has_alpha == wa == (c == 4) and got.shape == (h, w)
    assert np.array_equal(got, want)
    up.close()


def test_decode_into_the_pinned_slot_and_ring_reuse(dev):
    """bh_uploader_begin/commit (zero-copy staging) over more submits than slots; every batch intact."""
    import brush_amd as ba
    h, w = 120, 200
    up = ba.BatchUploader(h * w, slots=2)
    for k in range(7):
        img = _img(h, w, 3, k)
        slot, buf = up.map(h * w * 3)
        buf[:] = img.reshape(-1)          # "decoder" writes straight into pinned memory
        up.commit(slot, w, h, 3, False)
        packed, _ = up.acquire(slot)
        got = util.u32(packed).copy()
        up.release(slot)
        assert np.array_equal(got, oscene.view_to_packed_data(img)[0]), k
    up.close()


def test_misuse_returns_errors(dev):
    import brush_amd as ba
    up = ba.BatchUploader(64, slots=2)
    with pytest.raises(ba.BrushHipError):
        up.submit(_img(9, 9, 3, 0))                 # larger than the slot
    with pytest.raises(ba.BrushHipError):
        up.acquire(0)                               # nothing committed
    s0 = up.submit(_img(8, 8, 3, 0))
    s1 = up.submit(_img(8, 8, 4, 1))
    with pytest.raises(ba.BrushHipError):
        up.submit(_img(8, 8, 3, 2))                 # ring full: oldest never acquired
    up.acquire(s0)
    with pytest.raises(ba.BrushHipError):
        up.release(s1)                              # not acquired
    up.release(s0)
    up.acquire(s1)
    up.release(s1)
    assert up.submit(_img(8, 8, 3, 3)) == s0
    with pytest.raises(ba.BrushHipError):
        ba.BatchUploader(64, slots=1)
    up.close()


def test_scene_loader_visits_every_view_once_per_epoch_and_shards(dev):
    import brush_amd as ba
    h, w = 48, 64
    views = [(_img(h, w, 3 if i % 2 else 4, i), ba.Camera(position=(float(i), 0.0, 0.0)), i % 4 == 0) for i in range(10)]
    want = [oscene.view_to_packed_data(v[0], transparent_alpha=not v[2])[0] for v in views]
    ld = ba.SceneLoader(views, seed=11, slots=3)
    for epoch in range(3):
        seen = []
        for _ in range(len(views)):
            b = ld.next_batch()
            i = b.view_index
            assert b.camera.position[0] == float(i) and b.has_alpha == (i % 2 == 0) and b.alpha_is_mask == (i % 4 == 0)
            assert np.array_equal(util.u32(b.img_packed), want[i]), (epoch, i)
            seen.append(i)
        assert sorted(seen) == list(range(10))
        assert seen == ld.epoch_order(epoch)
    assert ld.epoch_order(0) != ld.epoch_order(1)
    ld.close()
    # data-parallel sharding: rank r owns views i % world == r, together the ranks cover the list
    parts = []
    for r in range(2):
        ldr = ba.SceneLoader(views, seed=5, slots=2, rank=r, world=2)
        idx = sorted(ldr.views[ldr.next_batch().view_index][1].position[0] for _ in range(5))
        parts.append(idx)
        ldr.close()
    assert parts[0] == [0.0, 2.0, 4.0, 6.0, 8.0] and parts[1] == [1.0, 3.0, 5.0, 7.0, 9.0]


def test_training_through_the_loader_equals_training_on_resident_batches(dev):
    """Steps fed by the overlapped uploader follow the same trajectory as steps fed with the same views
    uploaded up front (the hand-over is an event wait on the ctx stream, not a host sync).  The backward
    accumulates with float atomics, so two runs agree to rounding, not bit for bit: per-step losses to 1e-5,
    parameters to a small fraction of the accumulated learning rate."""
    import brush_amd as ba
    n, w, h = 3000, 160, 96
    sc = synth.make_scene(n, 0xD7, sh_degree=0, log_scale_range=(math.log(0.02), math.log(0.2)),
                          tan_half_fov=(math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w))
    cp = synth.default_camera_params(w, h)
    cams = [ba.Camera(position=cp["pos"], rotation=util.quat_from_axis_angle((0, 1, 0), 0.03 * k), fov_x=cp["fov_x"], fov_y=cp["fov_y"]) for k in range(4)]
    imgs = [_img(h, w, 3, 100 + k) for k in range(4)]
    views = list(zip(imgs, cams))
    cfg = ba.TrainConfig(mean_noise_weight=0.0)

    def run(loader):
        spl = ba.Splats(sc["transforms"].copy(), sc["sh"].copy(), sc["raw_opac"].copy(), device=dev)
        tr = ba.SplatTrainer(cfg, median_scene_scale=3.0)
        order, losses = [], []
        for step in range(12):
            if loader is not None:
                b = loader.next_batch()
                order.append(b.view_index)
            else:
                i = ORDER[step]
                packed = torch.from_numpy(oscene.view_to_packed_data(imgs[i])[0].view(np.int32)).to(dev)
                b = ba.SceneBatch(packed, cams[i])
            tr.step(b, spl)
            losses.append(tr.stats().loss)
        return spl, order, losses
    ld = ba.SceneLoader(views, seed=3, slots=3)
    spl_a, ORDER, loss_a = run(ld)
    ld.close()
    spl_b, _, loss_b = run(None)
    assert len(set(ORDER)) == 4
    assert np.allclose(loss_a, loss_b, rtol=1e-5, atol=1e-7), (loss_a, loss_b)
    d = (spl_a.transforms - spl_b.transforms).abs()
    assert d[:, 7:10].max().item() <= 0.1 * cfg.lr_scale * 12 and d.mean().item() <= 1e-6
    assert (spl_a.raw_opacities - spl_b.raw_opacities).abs().mean().item() <= 1e-6
