"""The whole loop end to end (BASELINE.json configs[3] stand-in, scripts/train_synthetic.py): synthetic
multi-view images -> SceneLoader -> step (+noise) -> refine (prune / split / decay) — the model must
actually learn the scene: held-out PSNR rises by several dB.  This is the integration check the reference's
own integration tests leave at "does not crash, splats > 0" (crates/brush-bench-test/tests/integration.rs)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


@pytest.mark.parametrize("filter3d", [False, True])
def test_training_learns_the_scene(dev, filter3d):
    import train_synthetic as ts
    args = ts.parse(["--steps", "400", "--res", "128", "--views", "16", "--gt-splats", "2000", "--init-splats", "1000",
                     "--refine-every", "100"] + (["--filter3d"] if filter3d else []))
    out = ts.run(args, log=lambda *_: None)
    assert out["psnr_after"] >= out["psnr_before"] + 4.0, out
    assert out["psnr_after"] >= 20.0, out
    assert len(out["refines"]) == 3 and out["splats"] > 0
