"""Developer tool (GPU box): the fused depth sort on depth distributions that defeat a linear split of the key range — the same
frames with the split at the previous frame's quantiles (option dsort_splitters=1, the default) and with the linear split (0),
checked against the generic radix sort + scan (generic_depth_sort=1).  Prints the forward's time per frame (complete lists,
1 M splats, 1080p; torch events around 20 frames of ONE view after 3 warm-up frames) and whether order / scan / image agree.
    python scripts/dsort_probe.py [n_splats]
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import brush_amd as ba   # noqa: E402
from brush_amd import synth   # noqa: E402
import util   # noqa: E402


def scenes(n):
    rng = np.random.default_rng(11)
    base = synth.make_scene(n, 0xD5, sh_degree=0)
    out = {"uniform_z": base}
    s = {k: v.copy() for k, v in base.items()}
    # a thin shell: 97 % of the splats within 1 % of one depth, the rest spread over the frustum
    z = np.where(rng.random(n) < 0.97, 6.0 + rng.random(n) * 0.06, 2.0 + rng.random(n) * 10.0).astype(np.float32)
    f = z / s["transforms"][:, 2]
    s["transforms"][:, 0] *= f; s["transforms"][:, 1] *= f; s["transforms"][:, 2] = z
    out["thin_shell"] = s
    s = {k: v.copy() for k, v in base.items()}
    s["transforms"][0, 0:3] = (0.0, 0.0, 4.0e5)      # one far outlier stretches the key range
    s["transforms"][0, 7:10] = math.log(1.5e3)
    out["far_outlier"] = s
    s = {k: v.copy() for k, v in base.items()}
    z = np.exp(rng.normal(math.log(5.0), 0.08, n)).astype(np.float32)   # log-normal around one depth
    f = z / s["transforms"][:, 2]
    s["transforms"][:, 0] *= f; s["transforms"][:, 1] *= f; s["transforms"][:, 2] = z
    out["lognormal"] = s
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dev = torch.device("cuda:0")
    w, h = 1920, 1080
    cp = synth.default_camera_params(w, h)
    warm = ba.Context(dev)   # the process's first-use costs (code objects, clocks) are paid here, not by the first line
    wsc = synth.make_scene(200_000, 0xD6, sh_degree=0)
    wspl = ba.Splats(wsc["transforms"], wsc["sh"], wsc["raw_opac"], device=dev)
    for _ in range(30):
        ba.render_splats(wspl, util.hip_camera(ba, cp), (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=warm, copy=False)
    torch.cuda.synchronize()
    warm.close()
    for name, sc in scenes(n).items():
        ref = None
        for opts in ({"generic_depth_sort": 1}, {"dsort_splitters": 0}, {"dsort_splitters": 1}):
            ctx = ba.Context(dev, options=opts)
            spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
            cam = util.hip_camera(ba, cp)
            res = None
            # the context's FIRST frame of the view on its own (no table yet: a sample is sorted first, depth_sort.hip SPL_SAMPLE_STRIDE)
            torch.cuda.synchronize()
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record()
            ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, copy=False)
            f1.record()
            torch.cuda.synchronize()
            first_ms = f0.elapsed_time(f1)
            for _ in range(3):
                img, aux = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx)
            nv = aux.num_visible
            res = (util.u32(aux.global_from_compact_gid)[:nv].copy(), util.u32(aux.cum_tiles_hit)[:nv].copy(), img.cpu().numpy())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward, ctx=ctx, copy=False)
            e1.record()
            torch.cuda.synchronize()
            same = "reference" if ref is None else all(np.array_equal(a, b) for a, b in zip(ref, res))
            if ref is None:
                ref = res
            print("%-12s %-28s visible %8d pairs %9d  forward %.3f ms/frame (first frame of the context: %.3f ms)  equal: %s"
                  % (name, opts, nv, aux.num_intersections, e0.elapsed_time(e1) / 20, first_ms, same), flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
