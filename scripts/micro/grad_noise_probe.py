"""Developer probe (GPU box): run-to-run deviation of the backward's gradients — fresh contexts (complete lists both times: the
float atomics' order only) and ONE context with a view id (the first call seeds the per-tile cuts, the later ones list against them)."""
import sys, math, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import util
from brush_amd import synth
import brush_amd as ba
dev = torch.device('cuda:0')
w, h = 208, 160
tans = (math.tan(math.radians(30)), math.tan(math.radians(30)) * h / w)
sc = synth.make_scene(9000, 0x61, tan_half_fov=tans, log_scale_range=(math.log(0.03), math.log(0.3)))
rng = np.random.default_rng(7)
KEYS = ("v_transforms", "v_sh_coeffs", "v_raw_opacities", "v_refine_weight")
def rel(a, b):
    return {k: float(np.abs(a[k] - b[k]).max()) / max(float(np.abs(b[k]).max()), 1e-20) for k in KEYS}
for trial in range(6):
    cp = synth.default_camera_params(w, h)
    cp["pos"] = (float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.3, 0.3)), float(rng.choice([0.0, 0.0, -3.0, 1.0])))
    cp["rot_xyzw"] = util.quat_from_axis_angle((0, 1, 0), float(rng.uniform(-0.2, 0.2)))
    cam = util.hip_camera(ba, cp)
    v_out = torch.from_numpy((rng.normal(size=(h, w, 4)) / (h * w)).astype(np.float32)).to(dev)
    def run(ctx, sliced):
        spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
        r = ba.render_splats_bwd(spl, cam, (w, h), (0.0, 0.2, 0.7), v_out, ba.RasterPass.Backward, ctx=ctx, sliced=sliced)
        return {k: r[k].cpu().numpy().reshape(-1).copy() for k in KEYS}, r
    F = ba.Context(dev); ref, _ = run(F, False); F.close()
    fresh = {k: 0.0 for k in KEYS}
    for rep in range(6):
        C = ba.Context(dev); g, _ = run(C, bool(rep % 2)); C.close()
        d = rel(g, ref); fresh = {k: max(fresh[k], d[k]) for k in KEYS}
    A = ba.Context(dev); ba.set_view_id(3, A)
    hist = {k: 0.0 for k in KEYS}
    shares = []
    for rep in range(8):
        g, r = run(A, True)
        d = rel(g, ref); hist = {k: max(hist[k], d[k]) for k in KEYS}
        shares.append(round(float(A.lib.bh_last_list_share(A._h)), 3))
    A.close()
    print("trial", trial, "fresh", {k: "%.1e" % v for k, v in fresh.items()}, "one ctx + view id", {k: "%.1e" % v for k, v in hist.items()}, "shares", shares)
