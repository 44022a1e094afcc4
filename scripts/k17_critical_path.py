import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import brush_amd as ba
from brush_amd import synth
dev = torch.device("cuda:0")
for wl in ("1m_1080p", "1m_1080p_lowopac", "1m_1080p_centered"):
    sc, w, h = synth.config_scene(wl, 0)
    cp = synth.default_camera_params(w, h)
    cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
    spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
    ctx = ba.Context(dev)
    v_out = torch.full((h, w, 4), 1e-6, device=dev)
    for _ in range(3):
        res = ba.render_splats_bwd(spl, cam, (w, h), (0, 0, 0), v_out, ctx=ctx)
    ctx.sync(); ctx.profile(1); ctx.profile_fetch()
    for _ in range(5):
        res = ba.render_splats_bwd(spl, cam, (w, h), (0, 0, 0), v_out, ctx=ctx)
    ctx.sync(); prof = ctx.profile_fetch(); ctx.profile(0)
    to = res["aux"].tile_offsets.to(torch.int64)
    work = (to[:, 1] - to[:, 0]).clamp(min=0).cpu().numpy()
    k17 = prof["RasterizeBackwards"][0] / prof["RasterizeBackwards"][1]; k16 = prof["Rasterize"][0] / prof["Rasterize"][1]
    srt = np.sort(work)[::-1]
    print("%s: blended %.2f M, tile work mean %.0f p50 %d p90 %d p99 %d max %d | K17 %.1f us = %.3f us per pair of the heaviest tile, %.3f ns per pair overall | K16 %.1f us = %.3f us per max-tile pair"
          % (wl, work.sum() / 1e6, work.mean(), np.percentile(work, 50), np.percentile(work, 90), np.percentile(work, 99), work.max(), k17 * 1e3, k17 * 1e3 / work.max(), k17 * 1e6 / work.sum(), k16 * 1e3, k16 * 1e3 / work.max()))
    print("   top tiles:", srt[:12].tolist())
    ctx.close()
