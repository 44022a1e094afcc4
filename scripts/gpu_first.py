"""First-contact GPU script: smoke + stagewise parity on the 10k scene + sort/scan checks."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import brush_amd as ba
from brush_amd import synth
from oracle import bo
import __graft_entry__ as g

t=time.time(); g.smoke(); print("smoke time", time.time()-t)
dev = torch.device("cuda:0")
# sort
rng = np.random.default_rng(0)
for n, bits in ((15, 32), (1000, 32), (4096, 32), (4097, 13), (100000, 32), (1_000_003, 32), (3_000_000, 13)):
    keys = rng.integers(0, 2**32 if bits == 32 else 2**bits, size=n, dtype=np.uint64).astype(np.uint32)
    vals = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    ok, ov = ba.radix_argsort(torch.from_numpy(keys.view(np.int32)).to(dev), torch.from_numpy(vals.view(np.int32)).to(dev), bits)
    rk, rv = bo.radix_argsort(keys, vals, bits)
    print("sort", n, bits, np.array_equal(ok.cpu().numpy().view(np.uint32), rk), np.array_equal(ov.cpu().numpy().view(np.uint32), rv))
for n in (4, 1024, 4096, 4097, 41083, 3_000_001):
    x = rng.integers(0, 1000, size=n).astype(np.uint32)
    o = ba.prefix_sum(torch.from_numpy(x.view(np.int32)).to(dev)).cpu().numpy().view(np.uint32)
    print("scan", n, np.array_equal(o, bo.prefix_sum(x)))
# 10k stagewise
sc, w, h = synth.config_scene("10k_256", 0)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
img, aux = ba.render_splats(spl, cam, (w, h), (0, 0, 0), ba.RasterPass.Backward)
R = bo.Render().forward(bo.camera(**cp), sc["transforms"], sc["sh"], sc["raw_opac"])
print("nv", aux.num_visible, R.num_visible, "ni", aux.num_intersections, R.num_intersections)
def eq(name, a, b):
    a = a.cpu().numpy(); 
    if a.dtype == np.int32: a = a.view(np.uint32)
    b = b.reshape(a.shape)
    print(name, "exact" if np.array_equal(a, b) else "DIFF max %g count %d" % (np.abs(a.astype(np.float64)-b).max(), (a!=b).sum()))
eq("intersect_counts", aux.intersect_counts, R.get("intersect_counts"))
eq("max_radius", aux.max_radius, R.get("max_radius"))
eq("gfc", aux.global_from_compact_gid, R.get("global_from_compact_gid"))
eq("depths", aux.depths_sorted, R.get("depths_sorted"))
eq("cum", aux.cum_tiles_hit, R.get("cum_tiles_hit"))
eq("projected", aux.projected_splats, R.get("projected"))
eq("tile_ids", aux.tile_id_from_isect, R.get("tile_id_from_isect"))
eq("isect_gids", aux.compact_gid_from_isect, R.get("compact_gid_from_isect"))
eq("tile_offsets", aux.tile_offsets, R.get("tile_offsets"))
eq("visible", aux.visible, R.get("visible"))
eq("img", img, R.get("out_img"))
# 1M timing
sc, w, h = synth.config_scene("1m_1080p", 0)
cp = synth.default_camera_params(w, h)
cam = ba.Camera(position=cp["pos"], rotation=cp["rot_xyzw"], fov_x=cp["fov_x"], fov_y=cp["fov_y"], center_uv=cp["center_uv"])
spl = ba.Splats(sc["transforms"], sc["sh"], sc["raw_opac"], device=dev)
ctx = ba.get_context(dev)
vout = torch.full((h, w, 4), 1.0/(h*w*4), device=dev)
for it in range(3):
    ctx.profile(it == 2)
    torch.cuda.synchronize(); t = time.time()
    res = ba.render_splats_bwd(spl, cam, (w, h), (0, 0, 0), vout)
    torch.cuda.synchronize(); print("1M fwd+bwd wall (incl copies)", time.time() - t, "nv", res["aux"].num_visible, "ni", res["aux"].num_intersections)
for k, v in ctx.profile_fetch().items(): print("  %-28s %8.3f ms x%d" % (k, v[0], v[1]))
