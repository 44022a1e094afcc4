#!/usr/bin/env python
"""Fills the two generated blocks of DESIGN.md (between `<!-- figures:begin/end -->` and `<!-- headline:begin/end -->`) from the
round's committed bench lines, so that the document carries ONE current figure per kernel: profiles/<tag>_bench_n1.json (the driver's
command) and, when present, its siblings.      python scripts/design_figures.py r5"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
check_only = "--check" in sys.argv[1:]   # exit 1 if DESIGN.md does not carry exactly the figures of the committed profiles
tag = args_[0] if args_ else "r6"


def load(name):
    p = os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name))
    if not os.path.exists(p):
        return None
    txt = open(p).read().strip()
    lines = [ln for ln in txt.splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


d = load("bench_n1")
assert d, "profiles/%s_bench_n1.json missing" % tag
names = [("ProjectSplats", "K1 cull + count (+ fills, tile order)"), ("DepthSort", "K2+K3 depth order + scan (4 launches)"),
         ("MapGaussiansToIntersect", "K5 map (+ row gather)"), ("TileSort", "K6+K15 tile sort + offsets (5 launches)"), ("Rasterize", "K16 blend"),
         ("ImageLoss", "K19 loss pass A"), ("ImageLossBackward", "K20 loss pass B"), ("RasterizeBackwards", "K17 blend backward"),
         ("ProjectBackwards", "K18 project backward"), ("OptimizerStep", "Adam + stats + noise")]
rows = ["| stage | µs per step | share of kernel time | bytes (algorithmic) | achieved | note |", "|---|---|---|---|---|---|"]
tot = sum(v["ms"] for v in d["stages"].values())
for key, label in names:
    e = d["stages"].get(key)
    if not e:
        continue
    note = ""
    if "hbm_frac" in e:
        note = "%.2f of 8 TB/s" % e["hbm_frac"]
    elif "cache_resident" in e:
        note = "Infinity-Cache resident input"
    rows.append("| %s | %.1f | %.1f %% | %s | %s | %s |" % (label, e["ms"] * 1e3, 100.0 * e["ms"] / tot, ("%.0f MB" % e["MB"]) if "MB" in e else "—",
                                                          ("%.0f GB/s" % e["GBps"]) if "GBps" in e else "—", note))
rows.append("| **sum of kernels** | **%.1f** | | | | wall per step %.1f µs (median of %d replica windows: %s) |" % (
    tot * 1e3, d["ms_per_step"] * 1e3, len(d["windows_ms_per_step"]), " / ".join("%.4f" % x for x in d["windows_ms_per_step"])))
rf, rv = d["roofline"], d["roofline_valu"]
k17, k16 = rv.get("rasterize_backward_kernel", {}), rv.get("rasterize_kernel", {})
extra = ["",
         "K17 (HIP events on its own dispatch packets, timed steps of all windows): **%.1f µs** for %.3f M blended pairs = %.3f ns per pair; %.1f MB touched "
         "→ %.0f GB/s = **%.3f of the HBM roofline** (PMC traffic %.1f MB per launch: ratio %.2f); %.1f VALU instructions per pair → **%.2f of the nominal VALU "
         "issue peak**.  K16: %.3f ns per pair, %.1f instructions per pair, %.2f of the VALU peak." % (
             rf["avg_launch_ms"] * 1e3, rf["intersections_blended"] / 1e6, k17.get("ns_per_blended_intersection", 0.0), rf["bytes_per_launch"] / 1e6, rf["achieved"], rf["frac"],
             (rf.get("traffic") or 0) / 1e6, (rf.get("traffic") or 0) / max(rf["bytes_per_launch"], 1), k17.get("valu_insts_per_blended_intersection", 0.0), k17.get("frac", 0.0),
             k16.get("ns_per_blended_intersection", 0.0), k16.get("valu_insts_per_blended_intersection", 0.0), k16.get("frac", 0.0))]
fig = "\n".join(rows + extra)

h = []
h.append("**Headline** (`profiles/%s_bench_n1.json`, the driver's command, one MI355X): **%.4f ms per step = %.0f views/s** (replica windows %s; "
         "without the two events on K17: %s ms); fwd %.3f ms, fwd + bwd %.3f ms (kernel timestamps); near share %.2f–%.2f (mean %.2f), %d second attempts in the timed steps."
         % (tag, d["ms_per_step"], d["value"], " / ".join("%.4f" % x for x in d["windows_ms_per_step"]), d.get("ms_per_step_without_kernel_events"),
            d["fwd_ms"], d["fwd_bwd_ms"], d["config"]["near_share"]["min"], d["config"]["near_share"]["max"], d["config"]["near_share"]["mean"], d["config"]["far_slices_queued"]))
cb = d.get("cpu_baseline")
if cb:
    h.append("CPU oracle on %d host cores: %.3f views/s (%s)." % (cb["cores"], cb["value"], cb["kind"]))
tl = d.get("train_loop")
if tl:
    a, b, c = tl["cuts_view_ids"], tl["cuts_no_ids"], tl["exact_lists"]
    h.append("**Training loop at the named size on a converging scene** (`train_loop`: %s): per-tile cuts with automatic complete lists **%.4f ms per step** over the run "
             "(%d second attempts, near share mean %.2f, %d frames with complete lists), keyed by the camera %.4f, complete lists throughout %.4f; held-out PSNR %.1f → %.1f dB; "
             "splats %d → %d over %d refines (%.1f ms each)."
             % (tl["workload"], a["ms_per_step"], a["second_attempts"], a["near_share"]["mean"], a["frames_with_complete_lists"], b["ms_per_step"],
                c["ms_per_step"], a["psnr_held_out"][0], a["psnr_held_out"][1], a["splats_over_time"][0][1], a["splats_over_time"][-1][1], a["refine_calls"], a["refine_ms_each"]))
    seg_rows = ["| steps | ms/step (cuts) | ms/step (complete lists) | near share | 2nd attempts | pairs / frame | blended / frame | K16 ms (ns/pair) | K17 ms (ns/pair) | PSNR dB |",
                "|---|---|---|---|---|---|---|---|---|---|"]
    for sa, sc_ in zip(a["segments"], c["segments"]):
        seg_rows.append("| %d–%d | %.4f | %.4f | %.2f | %d | %.2f M | %.2f M | %.3f (%.3f) | %.3f (%.3f) | %.1f |" % (
            sa["steps"][0], sa["steps"][1], sa["ms_per_step"], sc_["ms_per_step"], sa.get("near_share_mean", 1.0), sa.get("second_attempts", 0), sa["pairs_per_frame"] / 1e6,
            sa["blended_pairs_per_frame"] / 1e6, sa["k16_ms"], sa["k16_ns_per_blended_pair"], sa["k17_ms"], sa["k17_ns_per_blended_pair"], sa["psnr_held_out"]))
    h.append("\n".join(seg_rows))
    last = c["segments"][-1]
    if last.get("stages_us"):
        st = last["stages_us"]
        tot_st = sum(st.values())
        h.append("**The late phase is the representative regime** (steps %d–%d, complete lists, HIP events around every stage over the segment's probe steps; %.0f µs of stages per step): "
                 % (last["steps"][0], last["steps"][1], tot_st) + "; ".join("%s %.0f µs (%.0f %%)" % (k, v, 100.0 * v / tot_st) for k, v in sorted(st.items(), key=lambda kv: -kv[1])) + ".")
lt = d.get("after_growth_stop")
if lt:
    h.append("After `growth_stop_iter` (the blend backward without the refine weight, same protocol as the headline): %.4f ms per step = %.0f views/s; K17 %.1f µs (HIP events around the stage)."
             % (lt["ms_per_step"], lt["views_per_s"], lt["k17_ms_hip_events_around_the_stage"] * 1e3))
ov = d.get("other_view_counts") or {}
if ov:
    h.append("Other view counts (same scene): " + "; ".join("%d view%s %.3f ms" % (v["views"], "" if v["views"] == 1 else "s", v["ms_per_step"]) for v in ov.values()) + ".")
fo = d.get("forward_only")
if fo:
    h.append("Forward only (configs[1], packed rgba8, per call incl. its count readback): %.3f ms with cut lists, %.3f ms with complete lists." % (fo["ms_sliced_lists"], fo["ms_exact_lists"]))
ns, s3 = d.get("non_saturating"), d.get("sh3")
if s3:
    h.append("SH degree 3 (in the same line): %.3f ms per step." % s3["ms_per_step"])
if ns:
    h.append("Non-saturating scene: %.3f ms per step, K17 at %.2f / K16 at %.2f of the VALU peak." % (
        ns["ms_per_step"], ns["roofline_valu"].get("rasterize_backward_kernel", {}).get("frac", 0.0), ns["roofline_valu"].get("rasterize_kernel", {}).get("frac", 0.0)))
oc = d.get("object_centric")
if oc:
    line = "Object-centric frame (half of the tiles empty, the heaviest blends 15x the mean; not a BASELINE config): %.3f ms per step, K16 %.0f µs, K17 %.0f µs" % (
        oc["ms_per_step"], oc["k16_ms"] * 1e3, oc["k17_ms"] * 1e3)
    if oc.get("one_wave_per_tile_forward"):
        line += "; with one wave per tile in the forward (`k16_split=0`) %.3f ms, K16 %.0f µs" % (oc["one_wave_per_tile_forward"]["ms_per_step"], oc["one_wave_per_tile_forward"]["k16_ms"] * 1e3)
    if oc.get("whole_tile_backward"):
        line += "; with whole tiles in the backward (`bwd_jobs=0`) %.3f ms, K17 %.0f µs" % (oc["whole_tile_backward"]["ms_per_step"], oc["whole_tile_backward"]["k17_ms"] * 1e3)
    h.append(line + ".")
sib = []
for name, label in (("bench_exact_lists_n1", "complete lists"), ("bench_no_view_ids_n1", "no view ids (keyed by camera)"), ("bench_sh3_n1", "SH degree 3"), ("bench_6m_4k_sh3_n1", "6 M / 4K / SH 3 on ONE GPU"),
                    ("bench_feed_loader_n1", "loader feed (PCIe-inclusive)"), ("bench_no_noise_n1", "without the stochastic terms"),
                    ("bench_pg1_native_dense", "1-rank RCCL, library communicator, dense"), ("bench_pg1_native_sparse", "… mask-keyed"),
                    ("bench_pg1_torch_dense", "1-rank RCCL, torch hook, dense"), ("bench_pg1_torch_sparse", "… mask-keyed")):
    x = load(name)
    if x:
        sib.append("%s %.4f ms" % (label, x["ms_per_step"]))
if sib:
    h.append("Same box, same call (`profiles/%s_bench_*.json`, each `--steps 20 --warmup 5`): " % tag + "; ".join(sib) + ".")
head = "\n\n".join(h)

p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
before = s
s = re.sub(r"<!-- figures:begin -->.*?<!-- figures:end -->", lambda m: "<!-- figures:begin -->\n" + fig + "\n<!-- figures:end -->", s, flags=re.S)
s = re.sub(r"<!-- headline:begin -->.*?<!-- headline:end -->", lambda m: "<!-- headline:begin -->\n" + head + "\n<!-- headline:end -->", s, flags=re.S)
if check_only:
    if s != before:
        print("DESIGN.md does not match profiles/%s_*.json: run `python scripts/design_figures.py %s`" % (tag, tag))
        sys.exit(1)
    print("DESIGN.md carries the figures of profiles/%s_*.json" % tag)
    sys.exit(0)
open(p, "w").write(s)
print(fig)
print()
print(head)
