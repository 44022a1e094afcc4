"""Copies the judged summaries from gpurun_out/prof_<tag>/ into profiles/ (tracked):
the rocprofv3 --kernel-trace --stats table, per-kernel HBM traffic from the two separate
PMC passes (FETCH_SIZE / WRITE_SIZE, KB per launch; on gfx950 FETCH_SIZE under-counts wide
coalesced reads by 2x — MI355X_MICROARCH.md §HBM — both raw and corrected are listed),
and the bench JSON line produced under the profiler."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
for f in ("bench_trace.json",):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, tag + "_" + f))


def agg(path):
    a = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].strip()
        a[k][0] += float(r["Counter_Value"])
        a[k][1] += 1
    return {k: (v / c, c) for k, (v, c) in a.items()}


fetch = agg(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"))
write = agg(os.path.join(src, "pmc_write", "write_counter_collection.csv"))
with open(os.path.join(dst, tag + "_hbm_traffic.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch_raw", "FETCH_KB_x2_gfx950_correction", "WRITE_SIZE_KB_per_launch", "HBM_MB_per_launch_corrected"])
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[0] * 2 + write.get(k, (0, 0))[0])):
        fr, c = fetch.get(k, (0.0, 0))
        wr, c2 = write.get(k, (0.0, 0))
        w.writerow([k, max(c, c2), round(fr, 1), round(fr * 2, 1), round(wr, 1), round((fr * 2 + wr) / 1024.0, 2)])
print(open(os.path.join(dst, tag + "_hbm_traffic.csv")).read()[:3000])


# SQ instruction counters of the blend kernels -> VALU wave-instructions per blended intersection (bench.py's roofline_valu)
sq = os.path.join(src, "pmc_sq", "sq_counter_collection.csv")
bj = os.path.join(src, "bench_sq.json")
if os.path.exists(sq) and os.path.exists(bj):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(sq)):
        k = r["Kernel_Name"].split("(")[0].strip()
        e = acc[k][r["Counter_Name"]]
        e[0] += float(r["Counter_Value"]); e[1] += 1
    try:
        blended = json.load(open(bj))["roofline"]["intersections_blended"]
    except Exception:
        blended = 0
    names = {"rasterize_backward_kernel": "rasterize_backward_kernel", "rasterize_kernel": "rasterize_kernel"}
    with open(os.path.join(dst, tag + "_sq_counters.csv"), "w") as f:
        w = csv.writer(f)
        cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
        w.writerow(["kernel", "full_name", "launches"] + [c + "_per_launch" for c in cols] + ["intersections_blended", "valu_per_blended_isect"])
        for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", [0, 1])[0]):
            short = next((v for n, v in names.items() if ("::" + n + "<") in k or k.endswith("::" + n)), k.split("::")[-1].split("<")[0])
            per = {c: (d[c][0] / d[c][1] if c in d and d[c][1] else 0.0) for c in cols}
            is_blend = short in names.values()
            w.writerow([short, k, max((d[c][1] for c in d), default=0)] + [round(per[c], 1) for c in cols] +
                       [blended if is_blend else "", round(per["SQ_INSTS_VALU"] / blended, 2) if is_blend and blended else ""])
    print(open(os.path.join(dst, tag + "_sq_counters.csv")).read()[:1500])
