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
