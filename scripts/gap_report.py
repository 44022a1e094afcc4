#!/usr/bin/env python
"""Where the GPU idles inside a train step: from a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W --windows 1 ...`,
the gaps between consecutive kernel dispatches of the timed steps, summed by (previous kernel -> next kernel).
    rocprofv3 --kernel-trace --output-format csv -d /tmp/t -o trace -- python bench.py --steps 20 --warmup 5 --windows 1 --no-extra --no-pmc --no-cpu-baseline --no-stages
    python scripts/gap_report.py /tmp/t 5 20"""
import csv
import glob
import os
import sys


def short(name):
    return name.replace("void ", "").replace("bh::", "").split("(")[0].split("<")[0]


def main():
    d, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    upd = [int(r["End_Timestamp"]) for r in rows if "train_update_kernel" in r["Kernel_Name"]]
    lo, hi = upd[warmup - 1], upd[warmup + steps - 1]
    sel = [r for r in rows if lo < int(r["End_Timestamp"]) <= hi]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
    gaps = {}
    prev_end, prev_name = lo, "train_update_kernel"
    for r in sel:
        g = int(r["Start_Timestamp"]) - prev_end
        k = (prev_name, short(r["Kernel_Name"]))
        a = gaps.setdefault(k, [0, 0])
        a[0] += max(g, 0)
        a[1] += 1
        prev_end, prev_name = max(prev_end, int(r["End_Timestamp"])), short(r["Kernel_Name"])
    span = hi - lo
    print("timed span %.1f us/step, kernels busy %.1f us/step, idle %.1f us/step, %d dispatches/step" % (span / steps / 1e3, busy / steps / 1e3, (span - busy) / steps / 1e3, len(sel) / steps))
    for (a, b), (ns, cnt) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print("  %-34s -> %-34s %7.2f us/step  (%.2f us x %.1f per step)" % (a, b, ns / steps / 1e3, ns / max(cnt, 1) / 1e3, cnt / steps))


if __name__ == "__main__":
    main()
