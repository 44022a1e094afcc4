#!/usr/bin/env python
"""Where the step's wall time goes that is NOT kernel time: per pair of consecutive kernels of a rocprofv3 --kernel-trace CSV, the
idle gap between the end of one and the start of the next, averaged over the steps of the trace's steady part.
    python scripts/gap_trace.py <kernel_trace.csv> [first_step] [steps]
A step starts at project_forward_kernel; only steps [first_step, first_step + steps) are read (default: the last 20 whole steps)."""
import csv
import sys
from collections import defaultdict


def short(name):
    return name.replace("void bh::", "").replace("bh::", "").split("(")[0].split("<")[0]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2] == "project_forward_kernel"]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(starts) - steps - 1)
    gaps, durs, wall = defaultdict(list), defaultdict(list), []
    for s in range(first, min(first + steps, len(starts) - 1)):
        a, b = starts[s], starts[s + 1]
        wall.append(rows[b][0] - rows[a][0])
        for i in range(a, b):
            durs[rows[i][2]].append(rows[i][1] - rows[i][0])
            nxt = rows[i + 1]
            gaps[(rows[i][2], nxt[2])].append(nxt[0] - rows[i][1])
    n = max(1, len(wall))
    print("steps %d  wall/step %.1f us  kernels/step %.1f us  gaps/step %.1f us" %
          (n, sum(wall) / n / 1e3, sum(sum(v) for v in durs.values()) / n / 1e3, sum(sum(v) for v in gaps.values()) / n / 1e3))
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        print("  %-28s -> %-28s  n/step %.2f  mean %.2f us  per step %.2f us" % (a[:28], b[:28], len(v) / n, sum(v) / len(v) / 1e3, sum(v) / n / 1e3))


if __name__ == "__main__":
    main()
