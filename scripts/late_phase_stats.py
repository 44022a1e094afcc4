#!/usr/bin/env python
"""Per-kernel statistics of the LATE PHASE of a training run from a rocprofv3 --kernel-trace CSV of `bench.py --loop-only <mode>`: the
kernels of the last `steps` train steps before the end of the trace (a step starts at project_forward_kernel; the steps of a probe —
eight per segment, each followed by a host sync — are train steps like the others).  Writes a small CSV (kernel, calls, per-step
count, average / min / max us, us per step, share).   python scripts/late_phase_stats.py <kernel_trace.csv> <out.csv> [steps=400]"""
import csv
import sys
from collections import defaultdict


def short(name):
    return name.replace("void bh::", "").replace("bh::", "").split("(")[0]


def main():
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        n = r["Kernel_Name"]
        if "bh::" not in n:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(n)))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("project_forward_kernel")]
    # the training run ends with the last refine-free stretch; renders of the held-out views follow every segment: keep to whole steps
    # that contain an update kernel (a train step), take the last `steps` of them
    train = [(starts[k], starts[k + 1]) for k in range(len(starts) - 1) if any(rows[i][2].startswith("train_update_kernel") for i in range(starts[k], starts[k + 1]))]
    train = train[-steps:]
    acc = defaultdict(list)
    for a, b in train:
        for i in range(a, b):
            acc[rows[i][2]].append((rows[i][1] - rows[i][0]) / 1e3)
    n = max(1, len(train))
    total = sum(sum(v) for v in acc.values()) / n
    with open(sys.argv[2], "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "calls_per_step", "avg_us", "min_us", "max_us", "us_per_step", "share_of_kernel_time"])
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), round(len(v) / n, 3), round(sum(v) / len(v), 2), round(min(v), 2), round(max(v), 2), round(sum(v) / n, 2), round(sum(v) / n / total, 4)])
        w.writerow(["TOTAL (steps %d, kernels per step %.1f us)" % (n, total), "", "", "", "", "", round(total, 2), 1.0])
    print(open(sys.argv[2]).read())


if __name__ == "__main__":
    main()
