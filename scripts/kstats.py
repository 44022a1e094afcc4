"""Print the per-kernel averages of a rocprofv3 --kernel-trace --stats CSV (names shortened)."""
import csv
import sys
for f in sys.argv[1:]:
    print("==", f)
    for r in list(csv.DictReader(open(f)))[:int(1e9)]:
        if not r["Name"].startswith(("void bh::", "bh::", "__amd")):
            continue
        name = r["Name"].replace("void bh::", "").replace("bh::", "").split("(")[0]
        print("  %-44s calls %6s avg_us %8.2f total_ms %8.2f" % (name[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
