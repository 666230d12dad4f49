"""Condenses rocprofv3 output (kernel-trace stats + PMC csv) into a short text summary per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("%-70s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:70], row.get("Calls"),
                  row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
print("== PMC (sum over dispatches, per kernel) ==")
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")[:60]
            c = row.get("Counter_Name")
            agg[k][c] += float(row.get("Counter_Value", 0))
            cnt[k][c] += 1
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        print("   %-24s sum=%.6g dispatches=%d per_dispatch=%.6g" % (c, agg[k][c], cnt[k][c], agg[k][c] / max(cnt[k][c], 1)))
