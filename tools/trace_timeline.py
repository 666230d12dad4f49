"""Summarises a rocprofv3 --kernel-trace CSV as a timeline: per kernel name count / total / mean, and for the last call of the
traced program the start / end of every kernel relative to the first (to see which stages overlap).
usage: python tools/trace_timeline.py <dir or kernel_trace.csv> [last_n_fill_kernels]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", "")))
rows.sort()
agg = {}
for s, e, n, _ in rows:
    a = agg.setdefault(n, [0, 0])
    a[0] += 1; a[1] += e - s
print("%-62s %8s %12s %10s" % ("kernel", "calls", "total ms", "mean us"))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %8d %12.3f %10.2f" % (n, c, t / 1e6, t / c / 1e3))
fills = [i for i, r in enumerate(rows) if "fill" in r[2] and "single" not in r[2]]
if fills:
    first = fills[-nlast] if len(fills) >= nlast else fills[0]
    t0 = rows[first][0]
    print("\ntimeline of the last %d fill kernels and everything after the first of them (ms from its start):" % min(nlast, len(fills)))
    for s, e, n, st in rows[first:]:
        print("  %9.3f -> %9.3f  (%8.3f)  %-50s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n, st))
