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

# --json <file> --key <name>: the PMC figures bench.py reads for roofline.traffic / valu / LDS, stamped with a hash of the kernel
# sources they were measured on (bench.py marks the figures stale when the sources have changed since)
if "--json" in sys.argv:
    import hashlib
    import json
    path = sys.argv[sys.argv.index("--json") + 1]
    key = sys.argv[sys.argv.index("--key") + 1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("aln_kernels.hip", "aln_fast.h", "aln_device.h"):
        h.update(open(os.path.join(root, "aligner_amd", "csrc", f), "rb").read())
    fill = [k for k in agg if "aln_fill_fast_kernel" in k]
    if fill:
        k = fill[0]
        per = lambda c: agg[k][c] / max(cnt[k][c], 1)
        try:
            prev = json.load(open(path))
        except Exception:
            prev = {}
        prev[key] = {
            "kernel": k, "kernel_src_sha16": h.hexdigest()[:16],
            # FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is not doubled: the reads are 1-8 B per
            # lane chunk loads, not wide coalesced streams
            "hbm_bytes_per_launch": (per("FETCH_SIZE") + per("WRITE_SIZE")) * 1024.0,
            "fetch_size_kb_per_dispatch": per("FETCH_SIZE"), "write_size_kb_per_dispatch": per("WRITE_SIZE"),
            "valu_wave_insts_per_launch": per("SQ_INSTS_VALU"),
            "lds_bank_conflict_ratio": (per("SQ_LDS_BANK_CONFLICT") / per("SQ_LDS_IDX_ACTIVE")) if per("SQ_LDS_IDX_ACTIVE") else None,
            "source": "tools/profile.sh: separate rocprofv3 --pmc passes per fill dispatch, summary next to this file",
        }
        json.dump(prev, open(path, "w"), indent=1)
