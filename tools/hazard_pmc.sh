#!/bin/bash
# VALU instructions of the C5 fill (40 000 pairs) with and without the row-1 hazard (del != ext / del == ext)
set -o pipefail
OUT=gpurun_out/r02/hazard_pmc; mkdir -p $OUT; export TMPDIR=/tmp
for DE in "11 2" "11 11"; do
  T=$(echo $DE | tr ' ' '_')
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $OUT/$T -- python3 tools/hazard_cost.py $DE > $OUT/log_$T.txt 2>&1 || { echo failed; tail -3 $OUT/log_$T.txt; }
  grep "del/ext" $OUT/log_$T.txt
done
python3 - <<PY
import csv,glob,collections
for t in ("11_2","11_11"):
    agg=collections.defaultdict(float); cnt=collections.defaultdict(int)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%t, recursive=True):
        for row in csv.DictReader(open(f)):
            if "aln_fill_fast" not in row["Kernel_Name"]: continue
            agg[row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[row["Counter_Name"]]+=1
    print(t, {c: "%.5g"%(agg[c]/cnt[c]) for c in sorted(agg)})
PY
rm -rf $OUT/11_2 $OUT/11_11
