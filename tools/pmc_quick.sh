#!/bin/bash
# quick PMC pass on the fill kernel: usage tools/pmc_quick.sh <tag> [pairs]   (counters in separate rocprofv3 runs, no trace domains)
set -o pipefail
TAG=$1
PAIRS=${2:-100000}
OUT=gpurun_out/r03/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--pairs $PAIRS --steps 2 --warmup 1 --no-cpu-baseline --no-single-pair --no-small-configs --no-end-to-end"
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  NAME=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/$NAME -- python3 bench.py $ARGS > $OUT/log_$NAME.txt 2>&1 || { echo "pmc $NAME failed"; tail -3 $OUT/log_$NAME.txt; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:40]; c=row["Counter_Name"]; agg[k][c]+=float(row["Counter_Value"]); cnt[k][c]+=1
for k in agg:
    if "fill" not in k: continue
    print(k)
    for c in sorted(agg[k]): print("   %-28s per_dispatch=%.6g (n=%d)"%(c, agg[k][c]/cnt[k][c], cnt[k][c]))
PY
rm -rf $OUT/SQ_*
