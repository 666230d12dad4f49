#!/bin/bash
# PMC pass on the generic f64 fill kernel: usage tools/pmc_f64.sh <tag> [pairs] [scale del ext]   (ALN_F64_OLD=1 in the environment: the old loop)
set -o pipefail
TAG=$1; PAIRS=${2:-20000}; SCHEME="${3:-0.37} ${4:-11.3} ${5:-2.1}"
OUT=gpurun_out/r03b/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
  NAME=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/$NAME -- python3 tools/bench_f64_staged.py $PAIRS $SCHEME > $OUT/log_$NAME.txt 2>&1 || { echo "pmc $NAME failed"; tail -3 $OUT/log_$NAME.txt; }
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:48]; c=row["Counter_Name"]; agg[k][c]+=float(row["Counter_Value"]); cnt[k][c]+=1
for k in agg:
    if "fill" not in k: continue
    print(k)
    for c in sorted(agg[k]): print("   %-28s per_dispatch=%.6g (n=%d)"%(c, agg[k][c]/cnt[k][c], cnt[k][c]))
PY
rm -rf $OUT/SQ_*
