#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats, then separate PMC passes (never combined with trace domains).
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-run}; shift
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline --no-single-pair --no-small-configs --no-end-to-end}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/bench_trace.log; exit 1; }
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  NAME=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$NAME -- python3 bench.py $ARGS > $OUT/bench_pmc_$NAME.log 2>&1 || { echo "pmc $NAME failed"; tail -3 $OUT/bench_pmc_$NAME.log; }
done
# (the JSON: what bench.py reads for roofline.traffic / valu / LDS -- copy it to profiles/hbm_traffic.json with the summary)
python3 tools/summarize_prof.py $OUT --json $OUT/hbm_traffic.json --key ${TRAFFIC_KEY:-c5_100000_n1} > $OUT/summary.txt 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace $OUT/pmc_*
cat $OUT/summary.txt
