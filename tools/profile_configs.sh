#!/bin/bash
# rocprofv3 on the small configurations: kernel stats of C2 / C3 (tools/bench_configs.py) and of the 10k x 10k single pair
# (tools/bench_single.py), PMC counters of the C3 fill.   usage: tools/profile_configs.sh <tag>  -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-cfg}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t1 -- python3 tools/bench_configs.py > $OUT/bench_configs.log 2>&1
cp $(find $OUT/t1 -name "*kernel_stats.csv" | head -1) $OUT/c2_c3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t2 -- python3 tools/bench_single.py 10000 > $OUT/bench_single.log 2>&1
cp $(find $OUT/t2 -name "*kernel_stats.csv" | head -1) $OUT/c4_kernel_stats.csv
for SET in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  NAME=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$NAME -- python3 tools/bench_configs.py > $OUT/pmc_$NAME.log 2>&1
done
python3 tools/summarize_prof.py $OUT > $OUT/summary_all.txt 2>&1
grep -A14 -E "aln_fill_duo_kernel|aln_fill_fast_kernel<0" $OUT/summary_all.txt > $OUT/c3_fill_pmc.txt
rm -rf $OUT/t1 $OUT/t2 $OUT/pmc_*/
tail -2 $OUT/bench_configs.log; tail -1 $OUT/bench_single.log; cat $OUT/c3_fill_pmc.txt
