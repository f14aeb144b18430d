#!/bin/bash
# dev tool (GPU box): kernel-trace stats + SQ / traffic counters of the kernels of a train step whose name matches a filter.
# usage: [WORKLOAD=train256] bash tools/pmc_step.sh [filter-regex]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${WORKLOAD:-train31}
OUT=$R/gpurun_out/pmc_step
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1 || { echo "stats failed"; tail -5 $OUT/stats.log; }
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 $R/tools/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 --filter "${1:-wino}" > $OUT/summary.txt
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
find $OUT -mindepth 2 -type f -delete
cat $OUT/summary.txt
