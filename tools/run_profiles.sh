#!/bin/bash
# Run on the GPU box (through gpurun): bench lines + rocprofv3 kernel stats + HBM-traffic PMC passes for the three measured
# workloads.  Outputs land in gpurun_out/<tag>/; `python tools/make_profiles.py <tag>` (CPU, afterwards) turns them into the
# committed summaries under profiles/.   usage: bash tools/run_profiles.sh r02_mid [workloads...]
set -o pipefail
TAG=${1:-r03}; shift
WL=${@:-"train31 train64 train256 infer1024_bf16"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  case $w in
    train31)  PARGS="--workload train31 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline";;
    train256) PARGS="--workload train256 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline";;
    train64)  PARGS="--workload train64 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline";;
    infer1024_bf16) PARGS="--workload infer1024_bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline";;
    infer1024_f32) PARGS="--workload infer1024_f32 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline";;
  esac
  echo "== $w: rocprofv3 --kernel-trace --stats"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${w}_stats -- python3 $R/bench.py $PARGS > $OUT/${w}_stats.log 2>&1 || { echo "stats $w failed"; tail -n 5 $OUT/${w}_stats.log; exit 1; }
  echo "== $w: --pmc FETCH_SIZE"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${w}_fetch -- python3 $R/bench.py $PARGS > $OUT/${w}_fetch.log 2>&1 || { echo "fetch $w failed"; exit 1; }
  echo "== $w: --pmc WRITE_SIZE"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${w}_write -- python3 $R/bench.py $PARGS > $OUT/${w}_write.log 2>&1 || { echo "write $w failed"; exit 1; }
  # keep only what the summariser needs (gpurun_out/ merges back at most 64 MiB)
  find $OUT/${w}_stats -type f ! -name "*kernel_stats.csv" -delete
  find $OUT/${w}_fetch $OUT/${w}_write -type f ! -name "*counter_collection.csv" -delete
done
# the bench lines come LAST: bench.py takes roofline.traffic from profiles/<tag>_<workload>_hbm_traffic.json, i.e. from the PMC
# passes above (make_profiles.py here writes them into the box's copy of profiles/; run it again after the call to commit them)
python3 $R/tools/make_profiles.py $TAG > $OUT/make_profiles.log 2>&1 || { echo "make_profiles failed"; tail -n 5 $OUT/make_profiles.log; exit 1; }
for w in $WL; do
  case $w in
    train31)  ARGS="--workload train31";;
    train256) ARGS="--workload train256 --steps 10 --warmup 3";;
    train64)  ARGS="--workload train64";;
    infer1024_bf16) ARGS="--workload infer1024_bf16 --steps 50 --warmup 10";;
    infer1024_f32) ARGS="--workload infer1024_f32 --steps 20 --warmup 5";;
  esac
  echo "== $w: bench"; python3 $R/bench.py $ARGS > $OUT/${w}_bench.json 2> $OUT/${w}_bench.err || { echo "bench $w failed"; tail -n 5 $OUT/${w}_bench.err; exit 1; }
  tail -c 300 $OUT/${w}_bench.json; echo
done
echo "== RCCL group at world 1 (torch.distributed.run, one rank)"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/train31_rccl_world1.json 2> $OUT/train31_rccl_world1.err || { echo "rccl run failed"; tail -n 8 $OUT/train31_rccl_world1.err; exit 1; }
tail -c 250 $OUT/train31_rccl_world1.json; echo
ls $OUT
