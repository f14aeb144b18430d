#!/bin/bash
# dev tool (GPU box): rocprofv3 kernel stats of a short bench run, rows matching a filter.  usage: [WORKLOAD=train31] [ENVV="A=1 B=2"] bash tools/kstats.sh <regex>
R=${GRAFT_REPO_ROOT:-$(pwd)}; W=${WORKLOAD:-train31}; OUT=$R/gpurun_out/kstats; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in $ENVV; do export $kv; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
f=$(find $OUT/s -name "*kernel_stats.csv" | head -1)
python3 - "$f" "${1:-.}" <<'P'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print(f'{int(r["Calls"]):5d} {float(r["AverageNs"])/1e3:9.1f} us  {float(r["TotalDurationNs"])/1e6:8.3f} ms  {r["Name"][:110]}')
P
find $OUT -mindepth 2 -type f -delete
