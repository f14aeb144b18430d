#!/bin/bash
# dev tool (GPU box): time bench.py against several prebuilt library variants back-to-back
# usage: tools/ab.sh [-w workload] name1 name2 ...     name = cur | <variant in tools/_bin> ; optional ":ENV=VAL[,ENV=VAL]" suffix
wl=train31; if [ "$1" = -w ]; then wl=$2; shift 2; fi
for rep in 1 2; do
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  if [ "$v" = cur ]; then lib=""; else lib=$PWD/tools/_bin/libssie_hip_$v.so; fi
  r=$(env SSIE_DEBUG=1 $envs SSIE_HIP_LIB=$lib timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") || exit 1
  echo "$spec $r"
done; done
