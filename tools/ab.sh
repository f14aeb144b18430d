#!/bin/bash
# dev tool (GPU box): time bench.py against several prebuilt library variants back-to-back; usage: tools/ab.sh name1 name2 ...
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = cur ]; then lib=""; else lib=$PWD/tools/_bin/libssie_hip_$v.so; fi
  r=$(SSIE_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") || exit 1
  echo "$v $r"
done; done
