#!/bin/bash
# Developer tool: sweep dispatch tunables of the conv kernels in ONE gpurun call.  usage: tools/sweep.sh "VAR=val ..." ...
for cfg in "$@"; do
  r=$(env $cfg timeout -k 10 150 python bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$cfg -> $r ms"
done
