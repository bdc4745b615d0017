#!/bin/bash
# GPU box: bench one tunable over several values.  usage: tools/tune_sweep.sh <name> <v1> <v2> ...   (single graph and pipelined ms/step)
n=$1; shift
for v in "$@"; do
  a=$(timeout -k 10 120 python bench.py --steps 40 --tune $n=$v --no-api --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(timeout -k 10 120 python bench.py --steps 40 --tune $n=$v --no-api --no-cpu-baseline --no-roofline --no-pipeline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$n=$v pipelined $a ms  single-graph $b ms"
done
