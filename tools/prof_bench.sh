#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel-trace summary of the default bench step.  usage: tools/prof_bench.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$1 -o p --output-format csv -- python $R/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_$1.log 2>&1
f=$(find $R/gpurun_out/prof_$1 -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/prof_$1_kernel_stats.csv
find $R/gpurun_out/prof_$1 -name "*kernel_trace.csv" -delete
head -45 $f | cut -c1-150
