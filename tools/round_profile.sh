#!/bin/bash
# GPU box: the measurements a round commits under profiles/.  usage: tools/round_profile.sh <tag>
#   1. bench.py (default flags) -> gpurun_out/bench_<tag>.json      2. rocprofv3 --kernel-trace --stats of the same step
#   3./4. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no trace domains) -> per-kernel traffic
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$1
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python $R/bench.py > $O/bench_$T.json 2> $O/bench_$T.err && tail -c 400 $O/bench_$T.json && echo &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/prof_$T.log 2>&1 &&
cp $(find $O/prof_$T -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv && find $O/prof_$T -name "*kernel_trace.csv" -delete &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmcf_$T -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-pipeline > $O/pmcf_$T.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmcw_$T -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-pipeline > $O/pmcw_$T.log 2>&1 &&
python3 $R/tools/traffic_from_pmc.py $(find $O/pmcf_$T -name "*counter_collection.csv" | head -1) $(find $O/pmcw_$T -name "*counter_collection.csv" | head -1) $O/${T}_pmc_traffic.json &&
find $O/pmcf_$T $O/pmcw_$T -name "*counter_collection.csv" -delete
