#!/bin/bash
# GPU box: the measurements a round commits under profiles/.  usage: tools/round_profile.sh <tag>
#   1. bench.py (default flags) -> gpurun_out/bench_<tag>.json      2. rocprofv3 --kernel-trace --stats of the same step
#   3./4. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no trace domains) -> per-kernel traffic
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$1
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python $R/bench.py --roofline-csv $O/${T}_roofline_table.csv > $O/bench_$T.json 2> $O/bench_$T.err && tail -c 400 $O/bench_$T.json && echo &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-api > $O/prof_$T.log 2>&1 &&
cp $(find $O/prof_$T -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats.csv && find $O/prof_$T -name "*kernel_trace.csv" -delete &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmcf_$T -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-api --no-pipeline > $O/pmcf_$T.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmcw_$T -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-api --no-pipeline > $O/pmcw_$T.log 2>&1 &&
python3 $R/tools/traffic_from_pmc.py $(find $O/pmcf_$T -name "*counter_collection.csv" | head -1) $(find $O/pmcw_$T -name "*counter_collection.csv" | head -1) $O/${T}_pmc_traffic.json &&
find $O/pmcf_$T $O/pmcw_$T -name "*counter_collection.csv" -delete
# the other single-GPU configurations of BASELINE.json (C2: YOLO11n; C5 shape: 1280x1280 batch 8) and the GC10-DET class count
timeout -k 10 300 python $R/bench.py --model yolo11n.yaml --no-cpu-baseline --no-api > $O/bench_${T}_c2_yolo11n.json 2>> $O/bench_$T.err
timeout -k 10 300 python $R/bench.py --imgsz 1280 --batch 8 --no-cpu-baseline --no-api > $O/bench_${T}_c5_1280_b8.json 2>> $O/bench_$T.err
timeout -k 10 300 python $R/bench.py --nc 10 --no-cpu-baseline --no-api > $O/bench_${T}_nc10.json 2>> $O/bench_$T.err
timeout -k 10 300 python $R/bench.py --no-pipeline --no-cpu-baseline --no-api --no-roofline > $O/bench_${T}_single_graph.json 2>> $O/bench_$T.err
timeout -k 10 300 python $R/bench.py --regime sparse --no-cpu-baseline --no-api > $O/bench_${T}_sparse.json 2>> $O/bench_$T.err
timeout -k 10 300 python $R/bench.py --regime sparse --cls-bias-shift 5 --no-cpu-baseline --no-api --no-roofline > $O/bench_${T}_sparse_shift5.json 2>> $O/bench_$T.err
timeout -k 10 300 python $R/bench.py --force-gather --no-cpu-baseline --no-api --no-roofline > $O/bench_${T}_force_gather.json 2>> $O/bench_$T.err
for f in c2_yolo11n c5_1280_b8 nc10 single_graph sparse sparse_shift5 force_gather; do python3 -c "import json,sys; d=json.load(open('$O/bench_${T}_$f.json')); print('$f', d['value'], d['ms_per_step'])"; done
