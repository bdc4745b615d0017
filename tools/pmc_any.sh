#!/bin/bash
# Developer tool (GPU box): SQ counters of any tools/*.py bench.  usage: tools/pmc_any.sh <tag> <script.py> args...
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
timeout -k 10 200 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_$tag -o p --output-format csv -- python $R/$script "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$tag/p_counter_collection.csv | grep -A1 "dsconv\|dwconv\|conv_\|conv3\|linattn\|head_decode\|stem"
