#!/bin/bash
# Developer tool (GPU box): second SQ counter set (pipe activity) of any tools/*.py bench.  usage: tools/pmc_any2.sh <tag> <script.py> args...
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES"
timeout -k 10 200 rocprofv3 --pmc $C -d $R/gpurun_out/pmc2_$tag -o p --output-format csv -- python $R/$script "$@" > $R/gpurun_out/pmc2_$tag.log 2>&1
python3 - "$R/gpurun_out/pmc2_$tag/p_counter_collection.csv" <<'PY'
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    if 'kernel' not in k:
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m['SQ_WAVE_CYCLES']
    print(k)
    print('   waves %d wave_cycles %.3g | active VALU %.0f%% LDS %.0f%% VMEM %.0f%% of wave cycles | mfma_busy %.3g  lds_conflict %.3g  sq_busy %.3g' % (
        m['SQ_WAVES'], wc, 100 * m['SQ_ACTIVE_INST_VALU'] / wc, 100 * m['SQ_ACTIVE_INST_LDS'] / wc, 100 * m['SQ_ACTIVE_INST_VMEM'] / wc,
        m['SQ_VALU_MFMA_BUSY_CYCLES'], m['SQ_LDS_BANK_CONFLICT'], m['SQ_BUSY_CYCLES']))
PY
