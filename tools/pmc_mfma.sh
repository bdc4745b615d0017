#!/bin/bash
# GPU box: MFMA-pipe busy share per kernel of the bench step (rocprofv3 --pmc, own pass).  usage: tools/pmc_mfma.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcm_$T -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-api --no-pipeline > $R/gpurun_out/pmcm_$T.log 2>&1
python3 - $(find $R/gpurun_out/pmcm_$T -name "*counter_collection.csv" | head -1) $R/gpurun_out/${T}_pmc_mfma.csv <<'PY'
import collections, csv, sys
sys.path.insert(0, sys.argv[0] and '.')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if 'GRBM_GUI_ACTIVE' not in m or m['GRBM_GUI_ACTIVE'] <= 0:
        continue
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the SIMDs' MFMA pipes; GRBM_GUI_ACTIVE: active cycles summed over the 8 XCDs
    # 1024 SIMDs: busy share = mfma_busy / (1024 * gui_active / 8)
    share = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024.0 * m['GRBM_GUI_ACTIVE'] / 8.0)
    rows.append((m['GRBM_GUI_ACTIVE'] * len(v['GRBM_GUI_ACTIVE']), k, len(v['GRBM_GUI_ACTIVE']), m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), m['GRBM_GUI_ACTIVE'], share))
rows.sort(reverse=True)
with open(sys.argv[2], 'w') as f:
    f.write('kernel,launches,avg_SQ_VALU_MFMA_BUSY_CYCLES,avg_GRBM_GUI_ACTIVE_sum_over_8_XCDs,mfma_pipe_busy_share\n')
    for _, k, n, mf, gui, sh in rows:
        f.write('"%s",%d,%.0f,%.0f,%.4f\n' % (k[:90], n, mf, gui, sh))
for _, k, n, mf, gui, sh in rows[:16]:
    print('%-70s n=%3d mfma busy %.1f%%' % (k[:70], n, 100 * sh))
PY
find $R/gpurun_out/pmcm_$T -name "*counter_collection.csv" -delete
