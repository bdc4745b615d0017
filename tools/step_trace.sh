#!/bin/bash
# GPU box: kernel sequence of the last bench step (rocprofv3 kernel trace).  usage: tools/step_trace.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_s -o p --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/trace_s.log 2>&1
python3 - $R/gpurun_out/trace_s <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nms_select' in r['Kernel_Name']]
a, b = idx[-3], idx[-1]
seg = rows[a + 1: b + 1]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    n = r['Kernel_Name'][:46]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q{r.get('Queue_Id', '?'):>3} {n}")
PY
find $R/gpurun_out/trace_s -name "*kernel_trace.csv" -delete
