#!/bin/bash
# usage (GPU box): tools/nms_prof.sh <tag> [cases...]  -> gpurun_out/<tag>_nms_kernels.txt (per regime: kernel, calls, avg us)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cases=${@:-dense empty sparse001 synsparse syndense}
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/${tag}_nms_kernels.txt
for c in $cases; do
  rm -rf /tmp/nmsprof_$c
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/nmsprof_$c -o p --output-format csv -- python $R/tools/nms_prof.py $c > /tmp/nmsprof_$c.log 2>&1 || { tail -5 /tmp/nmsprof_$c.log; exit 1; }
  f=$(find /tmp/nmsprof_$c -name "*kernel_stats.csv" | head -1)
  python - "$f" "$c" >> $R/gpurun_out/${tag}_nms_kernels.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Name"] for k in ("nf_", "nms_"))]
print(f"== {sys.argv[2]}")
for r in rows:
    print(f"  {r['Name'].split('(')[0]:28s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}  max {float(r['MaxNs']) / 1e3:8.1f}")
PY
done
cat $R/gpurun_out/${tag}_nms_kernels.txt
