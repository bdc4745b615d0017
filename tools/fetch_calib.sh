#!/bin/bash
# GPU box: FETCH_SIZE calibration table (tools/micro/fetch_calib.hip).  usage: tools/fetch_calib.sh  -> gpurun_out/fetch_calib.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/micro/fetch_calib.hip -o /tmp/fetch_calib || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/fc -o p --output-format csv -- /tmp/fetch_calib > $R/gpurun_out/fetch_calib.log 2>&1
python3 - $(find /tmp/fc -name "*counter_collection.csv" | head -1) > $R/gpurun_out/fetch_calib.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
GiB = 1 << 30
want = {"uint4": GiB, "uint2": GiB, "unsigned int": GiB, "unsigned short": GiB, "seg288": GiB // 1280 * 288}
print("FETCH_SIZE calibration (gfx950, 1 GiB buffer read once; KB -> bytes x 1024):")
for k, v in agg.items():
    kb = sum(v) / len(v)
    w = next((b for t, b in want.items() if t in k), None)
    if w:
        print(f"  {k:60s} FETCH_SIZE {kb * 1024 / 1e6:9.1f} MB for {w / 1e6:8.1f} MB read -> bytes / FETCH_SIZE = {w / (kb * 1024):.3f}")
PY
cat $R/gpurun_out/fetch_calib.txt
