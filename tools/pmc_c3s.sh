#!/bin/bash
# GPU box: FETCH_SIZE + SQ counters of single conv shapes (through tools/conv_bench.py).  usage: tools/pmc_c3s.sh <tag> shape...
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_$tag /tmp/ps_$tag
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$tag -o p --output-format csv -- python $R/tools/conv_bench.py 3 "$@" > /tmp/pf_$tag.log 2>&1 || { tail -3 /tmp/pf_$tag.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS -d /tmp/ps_$tag -o p --output-format csv -- python $R/tools/conv_bench.py 3 "$@" > /tmp/ps_$tag.log 2>&1 || { tail -3 /tmp/ps_$tag.log; exit 1; }
rm -rf /tmp/pl_$tag
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d /tmp/pl_$tag -o p --output-format csv -- python $R/tools/conv_bench.py 3 "$@" > /tmp/pl_$tag.log 2>&1 || { tail -3 /tmp/pl_$tag.log; }
python3 - $(find /tmp/pl_$tag -name "*counter_collection.csv" | head -1) > $R/gpurun_out/${tag}_pmc_lds.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "conv" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(f"{k[0]:60s} grid {k[1]:>8s}: " + "  ".join(f"{c}={m[c]:.3g}" for c in sorted(m)))
PY
cat $R/gpurun_out/${tag}_pmc_lds.txt
python3 - $(find /tmp/pf_$tag -name "*counter_collection.csv" | head -1) $(find /tmp/ps_$tag -name "*counter_collection.csv" | head -1) > $R/gpurun_out/${tag}_pmc.txt <<'PY'
import csv, sys, collections
f = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "conv" in r["Kernel_Name"]:
        f[(r["Kernel_Name"][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in f.items():
    print(f"{k[0]:60s} grid {k[1]:>8s}: FETCH x2 = {2 * sum(v) / len(v) / 1e3:8.1f} MB  ({len(v)} dispatches)")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[2])):
    if "conv" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    w, wc = m["SQ_WAVES"], m["SQ_WAVE_CYCLES"]
    print(f"{k[0]:60s} grid {k[1]:>8s}: waves {w:.0f} cycles/wave {4 * wc / w:.0f} valu/wave {m['SQ_INSTS_VALU'] / w:.0f} | wait_any {100 * m['SQ_WAIT_ANY'] / wc:.0f}% wait_inst {100 * m['SQ_WAIT_INST_ANY'] / wc:.0f}% (lds {100 * m.get('SQ_WAIT_INST_LDS', 0) / wc:.0f}%) active {100 * m['SQ_ACTIVE_INST_ANY'] / wc:.0f}%")
PY
cat $R/gpurun_out/${tag}_pmc.txt
