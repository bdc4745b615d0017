#!/bin/bash
# GPU box: 3x3 conv shapes at batch 32 with the stream kernel on / off.  usage: tools/c3s_bench.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python - <<'PY' > gpurun_out/$1_c3s.txt 2>&1
import sys, torch
sys.path.insert(0, '.')
import edge_yolo_amd
from edge_yolo_amd import _lib as L
from edge_yolo_amd.nn import modules as M
SH = [("L3", 64, 64, 2, 160), ("L5", 128, 128, 2, 80), ("L7", 128, 256, 2, 40), ("L17", 64, 64, 2, 80), ("L20", 128, 128, 2, 40), ("head80", 64, 64, 1, 80),
      ("head40a", 128, 64, 1, 40), ("head40b", 64, 64, 1, 40), ("head20a", 256, 64, 1, 20), ("head20b", 64, 64, 1, 20), ("dsc3k 64->32", 64, 32, 1, 20)]
import os
if os.environ.get("C3P_ONLY"):
    SH = [s_ for s_ in SH if s_[1] == 64 and s_[3] == 1 and s_[2] == 64]
def t(m, xs, reps=8):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): m(xs[i % len(xs)])
    g.replay(); torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(5): g.replay()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / (5 * reps) * 1e3
for name, c1, c2, s, hw in SH:
    m = M.Conv(c1, c2, 3, s).cuda().half().eval(); m.fuse_bn()
    xs = [torch.randn(32, hw, hw, c1, device="cuda", dtype=torch.float16).permute(0, 3, 1, 2) for _ in range(4)]
    line = f"{name:14s} {c1}->{c2} s{s} {hw}x{hw}:"
    for cfg in (((0, 0, 0, 0), (0, 0, 2, 1), (0, 0, 2, 0)) if os.environ.get("C3P_ONLY") else ((0, 0, 0, 0), (1, 0, 1, 1), (2, 43, 0, 0), (2, 23, 0, 0), (0, 0, 2, 1))):
        L.check(L.lib().ey_tune_set(b"c3s", cfg[0]), "t"); L.check(L.lib().ey_tune_set(b"c3s_cfg", cfg[1]), "t"); L.check(L.lib().ey_tune_set(b"c3p", cfg[2]), "t")
        L.check(L.lib().ey_tune_set(b"c3p_fast", cfg[3]), "t")
        y = m(xs[0]); var = L.lib().ey_conv_last_variant()
        us = t(m, xs)
        fl = 2.0 * y.numel() * c1 * 9
        line += f" | {('c3p fast%d' % cfg[3] if cfg[3] else 'c3p plain') if cfg[2] == 2 else 'old' if not cfg[0] else 'default' if cfg[0] == 1 else 'cfg %2d' % cfg[1]} v{var}: {us:6.1f} us {fl / us / 1e6 / 25:5.1f}%"
    print(line, flush=True)
PY
cat gpurun_out/$1_c3s.txt
