"""Developer tool: time single DSConv / DWConv shapes through the module API, replayed from a hipGraph.
usage: ds_bench.py [reps] [c,k,hw ...]   (k < 0: DWConv(c, c, -k) instead of DSConv(c, c, k))"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import edge_yolo_amd  # noqa: E402,F401
from edge_yolo_amd.nn import modules as M  # noqa: E402

SHAPES = [(16, 3, 160), (16, 7, 160), (32, 3, 80), (32, 7, 80), (64, 3, 40), (64, 7, 40), (32, 3, 40), (32, 5, 40), (64, 3, 20), (64, 5, 20),
          (64, -3, 80), (80, -3, 80), (128, -3, 40), (256, -3, 20)]
if len(sys.argv) > 2:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NBUF = 6
for c, k, hw in SHAPES:
    m = (M.DSConv(c, c, k) if k > 0 else M.DWConv(c, c, -k)).cuda().half().eval()
    xs = [torch.randn(32, hw, hw, c, device="cuda", dtype=torch.float16).permute(0, 3, 1, 2) for _ in range(NBUF)]
    for x in xs[:2]:
        y = m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ys = [m(xs[i % NBUF]) for i in range(reps)]
    g.replay()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(5):
        g.replay()
    en.record()
    torch.cuda.synchronize()
    us = st.elapsed_time(en) / (5 * reps) * 1e3
    nbytes = (xs[0].numel() + ys[0].numel()) * 2
    print(f"{'DSConv' if k > 0 else 'DWConv'} C{c} k{abs(k)} {hw}x{hw}: {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s", flush=True)
