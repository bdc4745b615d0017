"""Developer tool: time single conv shapes through the module API, replayed from a hipGraph (clean per-kernel times).
usage: conv_bench.py [reps] [c1,c2,k,s,hw ...]      (dispatch tunables: EY_* environment variables)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import edge_yolo_amd  # noqa: E402,F401
from edge_yolo_amd.nn import modules as M  # noqa: E402

SHAPES = [(64, 64, 3, 1, 80), (64, 64, 1, 1, 80), (16, 32, 3, 2, 320), (32, 32, 1, 1, 160), (128, 128, 3, 2, 80), (128, 128, 1, 1, 20), (384, 256, 1, 1, 20),
          (64, 32, 1, 1, 40), (128, 64, 1, 1, 20), (64, 64, 1, 1, 40), (128, 128, 1, 1, 40), (256, 256, 1, 1, 20), (512, 256, 1, 1, 20), (256, 128, 1, 1, 20),
          (128, 384, 1, 1, 20), (80, 80, 1, 1, 40), (80, 80, 1, 1, 80), (256, 64, 3, 1, 20), (128, 64, 3, 1, 40), (128, 256, 3, 2, 40)]
if len(sys.argv) > 2:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NBUF = 6  # rotate inputs/outputs so that consecutive launches do not hit the same lines
for c1, c2, k, s, hw in SHAPES:
    m = M.Conv(c1, c2, k, s).cuda().half().eval()
    xs = [torch.randn(32, hw, hw, c1, device="cuda", dtype=torch.float16).permute(0, 3, 1, 2) for _ in range(NBUF)]
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
    y = ys[0]
    nbytes = (xs[0].numel() + y.numel()) * 2
    fl = 2.0 * y.numel() * c1 * k * k
    print(f"{c1}->{c2} k{k}s{s} {hw}x{hw}: {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s  {fl / us / 1e6:7.1f} TF/s", flush=True)
