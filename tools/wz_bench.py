"""Developer tool: time the fused half-resolution wavelet branch (ey_wavelet_z) at the shapes of EdgeLine-n, replayed from a hipGraph.
usage: wz_bench.py [reps] [c,hw ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import edge_yolo_amd  # noqa: E402,F401
from edge_yolo_amd.nn import _ops as ops  # noqa: E402
from edge_yolo_amd.nn.modules import block as B  # noqa: E402

SHAPES = [(16, 160), (32, 80), (64, 40), (128, 20)]
if len(sys.argv) > 2:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for c, hw in SHAPES:
    m = B._WaveletEnhancer(c).cuda().half().eval()
    for mod in m.modules():
        if hasattr(mod, "fuse_bn") and hasattr(mod, "bn"):
            mod.fuse_bn()
    xs = [torch.randn(32, hw, hw, c, device="cuda", dtype=torch.float16).permute(0, 3, 1, 2) for _ in range(6)]
    for x in xs[:2]:
        ops.wavelet_z(m, x, m._subband_sets, m._fuse_z)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        zs = [ops.wavelet_z(m, xs[i % 6], m._subband_sets, m._fuse_z) for i in range(reps)]
    g.replay()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(5):
        g.replay()
    en.record()
    torch.cuda.synchronize()
    us = st.elapsed_time(en) / (5 * reps) * 1e3
    print(f"wavelet_z C{c} {hw}x{hw}: {us:8.1f} us", flush=True)
