"""Developer tool: layers 0 + 1 at the benchmark shape (batch 32, 640x640, f16): one kernel (ey_stem_pair, 8x32 / 8x16 tiles) vs the two launches, graph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edge_yolo_amd  # noqa: F401
from edge_yolo_amd import _lib as L
from edge_yolo_amd.nn import _ops, modules as M

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m0, m1 = M.Conv(3, 16, 3, 2).cuda().half().eval(), M.Conv(16, 32, 3, 2).cuda().half().eval()
x = torch.rand(B, 3, 640, 640, device="cuda").half()
y = L.empty_nhwc(B, 32, 160, 160, torch.float16, "cuda")


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * reps) * 1e3


print(f"two launches: {timed(lambda: m1(m0(x), out=y)):6.1f} us")
for t in (5, 2, 3, 4):
    L.check(L.lib().ey_tune_set(b"stem_pair", t), "t")
    us = timed(lambda: _ops.stem_pair(m0, m1, x, out=y))
    name = {5: "8x32", 2: "8x16", 3: "4x16", 4: "4x32"}[t]
    print(f"stem_pair tile {name}: {us:6.1f} us  ({(x.numel() + y.numel()) * 2 / us / 1e3:5.0f} GB/s algorithmic)")
