"""Developer tool: SPPF pooling (256 ch -> c_ = 128, 20x20, batch 32, f16) for every channel-group width; run under rocprofv3 --pmc FETCH_SIZE for traffic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edge_yolo_amd
from edge_yolo_amd import _lib as L
from edge_yolo_amd.nn import _ops
x = torch.randn(32, 20, 20, 128, device="cuda", dtype=torch.float16).permute(0, 3, 1, 2)
buf = L.empty_nhwc(32, 512, 20, 20, torch.float16, "cuda")
ys = [buf[:, 128 * (i + 1):128 * (i + 2)] for i in range(3)]
for cv in (1, 2, 4, 8):
    L.check(L.lib().ey_tune_set(b"sppf_min_wg", 1), "t"); L.check(L.lib().ey_tune_set(b"sppf_cv", cv), "t")
    for _ in range(3): _ops.sppf_pool(x, *ys)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): _ops.sppf_pool(x, *ys)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    print(f"cv={cv} ({32 * 128 // (8 * cv)} workgroups): {a.elapsed_time(b) / 50 * 1e3:6.1f} us", flush=True)
