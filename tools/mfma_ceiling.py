"""Developer tool: what the K >= 576 3x3 convs of the network (layers 3, 5, 7, 17, 20 and the head box towers) reach
as a function of the batch -- the same kernels ey_conv2d dispatches at batch 32, timed from a replayed hipGraph at B = 32, 64, 128,
256, 512: TFLOP/s, fraction of the dense f16 MFMA peak (2500 TFLOP/s) and of HBM (algorithmic bytes).  Shows whether the gap to
the north-star "50 % MFMA utilisation" at batch 32 is problem size or kernel design.
    python tools/mfma_ceiling.py [--csv out.csv]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import edge_yolo_amd  # noqa: E402,F401
from edge_yolo_amd import _lib as L  # noqa: E402
from edge_yolo_amd.nn import modules as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--csv", default="")
ap.add_argument("--batches", default="32,64,128,256,512")
a = ap.parse_args()
# (name, c1, c2, k, s, input hw at 640x640)
SHAPES = [("layer 3", 64, 64, 3, 2, 160), ("layer 5", 128, 128, 3, 2, 80), ("layer 7", 128, 256, 3, 2, 40), ("layer 17", 64, 64, 3, 2, 80),
          ("layer 20", 128, 128, 3, 2, 40), ("head box 80x80", 64, 64, 3, 1, 80), ("head box 40x40 (first)", 128, 64, 3, 1, 40),
          ("head box 20x20 (first)", 256, 64, 3, 1, 20)]
rows = []
print(f"{'conv':26s} {'B':>4s} {'us':>9s} {'TFLOP/s':>9s} {'% MFMA peak':>11s} {'GB/s (alg)':>11s} kernel")
for name, c1, c2, k, s, hw in SHAPES:
    m = M.Conv(c1, c2, k, s).cuda().half().eval()
    for B in [int(v) for v in a.batches.split(",")]:
        nbuf = 4
        xs = [torch.randn(B, hw, hw, c1, device="cuda", dtype=torch.float16).permute(0, 3, 1, 2) for _ in range(nbuf)]
        y = m(xs[0])
        var = L.lib().ey_conv_last_variant() or L.lib().ey_conv_variant(0, c2, c1, k, s, 1, y.shape[0] * y.shape[2] * y.shape[3], 1)
        torch.cuda.synchronize()
        reps = 8
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(reps):
                m(xs[i % nbuf])
        g.replay()
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(5):
            g.replay()
        en.record()
        torch.cuda.synchronize()
        us = st.elapsed_time(en) / (5 * reps) * 1e3
        fl = 2.0 * y.numel() * c1 * k * k
        nbytes = (xs[0].numel() + y.numel() + c1 * c2 * k * k) * 2
        tf = fl / us / 1e6
        rows.append((name, f"{c1}->{c2} k{k}s{s} {hw}x{hw}", B, round(us, 2), round(tf, 1), round(tf / 2500 * 100, 2), round(nbytes / us / 1e3), var))
        print(f"{name:26s} {B:4d} {us:9.1f} {tf:9.1f} {tf / 25:10.1f}% {nbytes / us / 1e3:11.0f} variant {var}", flush=True)
        del xs, g
if a.csv:
    with open(a.csv, "w") as f:
        f.write("conv,shape,batch,avg_us,tflops,pct_of_dense_f16_mfma_peak_2500,alg_GBps,kernel_variant_code\n")
        for r in rows:
            f.write(",".join(str(v) for v in r) + "\n")
