"""Developer tool: per-stage time split of the block programs (workgroup 0's wall_clock64 stamps, ey_block_run_timed).
    python tools/block_stage_times.py [--model yolo11n-test.yaml] [--batch 32] [--imgsz 640]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="yolo11n-test.yaml")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--imgsz", type=int, default=640)
a = ap.parse_args()
model, _ = bench.build_model(a.model, torch.float16, torch.device("cuda:0"))
model.block_fusion = model.model[-1].block_fusion = True
x = torch.rand(a.batch, 3, a.imgsz, a.imgsz, device="cuda").half()
for _ in range(3):
    model(x)
torch.cuda.synchronize()
progs = [p for c in model._block_caches.values() for p in c.progs] + [p for c in model.model[-1]._blk.values() for p in c.progs]
for p in progs:
    p.timing = torch.zeros(p.n + 1, dtype=torch.int64, device="cuda")
acc = {id(p): torch.zeros(p.n, dtype=torch.float64) for p in progs}
reps = 5
for _ in range(reps):
    model(x)
    torch.cuda.synchronize()
    for p in progs:
        t = p.timing.cpu().double()
        acc[id(p)] += (t[1:] - t[:-1]) / 100.0  # 100 MHz -> us
for p in progs:
    us = acc[id(p)] / reps
    print(f"== {p.tag}: {p.n} stages, {float(us.sum()):.1f} us (workgroup 0)")
    for i, d in enumerate(p.desc):
        print(f"  {i:2d} {d:60s} {float(us[i]):8.1f} us")
