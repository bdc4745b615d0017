"""Developer tool: per-launch table (kernel, shape, us, achieved GB/s and TFLOP/s) of one eager predict step.
    python tools/layer_profile.py [--model yolo11n-test.yaml] [--batch 32] [--imgsz 640] [--dtype f16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from edge_yolo_amd import profiling  # noqa: E402
from edge_yolo_amd.utils import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="yolo11n-test.yaml")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--imgsz", type=int, default=640)
ap.add_argument("--dtype", default="f16")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dt = torch.float16 if a.dtype == "f16" else torch.float32
model, _ = bench.build_model(a.model, dt, torch.device("cuda:0"))
x = torch.rand(a.batch, 3, a.imgsz, a.imgsz, device="cuda").to(dt)


def step():
    pred, _ = model(x)
    return ops.nms_device(pred, 0.25, 0.7)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profiling.trace() as t:
    for _ in range(a.reps):
        step()
torch.cuda.synchronize()
n = len(t.records) // a.reps
tot = 0.0
print(f"{'#':>3} {'kernel':34s} {'shape':34s} {'us':>8s} {'GB/s':>8s} {'TF/s':>7s}")
for i in range(n):
    k, b, f, _, _, note, _ = t.records[i]
    us = sum(t.records[i + r * n][3].elapsed_time(t.records[i + r * n][4]) for r in range(a.reps)) / a.reps * 1e3
    tot += us
    print(f"{i:3d} {k:34s} {note:34s} {us:8.1f} {b / us / 1e3:8.0f} {f / us / 1e6:7.1f}")
print(f"total {tot:.0f} us over {n} launches")
