"""Developer tool: throughput of the predict step when the per-GPU batch is split into L concurrent lanes (each lane = its
own forward/NMS hipGraphs on its own HIP streams).    python tools/lanes_bench.py [--batch 32] [--lanes 1,2,4]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from edge_yolo_amd.engine.predictor import PipelinedRunner  # noqa: E402
from edge_yolo_amd.utils import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="yolo11n-test.yaml")
ap.add_argument("--batch", default="32")
ap.add_argument("--imgsz", type=int, default=640)
ap.add_argument("--lanes", default="1,2,4")
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda:0")
model, _ = bench.build_model(a.model, torch.float16, dev)
for B in [int(v) for v in a.batch.split(",")]:
    x = torch.rand(B, 3, a.imgsz, a.imgsz, device=dev).half()
    for L in [int(v) for v in a.lanes.split(",")]:
        if B % L:
            continue
        pipes = []
        for l in range(L):
            xs = x[l * (B // L):(l + 1) * (B // L)].contiguous()
            p = PipelinedRunner(lambda im: model(im)[0], lambda pred: ops.nms_device(pred, 0.25, 0.7, max_det=300)[:2], xs)
            for j in range(2):
                p.static_input(j).copy_(xs)
            pipes.append(p)

        def step():
            for p in pipes:
                p.submit()

        for _ in range(8):
            step()
        for p in pipes:
            p.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        for p in pipes:
            p.wait()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"batch {B} lanes {L}: {dt * 1e3:.3f} ms/step  {B / dt:.0f} img/s", flush=True)
        del pipes
