"""Developer tool: host-side enqueue time per bench step (is the loop CPU-bound?).  usage: host_rate.py [--gather]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from edge_yolo_amd.engine.predictor import PipelinedRunner  # noqa: E402
from edge_yolo_amd.utils import ops  # noqa: E402

dev = torch.device("cuda:0")
model, _ = bench.build_model("yolo11n-test.yaml", torch.float16, dev)
x = torch.rand(32, 3, 640, 640, device=dev).half()
pipe = PipelinedRunner(lambda im: model(im)[0], lambda pred: ops.nms_device(pred, 0.25, 0.7, max_det=300)[:2], x)
for j in range(2):
    pipe.static_input(j).copy_(x)
for _ in range(10):
    pipe.submit()
pipe.wait()
torch.cuda.synchronize()
for n in (1, 50):
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.submit()
    t1 = time.perf_counter()
    pipe.wait()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, total {1e3 * (t2 - t0) / n:.3f} ms/step", flush=True)
g = pipe.sets[0]["g1"]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"forward graph alone: host {1e3 * (t1 - t0) / 20:.3f} ms/replay, total {1e3 * (t2 - t0) / 20:.3f} ms/replay")
