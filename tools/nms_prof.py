"""Developer tool (under rocprofv3 --kernel-trace --stats): one NMS regime per process so that the per-kernel averages are per regime.
    python tools/nms_prof.py dense|empty|sparse001|synsparse|syndense"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import synthdata as synth
from edge_yolo_amd.utils import ops
case = sys.argv[1]
dev = torch.device("cuda:0")
x = torch.rand(32, 3, 640, 640, device=dev).half()
if case == "dense":
    m, _ = bench.build_model("yolo11n-test.yaml", torch.float16, dev)
    pred, conf = m(x, head_nms={"conf": 0.25, "classes": None})[0], 0.25
elif case in ("empty", "sparse001"):
    m, _ = bench.build_model("yolo11n-test.yaml", torch.float16, dev, regime="sparse")
    conf = 0.25 if case == "empty" else 0.001
    pred = m(x, head_nms={"conf": conf, "classes": None})[0]
else:
    pred, conf = synth.synth_pred(32, 80, 8400, seed=2 if case == "synsparse" else 3, dense=case == "syndense").to(dev), 0.25
torch.cuda.synchronize()
for _ in range(20):
    ops.nms_device(pred, conf, 0.7, max_det=300)
torch.cuda.synchronize()
