"""Developer tool: time the predict-mode NMS on the model's own predictions (dense regime = random-init weights, sparse regime =
bias_init biases at conf 0.001, synthetic sparse pred) for the fast path at several K and the general kernel alone (K = 0).
    python tools/nms_model_bench.py [--graph]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import synthdata as synth
from edge_yolo_amd import _lib
from edge_yolo_amd.utils import ops


def timed(fn, n=30):
    for _ in range(5):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3, out


dev = torch.device("cuda:0")
x = torch.rand(32, 3, 640, 640, device=dev).half()
cases = {}
model, _ = bench.build_model("yolo11n-test.yaml", torch.float16, dev)
cases["model dense (conf .25)"] = (model(x)[0], 0.25)
cand, _ = model(x, head_nms={"conf": 0.25, "classes": None})
cases["model dense, fused candidates"] = (cand, 0.25)
ms, _ = bench.build_model("yolo11n-test.yaml", torch.float16, dev, regime="sparse")
cases["model sparse regime (conf .001)"] = (ms(x)[0], 0.001)
cases["model sparse regime (conf .25: nothing passes)"] = (ms(x)[0], 0.25)
cases["synthetic sparse (SURVEY 8d)"] = (synth.synth_pred(32, 80, 8400, seed=2).to(dev), 0.25)
cases["synthetic dense"] = (synth.synth_pred(32, 80, 8400, seed=3, dense=True).to(dev), 0.25)
torch.cuda.synchronize()
for name, (pred, conf) in cases.items():
    line = []
    for K in (0, 1024, 1536, 2048):
        _lib.check(_lib.lib().ey_tune_set(b"nms_fast_k", K), "tune")
        us, out = timed(lambda: ops.nms_device(pred, conf, 0.7, max_det=300))
        line.append(f"K={K}: {us:7.1f} us")
    print(f"{name:48s} kept/img {float(out[1].float().mean()):6.1f} | " + " | ".join(line), flush=True)
_lib.check(_lib.lib().ey_tune_set(b"nms_fast_k", 2048), "tune")
p = cases["model dense (conf .25)"][0]
b, c, idx = ops.nms_device(p, 0.25, 0.7, max_det=300)
score = p[:, 4:].amax(1)
ranks = [int((score[i] > b[i, int(c[i]) - 1, 4]).sum()) for i in range(32)]
print("rank of the last kept score per image:", ranks)
