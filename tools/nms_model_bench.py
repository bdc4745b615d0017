"""Developer tool: time ey_nms on the model's own (dense-regime) predictions for several max_det."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from edge_yolo_amd.utils import ops
model, _ = bench.build_model("yolo11n-test.yaml", torch.float16, torch.device("cuda:0"))
x = torch.rand(32, 3, 640, 640, device="cuda").half()
pred, _ = model(x)
torch.cuda.synchronize()
print("candidates/img", float((pred[:, 4:].amax(1) > 0.25).sum(1).float().mean()))
for md in (1, 50, 100, 200, 300):
    for _ in range(3):
        ops.nms_device(pred, 0.25, 0.7, max_det=md)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        out = ops.nms_device(pred, 0.25, 0.7, max_det=md)
    e.record(); torch.cuda.synchronize()
    idx = out[2][:, :md]
    print(f"max_det={md}: {s.elapsed_time(e)/20*1e3:7.1f} us  kept/img {out[1].float().mean():.0f}")
# how deep into the sorted candidate list is the 300th kept box?
b, c, idx = ops.nms_device(pred, 0.25, 0.7, max_det=300)
score = pred[:, 4:].amax(1)
for i in range(3):
    last = b[i, int(c[i]) - 1, 4]
    print("img", i, "rank of last kept score:", int((score[i] > last).sum()))
