"""Developer tool: time ey_nms alone on synthetic predictions (dense/sparse, varying max_det)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import edge_yolo_amd  # noqa: E402,F401
from edge_yolo_amd.utils import ops  # noqa: E402
import synthdata as synth  # noqa: E402


def t(pred, n=20, **kw):
    for _ in range(3):
        ops.nms_device(pred, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        out = ops.nms_device(pred, **kw)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3, out[1].float().mean().item()


for B, A in ((32, 8400), (8, 33600)):
    for dense in (True, False):
        p = synth.synth_pred(B, 80, A, seed=1, dense=dense).cuda()
        for md in (1, 300):
            us, cnt = t(p, conf_thres=0.25, iou_thres=0.7, max_det=md)
            ncand = int((p[:, 4:].amax(1) > 0.25).sum()) // B
            print(f"B={B} A={A} dense={dense} max_det={md}: {us:8.1f} us  (candidates/img ~{ncand}, kept/img {cnt:.0f})")
