"""Developer tool: layers 2.cv2 + 3 at the benchmark shape (batch 32, 160x160, 16 + 32 -> 64 -> 64 / s2, f16): one kernel (ey_conv_pw_conv3s2) vs two launches, graph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edge_yolo_amd  # noqa: F401
from edge_yolo_amd import _lib as L
from edge_yolo_amd.nn import _ops, modules as M

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cv2, c3 = M.Conv(48, 64, 1, 1).cuda().half().eval(), M.Conv(64, 64, 3, 2).cuda().half().eval()
t = L.empty_nhwc(B, 32, 160, 160, torch.float16, "cuda"); t.copy_((torch.rand(B, 32, 160, 160) - 0.5).half())
buf = L.empty_nhwc(B, 32, 160, 160, torch.float16, "cuda"); buf.copy_((torch.rand(B, 32, 160, 160) - 0.5).half())
srcs = [t[:, :16], buf]
y = L.empty_nhwc(B, 64, 80, 80, torch.float16, "cuda")


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * reps) * 1e3


print(f"two launches: {timed(lambda: c3(_ops.conv2d(cv2, srcs, cv2.folded, 1, 1, 0, L.ACT_SILU), out=y)):6.1f} us")
for form, skew in ((2, 0), (1, 0), (1, 2), (1, 4), (1, 6), (1, 8), (1, 12)):
    L.check(L.lib().ey_tune_set(b"pw3", form), "t")
    L.check(L.lib().ey_tune_set(b"pw3_skew", skew), "t")
    us = timed(lambda: _ops.pw_conv3s2(cv2, c3, srcs, out=y))
    print(f"pw3 form {form} skew {skew}:   {us:6.1f} us  ({(B * 160 * 160 * 48 + y.numel()) * 2 / us / 1e3:5.0f} GB/s algorithmic)")
