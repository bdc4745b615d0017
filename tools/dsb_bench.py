"""Developer tool: DSBottleneck (k3 -> k5/k7 DSConv pair + residual) at the shapes of the EdgeLine-n step, one band kernel (ey_dsb_pair) against the
two-launch form, graph-replayed; sweeps the rows per band.   python tools/dsb_bench.py [--batch 32] [--rb 0,2,3,4,5,8,10]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edge_yolo_amd  # noqa: F401
from edge_yolo_amd import _lib as L
from edge_yolo_amd.nn import modules as M

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--rb", default="0,2,3,4,5,8,10")
ap.add_argument("--p2", default="0")
ap.add_argument("--shapes", default="32:5:40,64:5:20,64:7:40,32:7:80,64:5:40")
a = ap.parse_args()


def tune(k, v):
    L.check(L.lib().ey_tune_set(k.encode(), int(v)), k)


def timed(fn, reps=20):
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


for spec in a.shapes.split(","):
    c, k2, hw = (int(v) for v in spec.split(":"))
    m = M.DSBottleneck(c, c, True, 1.0, 3, k2).cuda().half().eval()
    x = L.empty_nhwc(a.batch, c, hw, hw, torch.float16, "cuda")
    x.copy_((torch.rand(a.batch, c, hw, hw) - 0.5).half())
    y = L.empty_nhwc(a.batch, c, hw, hw, torch.float16, "cuda")
    tune("dsb_max_px", 1 << 40)
    tune("dsb_pair", 0)
    line = f"C{c} k3->k{k2} {hw}x{hw} B{a.batch}: two launches {timed(lambda: m(x, out=y)):6.1f} us | pair"
    tune("dsb_pair", 1)
    for p2 in (int(v) for v in a.p2.split(",")):
        tune("dsb_p2", p2)
        if p2:
            line += f" | p2={p2}"
        for rb in (int(v) for v in a.rb.split(",")):
            tune("dsb_rb", rb)
            line += f"  rb={rb}: {timed(lambda: m(x, out=y)):5.1f}"
    tune("dsb_rb", 0)
    tune("dsb_p2", 0)
    print(line, flush=True)
