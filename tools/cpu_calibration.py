"""Calibrates bench.py's `cpu_baseline` (the CPU port in oracle/, kind "port") against the TRUE reference:
both run the same fused fp32 forward on the same images and threads in THIS container (the only place /root/reference exists);
the ratio is committed under profiles/ and bench.py copies it into the `cpu_baseline` object.  Forward only: the reference's NMS
needs torchvision, which is not installed (the port's NMS is timed inside bench.py's cpu_baseline).

    python tools/cpu_calibration.py [--images 64] [--out profiles/r02_cpu_calibration.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _ref_import  # noqa: E402

_ref_import.setup()
import torch  # noqa: E402

import synthdata as synth  # noqa: E402
from oracle import model as om  # noqa: E402
from ultralytics.nn.tasks import DetectionModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=64)
ap.add_argument("--model", default="yolo11n-test.yaml")
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_cpu_calibration.json"))
a = ap.parse_args()
torch.set_grad_enabled(False)
ncpu = os.cpu_count() or 1
threads = max(1, min(8, ncpu - 1))  # the reference's select_device rule (utils/torch_utils.py:225-226)
torch.set_num_threads(threads)
ref = DetectionModel(a.model, ch=3, nc=80, verbose=False).eval()
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()})
ref.load_state_dict(sd)
ref.fuse(verbose=False)
port = om.OracleModel(os.path.join(ROOT, "edge-yolo_amd", "cfg", "models", "11", a.model), sd)
x = torch.rand(8, 3, 640, 640, generator=torch.Generator().manual_seed(0))


def rate(fn):
    fn(x[:1])
    best = 0.0
    for _ in range(2):  # two passes, best of: the container's cores are shared
        t0 = time.perf_counter()
        done = 0
        while done < a.images:
            fn(x)
            done += x.shape[0]
        best = max(best, done / (time.perf_counter() - t0))
    return best


r_ref = rate(lambda t: ref(t))
r_port = rate(lambda t: port(t))
err = float((ref(x[:2])[0] - port(x[:2])[0]).abs().max())
out = {"model": a.model, "images": a.images, "batch": 8, "imgsz": 640, "threads": threads, "what": "fused fp32 forward only (no NMS)",
       "reference_img_s": round(r_ref, 2), "port_img_s": round(r_port, 2), "port_over_reference": round(r_port / r_ref, 4),
       "max_abs_diff_outputs": err, "host": "build container (8 vCPU)", "torch": torch.__version__}
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out))
