"""Per-launch timing of the HIP kernels with HIP events on the launch stream, plus the algorithmic bytes / FLOPs of
each launch (what the op must move/compute if every tensor is touched exactly once): the inputs of bench.py's
`roofline` object.  Tracing is off unless `trace()` is active; the product path never pays for it."""
import contextlib

import torch

from .nn import _ops


_SPIN_CYCLES = 60_000_000  # ~25-30 ms at 2.1-2.4 GHz: longer than the host needs to enqueue one instrumented step


class Trace:
    def __init__(self):
        self.records = []  # (kernel, alg_bytes, alg_flops, start_event, end_event)

    def launch(self, kernel, alg_bytes, alg_flops, note=""):
        return _Launch(self, kernel, alg_bytes, alg_flops, note)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for k, b, f, s, e, _ in self.records:
            a = agg.setdefault(k, {"launches": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0})
            a["launches"] += 1
            a["ms"] += s.elapsed_time(e)
            a["bytes"] += b
            a["flops"] += f
        return agg


class _Launch:
    def __init__(self, tr, kernel, b, f, note=""):
        self.tr, self.kernel, self.b, self.f, self.note = tr, kernel, b, f, note

    def __enter__(self):
        self.s = torch.cuda.Event(enable_timing=True)
        self.e = torch.cuda.Event(enable_timing=True)
        self.s.record()  # torch's current stream == the stream the kernel is launched on (_lib.stream())

    def __exit__(self, *a):
        self.e.record()
        self.tr.records.append((self.kernel, self.b, self.f, self.s, self.e, self.note))


@contextlib.contextmanager
def trace():
    t = Trace()
    _ops.TRACE = t
    try:
        yield t
    finally:
        _ops.TRACE = None


def _pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC pass (profiles/*_pmc_traffic.json, written by
    tools/round_profile.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs with the guide's gfx950 correction), or None."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel)
    except (OSError, ValueError, KeyError):
        return None, None
    return (k["hbm_bytes_per_launch"], os.path.basename(files[-1])) if k else (None, None)


def dominant_kernel_roofline(step_fn, steps, hbm_peak_gbs, mfma_peak_tflops):
    step_fn()
    torch.cuda.synchronize()
    with trace() as t:
        for _ in range(steps):
            # A spin kernel first, so that the host enqueues the whole step (launches + events) while the GPU is still busy:
            # the events then bracket back-to-back kernel executions.  Without it every bracket also contains the host's
            # ~6 us submission gap, which doubles the apparent time of the many 5-10 us launches (and disagrees with rocprofv3).
            torch.cuda._sleep(_SPIN_CYCLES)
            step_fn()
            torch.cuda.synchronize()
    agg = t.summary()
    total = sum(a["ms"] for a in agg.values())
    # The roofline is priced for the dominant kernel of the FORWARD stage.  The NMS stage (key build + per-image greedy
    # suppression: one workgroup per image, inherently sequential, neither HBM- nor MFMA-bound) runs on the post-processing stream
    # under the next batch's forward; it stays in top5 and its time is inside `value`, but a bandwidth fraction says nothing about it.
    name, a = max(((k, v) for k, v in agg.items() if not k.startswith("nms(")), key=lambda kv: kv[1]["ms"])
    avg_us = a["ms"] * 1e3 / a["launches"]
    bytes_per = a["bytes"] / a["launches"]
    flops_per = a["flops"] / a["launches"]
    balance = mfma_peak_tflops * 1e12 / (hbm_peak_gbs * 1e9)
    out = {"kernel": name, "launches_per_step": a["launches"] // steps, "avg_launch_us": round(avg_us, 3),
           "share_of_traced_device_time": round(a["ms"] / total, 4)}
    if flops_per / max(bytes_per, 1.0) >= balance:
        ach = flops_per / (avg_us * 1e-6) / 1e12
        out.update({"bound": "mfma", "achieved": round(ach, 3), "peak": mfma_peak_tflops, "unit": "TFLOP/s", "frac": round(ach / mfma_peak_tflops, 4)})
    else:
        ach = bytes_per / (avg_us * 1e-6) / 1e9
        out.update({"bound": "hbm", "achieved": round(ach, 2), "peak": hbm_peak_gbs, "unit": "GB/s", "frac": round(ach / hbm_peak_gbs, 4)})
    out["alg_bytes_per_launch"] = int(bytes_per)
    out["alg_flops_per_launch"] = int(flops_per)
    # HBM bytes per launch from PMC counters: they need their own rocprofv3 passes, so the number comes from the committed
    # summary of the newest pass over this same bench command (null when that pass did not see this kernel)
    out["traffic"], src = _pmc_traffic(name)
    if src:
        out["traffic_source"] = "profiles/" + src
    out["scope"] = "dominant kernel of the forward stage; nms(score+sort_greedy) is latency-bound (one workgroup per image) and overlapped on the post-processing stream"
    out["top5"] = [{"kernel": k, "ms_per_step": round(v["ms"] / steps, 4), "launches_per_step": v["launches"] // steps}
                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:5]]
    return out
