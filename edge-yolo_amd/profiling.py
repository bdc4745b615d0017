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

    def launch(self, kernel, alg_bytes, alg_flops, note="", kernels=1):
        return _Launch(self, kernel, alg_bytes, alg_flops, note, kernels)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for k, b, f, s, e, _, nk in self.records:
            a = agg.setdefault(k, {"launches": 0, "kernels": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0})
            a["launches"] += 1
            a["kernels"] += nk
            a["ms"] += s.elapsed_time(e)
            a["bytes"] += b
            a["flops"] += f
        return agg


class _Launch:
    def __init__(self, tr, kernel, b, f, note="", kernels=1):
        self.tr, self.kernel, self.b, self.f, self.note, self.kernels = tr, kernel, b, f, note, kernels

    def __enter__(self):
        self.s = torch.cuda.Event(enable_timing=True)
        self.e = torch.cuda.Event(enable_timing=True)
        self.s.record()  # torch's current stream == the stream the kernel is launched on (_lib.stream())

    def __exit__(self, *a):
        if a and a[0] is not None:  # the call raised (e.g. EY_EUNSUPPORTED: nothing was launched): no record
            return
        self.e.record()
        self.tr.records.append((self.kernel, self.b, self.f, self.s, self.e, self.note, self.kernels))


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


def _price(bytes_per, flops_per, avg_us, hbm_peak_gbs, mfma_peak_tflops):
    """bound + achieved + frac of one kernel: MFMA-bound when its arithmetic intensity exceeds the machine balance, else HBM-bound."""
    balance = mfma_peak_tflops * 1e12 / (hbm_peak_gbs * 1e9)
    if flops_per / max(bytes_per, 1.0) >= balance:
        ach = flops_per / (avg_us * 1e-6) / 1e12
        return {"bound": "mfma", "achieved": round(ach, 3), "peak": mfma_peak_tflops, "unit": "TFLOP/s", "frac": round(ach / mfma_peak_tflops, 4)}
    ach = bytes_per / (avg_us * 1e-6) / 1e9
    return {"bound": "hbm", "achieved": round(ach, 2), "peak": hbm_peak_gbs, "unit": "GB/s", "frac": round(ach / hbm_peak_gbs, 4)}


def step_roofline(step_fn, steps, ms_per_step, hbm_peak_gbs, mfma_peak_tflops, csv_path=None):
    """The `roofline` object of bench.py: an instrumented eager pass of the same step brackets every launch with HIP events on the
    launch stream and aggregates by kernel instantiation (the grouping of `rocprofv3 --stats`).

    * top level = the DOMINANT kernel (largest share of traced device time, NMS included): bound / achieved / peak / frac / traffic;
    * `table`  = every kernel: launches per step, average launch us, algorithmic bytes and FLOPs per launch, bound, frac -- so every
      fraction can be recomputed from the committed file (csv_path writes the same rows as CSV for profiles/);
    * `step`   = whole-step view against the measured ms_per_step of the timed (pipelined) run: algorithmic bytes / FLOPs of all
      launches of one step divided by the step time, as fractions of the HBM and dense-f16 MFMA peaks."""
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
    rows = []
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        n = a["launches"]
        avg_us, bytes_per, flops_per = a["ms"] * 1e3 / n, a["bytes"] / n, a["flops"] / n
        r = {"kernel": k, "launches_per_step": round(n / steps, 2), "avg_launch_us": round(avg_us, 3), "ms_per_step": round(a["ms"] / steps, 4),
             "share": round(a["ms"] / total, 4), "alg_bytes_per_launch": int(bytes_per), "alg_flops_per_launch": int(flops_per)}
        if a["kernels"] != n:  # an operator that is several kernels behind one C call (bracketed as a whole): rocprofv3 splits it (profiles/)
            r["kernels_per_launch"] = round(a["kernels"] / n, 2)
        r.update(_price(bytes_per, flops_per, avg_us, hbm_peak_gbs, mfma_peak_tflops))
        rows.append(r)
    top = next((r for r in rows if "kernels_per_launch" not in r), rows[0])  # a single kernel (composite operators are priced in `table`)
    out = {"kernel": top["kernel"], "launches_per_step": top["launches_per_step"], "avg_launch_us": top["avg_launch_us"],
           "share_of_traced_device_time": top["share"]}
    out.update({k: top[k] for k in ("bound", "achieved", "peak", "unit", "frac", "alg_bytes_per_launch", "alg_flops_per_launch")})
    # HBM bytes per launch from PMC counters: they need their own rocprofv3 passes, so the number comes from the committed
    # summary of the newest pass over this same bench command (null when that pass did not see this kernel)
    out["traffic"], src = _pmc_traffic(top["kernel"])
    if src:
        out["traffic_source"] = "profiles/" + src
    out["scope"] = ("dominant kernel = largest share of the traced device time of one step (every kernel of the step is in `table`, the NMS "
                    "stage included); averages from HIP events on the launch stream over an instrumented eager pass of the same step")
    step_bytes = sum(a["bytes"] for a in agg.values()) / steps
    step_flops = sum(a["flops"] for a in agg.values()) / steps
    sec = ms_per_step * 1e-3
    out["step"] = {"launches_per_step": round(sum(a["kernels"] for a in agg.values()) / steps, 1), "traced_device_ms_per_step": round(total / steps, 4),
                   "alg_bytes_per_step": int(step_bytes), "alg_flops_per_step": int(step_flops),
                   "hbm_frac": round(step_bytes / sec / (hbm_peak_gbs * 1e9), 4), "mfma_frac": round(step_flops / sec / (mfma_peak_tflops * 1e12), 4),
                   "ms_per_step": ms_per_step}
    out["table"] = rows
    if csv_path:
        import csv
        cols = ["kernel", "launches_per_step", "avg_launch_us", "ms_per_step", "share", "alg_bytes_per_launch", "alg_flops_per_launch", "bound", "achieved", "peak",
                "unit", "frac"]
        with open(csv_path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=cols)
            w.writeheader()
            for r in rows:
                w.writerow({c: r[c] for c in cols})
    return out
