#!/usr/bin/env python
"""Headline benchmark: images/s of the EdgeLine-YOLO detection forward path (stem -> backbone -> DWT neck ->
GFLv2 head decode -> batched NMS) at 640x640, fp16 storage, synthetic images already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model yolo11n-test.yaml] [--batch 32] [--imgsz 640] [--nc 80]

N>1: one process per GPU, images sharded across ranks (weak scaling: 32 images per GPU), one RCCL all_gather of the padded
(B,300,6) result rows + counts per `--gather-every` steps.  Rank 0 prints ONE JSON line.  Two launch forms:
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK/LOCAL_RANK/WORLD_SIZE from the environment), or
plain `python bench.py --gpus N`: the parent -- which has made no GPU call -- starts N fresh rank processes of this script
(`self_launch`), relays rank 0's JSON line and exits with the ranks' worst code (the reference launches its own ranks too,
ultralytics/utils/dist.py:25-67).

A "step" = one batch through the whole device path, replayed from captured hipGraphs: nothing is skipped (NMS and,
for N>1, the gather are inside the timed region).  `roofline` = per-kernel HIP-event timing of an instrumented eager pass of the
same step (dominant kernel at the top level, every kernel in `roofline.table`, whole-step fractions in `roofline.step`);
`cpu_baseline` = the CPU oracle (oracle/, a port of the reference's torch-CPU path, test infrastructure) timed on this box's host
cores, rank 0, N=1, with its calibration against the true reference (profiles/*_cpu_calibration.json, measured where both run);
`predict_batches` = images/s through the public API (`YOLO.predict_batches`: pinned host tensors in, `Results` out).

All behaviour is selected by the flags below: this file and the library read no EY_* environment variables.
"""
import argparse
import contextlib
import glob
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="yolo11n-test.yaml")
    ap.add_argument("--nc", type=int, default=80, help="number of classes (GC10-DET: 10)")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the YOLO.predict_batches (host tensors in, Results out) measurement")
    ap.add_argument("--cpu-images", type=int, default=64)
    ap.add_argument("--no-pipeline", action="store_true", help="one graph per batch, no overlap between consecutive batches")
    ap.add_argument("--stages", default="auto", help="pipeline stages: 2, 3, 4 or 'auto' (times 3 and 4 at start-up and keeps the faster)")
    ap.add_argument("--cuts", default="", help="explicit layer indices where a new pipeline stage starts, e.g. 9,20,23 (overrides --stages)")
    ap.add_argument("--nms-stage", action="store_true", help="experiment: decode + NMS as a pipeline stage of their own")
    ap.add_argument("--gather-every", type=int, default=8, help="N>1: exchange the result rows of this many steps with one all_gather")
    ap.add_argument("--force-gather", action="store_true", help="exercise the RCCL gather path on a single GPU (world_size 1)")
    ap.add_argument("--roofline-csv", default="", help="also write roofline.table as CSV (for profiles/)")
    ap.add_argument("--regime", default="dense", choices=["dense", "sparse"],
                    help="NMS regime (SURVEY.md §8d): dense = default-init head biases (every slot of max_det fills: the heaviest case, the headline); "
                         "sparse = Detect.bias_init biases (ultralytics/nn/modules/head.py:150-161), what a trained detector's score surface looks like")
    ap.add_argument("--cls-bias-shift", type=float, default=0.0, help="--regime sparse: added to the class biases (0 = the reference's bias_init formula)")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="test hook: run the launch / barrier / gather / report control flow on the gloo backend with a stub step and NO device work "
                         "(the JSON line says so; never a measurement)")
    ap.add_argument("--tune", action="append", default=[], metavar="NAME=VALUE",
                    help="developer option: set a dispatch tunable (ey_tune_set, csrc/tune.h) before anything is launched; recorded in the JSON line")
    return ap.parse_args(argv)


def sparse_biases(sd, head_index, nc, strides, cls_shift=0.0):
    """The `Detect.bias_init` values (ultralytics/nn/modules/head.py:150-161: box 1.0, cls log(5/nc/(640/s)^2); the one2one_* half too
    when the head has it) written into a state_dict: the SPARSE NMS regime of SURVEY.md §8(d).  cls_shift is added to the class biases
    (0 = the reference formula; > 0 moves the random-init score surface towards what a trained detector produces at conf 0.25)."""
    for pre in ("", "one2one_"):
        for i, s in enumerate(strides):
            kb, kc = f"model.{head_index}.{pre}cv2.{i}.2.bias", f"model.{head_index}.{pre}cv3.{i}.2.bias"
            if kb in sd:
                sd[kb] = torch.ones_like(sd[kb])
                c = sd[kc].clone()
                c[:nc] = math.log(5 / nc / (640 / float(s)) ** 2) + cls_shift
                sd[kc] = c
    return sd


def build_model(name, dtype, device, seed=0, nc=None, regime="dense", cls_shift=0.0):
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd.nn.tasks import DetectionModel
    import synthdata as synth  # name-keyed synthetic weights (neutral module shared with the tests; not part of the oracle)
    m = DetectionModel(name, nc=nc)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed)
    if regime == "sparse":
        sparse_biases(sd, len(m.model) - 1, m.model[-1].nc, [float(v) for v in m.stride], cls_shift)
    m.load_state_dict(sd)
    m = m.to(device).fuse()
    m = m.half() if dtype == torch.float16 else m.float()
    return m.eval(), sd


def cpu_baseline(name, sd, imgsz, n_images, conf, iou, nc=None):
    """The CPU port of the reference path (oracle/) on this box's host cores: forward + NMS, fp32."""
    from oracle import model as om, nms as onms
    ncpu = os.cpu_count() or 1
    threads = max(1, min(8, ncpu - 1))  # the reference's select_device rule (utils/torch_utils.py:225-226)
    torch.set_num_threads(threads)
    o = om.OracleModel(os.path.join(ROOT, "edge-yolo_amd", "cfg", "models", "11", name), {k: v.float() for k, v in sd.items()}, nc=nc)
    g = torch.Generator().manual_seed(0)
    bs = min(8, n_images)
    x = torch.rand(bs, 3, imgsz, imgsz, generator=g)
    o(x[:1])  # warm-up
    t0 = time.perf_counter()
    done, t_fwd = 0, 0.0
    while done < n_images:
        t1 = time.perf_counter()
        y, _ = o(x)
        t_fwd += time.perf_counter() - t1
        onms.non_max_suppression(y.numpy(), conf, iou)
        done += bs
    dt = time.perf_counter() - t0
    out = {"value": round(done / dt, 3), "unit": "images/s", "cores": threads, "kind": "port", "forward_only_images_s": round(done / t_fwd, 3),
           "sample": f"{done} images {imgsz}x{imgsz} fp32, batch {bs}, forward+NMS, oracle/ (torch-CPU port of the reference path), {threads} threads"}
    cal = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_cpu_calibration.json")))
    if cal:  # port vs TRUE reference, forward only, measured in the build container (tools/cpu_calibration.py): the reference cannot travel here
        try:
            c = json.load(open(cal[-1]))
            out["calibration"] = {"port_over_reference": c["port_over_reference"], "reference_img_s": c["reference_img_s"], "port_img_s": c["port_img_s"],
                                  "what": c["what"], "where": c["host"], "images": c["images"], "source": "profiles/" + os.path.basename(cal[-1])}
        except (OSError, ValueError, KeyError):
            pass
    return out


def pipeline_cuts(nlayers, nstages):
    """Layer indices where a new pipeline stage starts.  24-layer YAMLs: 4 stages = backbone | backbone tail + neck | last neck block |
    head + decode + NMS (cuts at ~39 % / ~87 % of the layer list and before the head); fewer stages drop the middle cuts."""
    c1, c2 = max(1, round(0.39 * (nlayers - 1))), max(2, round(0.87 * (nlayers - 1)))
    return {2: [nlayers - 1], 3: sorted({c1, nlayers - 1}), 4: sorted({c1, c2, nlayers - 1})}[nstages]


def timed_steps(step, drain, gather, steps, warmup, post_ctx, sync, barrier, reduce_max, on_block=None):
    """The bench's control flow: `warmup` untimed steps, then EXACTLY `steps` timed ones bracketed by barrier + device sync on both
    sides; the gather of every step's rows (and the flush of the last, possibly partial, block) is inside the timed region.  Shared
    by the real run and by the world-size-2 gloo rehearsal in tests/ (stub step, CPU tensors).  Returns (seconds, last step result)."""
    def one():
        res = step()
        if gather is not None:
            with post_ctx():  # the gather follows this batch's NMS on the post-processing stream
                k = gather(*res)
            if k is not None and on_block is not None:
                on_block(k)
        return res

    def finish():
        drain()
        if gather is not None:
            with post_ctx():
                k = gather.flush()
            gather.wait()
            if k is not None and on_block is not None:
                on_block(k)

    for _ in range(warmup):
        one()
    finish()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    res = None
    for _ in range(steps):
        res = one()
    finish()
    sync()
    barrier()
    sync()
    return reduce_max(time.perf_counter() - t0), res


def choose_variant(variants, trial, reduce_max):
    """Start-up auto-tune: `variants` = {name: builder}; every rank builds and times each one with `trial(obj) -> seconds`; the times
    are max-reduced over the ranks, so all ranks pick the same winner.  Returns (name, object, {name: seconds})."""
    best, times, objs = None, {}, {}
    for name, make in variants.items():
        objs[name] = make()
        times[name] = reduce_max(trial(objs[name]))
        if best is None or times[name] < times[best]:
            best = name
    for name in list(objs):
        if name != best:
            del objs[name]
    return best, objs[best], times


def api_throughput(name, sd, nc, batch, imgsz, half, steps, warmup, conf, iou, max_det, device, u8=False):
    """images/s through the public API: YOLO.predict_batches with pinned HOST tensors in and Results out (PCIe upload included).
    u8: decoded image batches (uint8 (B,H,W,3) BGR) instead of normalised float tensors -- converted on the device."""
    import edge_yolo_amd
    y = edge_yolo_amd.YOLO(name, nc=nc)
    y.model.load_state_dict(sd)
    dt = torch.uint8 if u8 else torch.float16 if half else torch.float32
    if u8:
        x = torch.randint(0, 256, (batch, imgsz, imgsz, 3), generator=torch.Generator().manual_seed(0), dtype=torch.uint8).pin_memory()
    else:
        x = torch.rand(batch, 3, imgsz, imgsz, generator=torch.Generator().manual_seed(0)).to(dt).pin_memory()
    total = warmup + steps
    t0, n, ndet = None, 0, 0
    for res in y.predict_batches((x for _ in range(total)), half=half, conf=conf, iou=iou, max_det=max_det, device=device):
        n += 1
        if n == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        elif n > warmup:
            ndet += sum(len(r) for r in res)
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    return {"value": round(steps * batch / sec, 2), "unit": "images/s", "ms_per_batch": round(sec / steps * 1e3, 4), "batches": steps,
            "what": f"YOLO.predict_batches(half={half}): pinned host {str(dt).split('.')[-1]} tensors ({x.numel() * x.element_size() / 1e6:.1f} MB per batch over PCIe) in, "
                    f"Results (device boxes + host counts) out; mean detections/img {ndet / max(1, steps * batch):.0f}"}


def self_launch(a, argv):
    """`python bench.py --gpus N` typed as is (no WORLD_SIZE in the environment): start N fresh rank processes of this script, one per
    GPU, relay rank 0's JSON line, exit with the ranks' worst code.  The parent has imported torch but made NO GPU call and never
    execs; the ranks are ordinary children (a process that has touched the GPU must not be replaced).  Reference counterpart:
    ultralytics/utils/dist.py:25-67 (generate_ddp_command: the reference builds and runs its own launch command)."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0's stdout is the JSON line; the other ranks' stdout goes to stderr so that exactly one line reaches the caller's stdout
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    line = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    for ln in (out0 or "").splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    worst = max((abs(c) for c in codes), default=0)
    if line:
        print(line[-1], flush=True)
    elif worst == 0:
        worst = 1
    if worst:
        print(f"bench.py: rank exit codes {codes}" + ("" if line else "; rank 0 printed no JSON line"), file=sys.stderr)
    return min(worst, 255)


def rehearse_cpu(a, world, rank):
    """--rehearse-cpu: the launch / rendezvous / barrier / gather / report control flow with a stub step on the gloo backend.  Test hook
    for the self-launcher (tests/test_dist_cpu.py); it performs no device work and its JSON line says so."""
    from edge_yolo_amd import dist as eydist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, md = a.batch, 300
    n = [0]

    def step():
        k = n[0]
        n[0] += 1
        boxes = torch.zeros(B, md, 6)
        count = torch.full((B,), 1, dtype=torch.int32)
        boxes[:, 0, 0] = torch.arange(B) + (k * world + rank) * B  # row 0, column 0 = global image number
        return boxes, count

    def reduce_max(v):
        if world > 1:
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t)
        return v

    gather = eydist.BoxGatherer(world, B, md, "cpu", own_stream=False, every=a.gather_every) if world > 1 else None
    seen = []
    dt, _ = timed_steps(step, lambda: None, gather, a.steps, a.warmup, contextlib.nullcontext, lambda: None,
                        (dist.barrier if world > 1 else (lambda: None)), reduce_max,
                        on_block=lambda k: seen.extend(float(r[0, 0]) for r in gather.results(k)))
    if gather is not None:  # every rank holds every image's rows, in global order (warm-up blocks included)
        assert seen == [float(v) for v in range((a.warmup + a.steps) * world * B)], seen[:8]
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        return {"metric": "REHEARSAL (control flow only: gloo backend, stub step, no device work) -- not a measurement", "value": 0.0,
                "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
                "data": "none", "rows_checked": len(seen)}
    return None


def make_pipeline(model, images, cuts, head_nms, conf, iou, max_det, streams=None, nms_stage=False):
    """The software pipeline bench.py times: the layer list cut at `cuts`, every stage a captured hipGraph on its own stream; the last
    model stage (the head) also decodes and runs the NMS, so head + decode + NMS of batch i run beside the backbone / neck of batch
    i+1.  Every buffer set is pre-filled with `images` (resident batch: no per-step copy).  Shared with the -m gpu tests."""
    from edge_yolo_amd.engine.predictor import PipelinedRunner
    from edge_yolo_amd.utils import ops
    nlayers = len(model.model)
    post = lambda st: ops.nms_device(st[0][0] if isinstance(st[0], (tuple, list)) else st[0], conf, iou, max_det=max_det)[:2]  # noqa: E731
    bounds = [0] + list(cuts) + [nlayers]
    stages = []
    for k in range(len(bounds) - 1):
        lo, hi = bounds[k], bounds[k + 1]
        if k == 0:
            stages.append(lambda im, lo=lo, hi=hi: model.forward_layers((im, []), lo, hi, head_nms=head_nms))
        else:
            stages.append(lambda stt, lo=lo, hi=hi: model.forward_layers(stt, lo, hi, head_nms=head_nms))
    if nms_stage:
        stages.append(post)
    else:
        last = stages.pop()
        stages.append(lambda stt, last=last: post(last(stt)))
    pipe = PipelinedRunner(*stages, images, streams=streams)
    for j in range(pipe.nsets):
        pipe.static_input(j).copy_(images)
    pipe.cuts = list(cuts)
    return pipe


@contextlib.contextmanager
def stdout_to_stderr():
    """File descriptor 1 -> 2 for the duration: the collective backend prints its banner ("RCCL version : ...") to stdout when the first
    communicator is created, and stdout must carry exactly ONE line, the JSON."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a, argv))
    with stdout_to_stderr():
        out = run(a)
    if out is not None:
        print(json.dumps(out), flush=True)


def run(a):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}, or unset WORLD_SIZE and let bench.py start its own ranks")
    if a.rehearse_cpu:
        return rehearse_cpu(a, world, rank)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_gather = world > 1 or a.force_gather
    if use_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    dtype = torch.float16 if a.dtype == "f16" else torch.float32
    conf, iou, max_det = 0.25, 0.7, 300

    if a.tune:
        from edge_yolo_amd import _lib
        for kv in a.tune:
            k, v = kv.split("=")
            _lib.check(_lib.lib().ey_tune_set(k.encode(), int(v)), f"--tune {kv}")
    from edge_yolo_amd.engine.predictor import GraphRunner
    from edge_yolo_amd.utils import ops
    from edge_yolo_amd import dist as eydist
    model, sd = build_model(a.model, dtype, dev, nc=a.nc, regime=a.regime, cls_shift=a.cls_bias_shift)

    g = torch.Generator(device=dev).manual_seed(rank)
    images = torch.rand(a.batch, 3, a.imgsz, a.imgsz, generator=g, device=dev).to(dtype)  # resident in HBM before timing

    head_nms = {"conf": conf, "classes": None}  # predict mode: the head decode builds the NMS candidates in the same pass (as YOLO.predict does)

    def device_step(im):
        cand, _ = model(im, head_nms=head_nms)
        boxes, count, index = ops.nms_device(cand, conf, iou, max_det=max_det)
        return boxes, count

    def reduce_max(sec):
        if world > 1:
            t = torch.tensor([sec], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return sec

    def barrier():
        if world > 1:
            dist.barrier()

    # pipelined mode: the gather rides on the post-processing stream (no stream of its own) and exchanges the rows of
    # --gather-every consecutive batches with one all_gather (see BoxGatherer)
    gather = eydist.BoxGatherer(world, a.batch, max_det, dev, own_stream=a.no_pipeline, every=a.gather_every) if use_gather else None
    chosen, trial_ms = None, None
    if a.no_pipeline:
        runner = GraphRunner(device_step)
        images = runner.static_input(images).copy_(images)  # the batch lives in the graph's input buffer: no per-step copy
        step, drain, post_ctx = (lambda: runner(images)), (lambda: None), contextlib.nullcontext
    else:
        # software pipeline over consecutive batches: every stage is a captured hipGraph on its own stream; stage s of batch i overlaps
        # stage s-1 of batch i+1 (every batch still runs every stage in full; the timed region ends with a drain of all stages)
        model.model[-1].head_streams = False  # the head runs as a pipeline stage of its own: no fork inside it (measured: 1.93 -> 1.89 ms)
        nlayers = len(model.model)
        shared_streams = [torch.cuda.Stream(device=dev) for _ in range(5)]  # one set for every candidate pipeline (hardware-queue binding, see PipelinedRunner)

        def make_pipe(cuts):
            return make_pipeline(model, images, cuts, head_nms, conf, iou, max_det, streams=shared_streams, nms_stage=a.nms_stage)

        def pipe_fns(pipe):
            def step():
                return pipe.outputs(pipe.submit())
            return step, pipe.wait, (lambda: torch.cuda.stream(pipe.sp))

        if a.cuts:
            pipe = make_pipe([int(v) for v in a.cuts.split(",") if v])
        elif a.stages != "auto":
            pipe = make_pipe(pipeline_cuts(nlayers, int(a.stages)))
        else:
            # Active streams = hardware queues (4 on this stack).  Which of 3 / 4 stages wins depends on what else is active (with a process
            # group the collective backend's stream is one more): measured at start-up, same decision on every rank (max-reduced times).
            def trial(p):
                s, d, pc = pipe_fns(p)
                return timed_steps(s, d, gather, 12, 4, pc, torch.cuda.synchronize, barrier, lambda v: v)[0]
            chosen, pipe, tt = choose_variant({f"{n}": (lambda n=n: make_pipe(pipeline_cuts(nlayers, n))) for n in (3, 4)}, trial, reduce_max)
            trial_ms = {k: round(v / 12 * 1e3, 4) for k, v in tt.items()}
            torch.cuda.empty_cache()
        step, drain, post_ctx = pipe_fns(pipe)

    dt, (boxes, count) = timed_steps(step, drain, gather, a.steps, a.warmup, post_ctx, torch.cuda.synchronize, barrier, reduce_max)

    out = None
    if rank == 0:
        ms = dt / a.steps * 1e3
        total_images = a.batch * world * a.steps
        pipe_desc = ("single graph per batch" if a.no_pipeline else
                     f"{len(pipe.cuts) + 1}-stage software pipeline over consecutive batches (layer cuts {pipe.cuts}; last stage = head + decode + NMS), "
                     "one hipGraph and one HIP stream per stage" + (f"; stage count auto-tuned at start-up, trial ms/step {trial_ms}" if trial_ms else ""))
        regime = ("random-init weights, default head biases (DENSE NMS regime" if a.regime == "dense" else
                  f"random-init weights + Detect.bias_init head biases{f' + {a.cls_bias_shift:g} on the class biases' if a.cls_bias_shift else ''} (SPARSE NMS regime")
        out = {
            "metric": f"images/sec @ {a.imgsz}x{a.imgsz} {'fp16' if a.dtype == 'f16' else 'fp32'} (EdgeLine-YOLO detection forward path: backbone + DWT neck + GFLv2 head decode + batched NMS)",
            "value": round(total_images / dt, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
            "data": "synthetic",
            "config": {"workload": f"{a.model} (scale n, nc={a.nc}) predict path, {a.imgsz}x{a.imgsz}, batch {a.batch}/GPU, "
                                   f"conf {conf} iou {iou} max_det {max_det}, {regime}: mean detections/img "
                                   f"{float(count.float().mean()):.0f})",
                       "regime": a.regime, "global_batch": a.batch * world, "imgsz": a.imgsz, "nc": a.nc, "pipeline": pipe_desc, **({"tunables": a.tune} if a.tune else {}),
                       "sharding": f"images x{world}" + (f", RCCL all_gather of boxes every {gather.every} steps" if world > 1 else "")},
        }
    if rank == 0 and not a.no_roofline:
        from edge_yolo_amd import profiling
        torch.cuda.synchronize()
        model.model[-1].head_streams = False
        out["roofline"] = profiling.step_roofline(lambda: device_step(images), steps=min(a.steps, 10), ms_per_step=out["ms_per_step"], hbm_peak_gbs=HBM_PEAK_GBS,
                                                  mfma_peak_tflops=MFMA_F16_PEAK_TFLOPS, csv_path=a.roofline_csv or None)
    barrier()
    if gather is not None and rank == 0:
        rows = gather.results()
        assert len(rows) % (a.batch * world) == 0 and len(rows) > 0 and all(r.shape[1] == 6 for r in rows)
    if rank == 0 and world == 1 and not a.no_api and not a.force_gather:
        if not a.no_pipeline:
            del pipe, step, drain
        torch.cuda.empty_cache()
        out["predict_batches"] = api_throughput(a.model, sd, a.nc, a.batch, a.imgsz, a.dtype == "f16", min(a.steps, 40), 8, conf, iou, max_det, dev)
        out["predict_batches_u8"] = api_throughput(a.model, sd, a.nc, a.batch, a.imgsz, a.dtype == "f16", min(a.steps, 40), 8, conf, iou, max_det, dev, u8=True)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.model, sd, a.imgsz, a.cpu_images, conf, iou, nc=a.nc)
    if use_gather:
        dist.destroy_process_group()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
