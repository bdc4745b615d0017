#!/usr/bin/env python
"""Headline benchmark: images/s of the EdgeLine-YOLO detection forward path (stem -> backbone -> DWT neck ->
GFLv2 head decode -> batched NMS) at 640x640, fp16 storage, synthetic images already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model yolo11n-test.yaml] [--batch 32] [--imgsz 640]

N>1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`; one process per GPU,
images sharded across ranks (weak scaling: 32 images per GPU), one RCCL all_gather of the padded (B,300,6) result
rows + counts per step on a side stream.  Rank 0 prints ONE JSON line.

A "step" = one batch through the whole device path, replayed from a captured hipGraph: nothing is skipped (NMS and,
for N>1, the gather are inside the timed region).  `roofline` = the dominant kernel of the step measured live with HIP
events on the launch stream in an instrumented eager pass of the same steps; `cpu_baseline` = the CPU oracle
(oracle/, a port of the reference's torch-CPU path, test infrastructure) timed on this box's host cores, rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="yolo11n-test.yaml")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=16)
    ap.add_argument("--no-pipeline", action="store_true", help="one graph per batch, no overlap of NMS(i) with forward(i+1)")
    return ap.parse_args()


def build_model(name, dtype, device, seed=0):
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd.nn.tasks import DetectionModel
    from oracle import synth  # synthetic weight generator only (name-keyed, shared with the tests)
    m = DetectionModel(name)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=seed)
    m.load_state_dict(sd)
    m = m.to(device).fuse()
    m = m.half() if dtype == torch.float16 else m.float()
    return m.eval(), sd


def cpu_baseline(name, sd, imgsz, n_images, conf, iou):
    """The CPU port of the reference path (oracle/) on this box's host cores: forward + NMS, fp32."""
    from oracle import model as om, nms as onms
    ncpu = os.cpu_count() or 1
    threads = max(1, min(8, ncpu - 1))  # the reference's select_device rule (utils/torch_utils.py:225-226)
    torch.set_num_threads(threads)
    o = om.OracleModel(os.path.join(ROOT, "edge-yolo_amd", "cfg", "models", "11", name), {k: v.float() for k, v in sd.items()})
    g = torch.Generator().manual_seed(0)
    bs = min(8, n_images)
    x = torch.rand(bs, 3, imgsz, imgsz, generator=g)
    o(x[:1])  # warm-up
    t0 = time.perf_counter()
    done = 0
    while done < n_images:
        y, _ = o(x)
        onms.non_max_suppression(y.numpy(), conf, iou)
        done += bs
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{done} images {imgsz}x{imgsz} fp32, batch {bs}, forward+NMS, oracle/ (torch-CPU port of the reference path), {threads} threads"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_gather = os.environ.get("EY_FORCE_GATHER") == "1"  # exercise the RCCL gather path on a single GPU (world_size 1)
    if world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    dtype = torch.float16 if a.dtype == "f16" else torch.float32
    conf, iou, max_det = 0.25, 0.7, 300

    from edge_yolo_amd.engine.predictor import GraphRunner, PipelinedRunner
    from edge_yolo_amd.utils import ops
    from edge_yolo_amd import dist as eydist
    model, sd = build_model(a.model, dtype, dev)

    g = torch.Generator(device=dev).manual_seed(rank)
    images = torch.rand(a.batch, 3, a.imgsz, a.imgsz, generator=g, device=dev).to(dtype)  # resident in HBM before timing

    def device_step(im):
        pred, _ = model(im)
        boxes, count, index = ops.nms_device(pred, conf, iou, max_det=max_det)
        return boxes, count

    # pipelined mode: the gather rides on the post-processing stream (no stream of its own)
    # ... and exchanges the rows of EY_GATHER_EVERY (8) consecutive batches with one all_gather (see BoxGatherer)
    gather = (eydist.BoxGatherer(world, a.batch, max_det, dev, own_stream=a.no_pipeline, every=int(os.environ.get("EY_GATHER_EVERY", "8")))
              if (world > 1 or force_gather) else None)
    if a.no_pipeline:
        runner = GraphRunner(device_step)
        images = runner.static_input(images).copy_(images)  # the batch lives in the graph's input buffer: no per-step copy

        def step():
            boxes, count = runner(images)
            if gather is not None:
                gather(boxes, count)
            return boxes, count

        def drain():
            pass
    else:
        # software pipeline over consecutive batches: every stage is a captured hipGraph on its own stream; stage s of batch i overlaps
        # stage s-1 of batch i+1 (every batch still runs every stage in full; the timed region ends with a drain of all stages)
        if "EY_HEAD_STREAMS" not in os.environ:  # the head runs as a pipeline stage of its own: no fork inside it (measured: 1.93 -> 1.89 ms)
            import edge_yolo_amd.nn.modules.head as _hm
            _hm._HEAD_STREAMS = False
        nlayers = len(model.model)
        # Active streams = hardware queues (4 on this stack; GPU_MAX_HW_QUEUES 6/8 are worse).  Measured on MI355X, ms/step: 2 stages 1.90,
        # 3 stages 1.82-1.92, 4 stages 1.52 (cuts 9,20,23), 5+ stages alias queues and collapse to 2.35.  With a process group (N > 1) the
        # collective backend's internal stream is the fourth one: 3 stages (cuts 9,23) 1.51, 4 stages 2.34.  Cuts sit at ~39 % / ~87 % of the
        # layer list and before the head (24-layer YAMLs: backbone | backbone tail + neck | last neck block | head + decode + NMS).
        c1, c2 = max(1, round(0.39 * (nlayers - 1))), max(2, round(0.87 * (nlayers - 1)))
        dflt = sorted({c1, nlayers - 1}) if (world > 1 or force_gather) else sorted({c1, c2, nlayers - 1})
        cuts = [int(v) for v in os.environ["EY_PIPE_CUTS"].split(",") if v] if os.environ.get("EY_PIPE_CUTS") else dflt  # layer indices where a new stage starts
        post = lambda st: ops.nms_device(st[0][0] if isinstance(st[0], (tuple, list)) else st[0], conf, iou, max_det=max_det)[:2]  # noqa: E731
        bounds = [0] + cuts + [nlayers]
        stages = []
        for k in range(len(bounds) - 1):
            lo, hi = bounds[k], bounds[k + 1]
            if k == 0:
                stages.append(lambda im, lo=lo, hi=hi: model.forward_layers((im, []), lo, hi))
            else:
                stages.append(lambda stt, lo=lo, hi=hi: model.forward_layers(stt, lo, hi))
        # the last model stage (the head) also decodes and runs the NMS: head + decode + NMS of batch i || backbone / neck of batch i+1
        if os.environ.get("EY_PIPE_NMS_STAGE") == "1":  # experiment: decode + NMS as a stage of their own
            stages.append(post)
        else:
            last = stages.pop()
            stages.append(lambda stt, last=last: post(last(stt)))
        pipe = PipelinedRunner(*stages, images)
        for j in range(pipe.n):
            pipe.static_input(j).copy_(images)  # every buffer set holds the resident batch: no per-step copy

        def step():
            j = pipe.submit()
            boxes, count = pipe.outputs(j)
            if gather is not None:
                with torch.cuda.stream(pipe.sp):  # the gather follows this batch's NMS on the post-processing stream
                    gather(boxes, count)
            return boxes, count

        def drain():
            pipe.wait()

    for _ in range(a.warmup):
        step()
    drain()
    if gather is not None:
        with torch.cuda.stream(pipe.sp if not a.no_pipeline else torch.cuda.current_stream()):
            gather.flush()
        gather.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        boxes, count = step()
    drain()
    if gather is not None:  # the last (possibly partial) block of rows is exchanged inside the timed region
        with torch.cuda.stream(pipe.sp if not a.no_pipeline else torch.cuda.current_stream()):
            gather.flush()
        gather.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        ms = dt / a.steps * 1e3
        total_images = a.batch * world * a.steps
        out = {
            "metric": f"images/sec @ {a.imgsz}x{a.imgsz} {'fp16' if a.dtype == 'f16' else 'fp32'} (EdgeLine-YOLO detection forward path: backbone + DWT neck + GFLv2 head decode + batched NMS)",
            "value": round(total_images / dt, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
            "data": "synthetic",
            "config": {"workload": f"{a.model} (EdgeLine-YOLO scale n, nc=80) predict path, {a.imgsz}x{a.imgsz}, batch {a.batch}/GPU, "
                                   f"conf {conf} iou {iou} max_det {max_det}, random-init weights (dense NMS regime: mean detections/img "
                                   f"{float(count.float().mean()):.0f})",
                       "global_batch": a.batch * world, "imgsz": a.imgsz,
                       "pipeline": "single graph per batch" if a.no_pipeline else f"{len(cuts) + 1}-stage software pipeline over consecutive batches (layer cuts {cuts}; last stage = head + decode + NMS), one hipGraph and one HIP stream per stage", "sharding": f"images x{world}" + (", RCCL all_gather of boxes" if world > 1 else "")},
        }
    if rank == 0 and not a.no_roofline:
        from edge_yolo_amd import profiling
        torch.cuda.synchronize()
        out["roofline"] = profiling.dominant_kernel_roofline(lambda: device_step(images), steps=min(a.steps, 10), hbm_peak_gbs=HBM_PEAK_GBS,
                                                             mfma_peak_tflops=MFMA_F16_PEAK_TFLOPS)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.model, sd, a.imgsz, a.cpu_images, conf, iou)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if gather is not None and rank == 0:
        rows = gather.results()
        assert len(rows) % (a.batch * world) == 0 and len(rows) > 0 and all(r.shape[1] == 6 for r in rows)
    if world > 1 or force_gather:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
