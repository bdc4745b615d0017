"""-m gpu: the RCCL leg of the multi-GPU path (BASELINE config C4) on the ONE GPU of the test box.

A world-size-1 `nccl` process group is a real RCCL communicator: `all_gather_into_tensor` runs RCCL's kernel on the collective
backend's stream, `BoxGatherer`'s CUDA branch (own_stream=False, `every` batches per collective, the partial final block and its
flush) is the code an 8-GPU run executes, and the collective backend's stream is the extra active stream the 3-vs-4-stage
decision of bench.py is about.  What these tests cannot show is the xGMI transfer itself (one rank = no peer).

  * rows gathered behind the real 3- and 4-stage bench pipeline == direct execution of the same batches, in global order;
  * bench.py's control flow (`timed_steps`, `choose_variant`) with the gatherer next to a live process group;
  * `python bench.py --force-gather` end to end as a child process (the launcher path the driver uses at N=1).
"""
import contextlib
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def nccl_group():
    """world-size-1 nccl group on cuda:0 (one per module: RCCL communicators are expensive to create)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29000 + os.getpid() % 2000), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        dist.destroy_process_group()
        for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            os.environ.pop(k, None)


def _direct(model, xs, conf=0.25, iou=0.7, max_det=300):
    from edge_yolo_amd.utils import ops
    out = []
    for x in xs:
        b, c, _ = ops.nms_device(model(x)[0], conf, iou, max_det=max_det)
        out.append((b.clone(), c.clone()))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("nstages", [3, 4])
def test_box_gatherer_nccl_behind_the_bench_pipeline(nccl_group, nstages):
    """7 different batches through bench.make_pipeline (3 / 4 stages, bench.pipeline_cuts) with the gatherer on the post-processing
    stream, every=3 -> blocks of 3, 3 and a flushed partial 1: every block's rows equal direct execution, in submission order."""
    import bench
    from edge_yolo_amd import dist as eyd
    dev = nccl_group
    B, H, W, md = 4, 256, 320, 300
    model, _ = bench.build_model("yolo11n-test.yaml", torch.float16, dev)
    model.model[-1].head_streams = False
    g = torch.Generator(device=dev).manual_seed(11)
    xs = [torch.rand(B, 3, H, W, generator=g, device=dev).half() for _ in range(7)]
    want = _direct(model, xs)
    head_nms = {"conf": 0.25, "classes": None}
    pipe = bench.make_pipeline(model, xs[0], bench.pipeline_cuts(len(model.model), nstages), head_nms, 0.25, 0.7, md)
    assert pipe.n == nstages
    gather = eyd.BoxGatherer(1, B, md, dev, own_stream=False, every=3)
    assert gather.cuda and gather.side is None and gather.every == 3
    blocks, k = [], [0]

    def step():
        j = pipe.submit(xs[k[0]])
        k[0] += 1
        return pipe.outputs(j)

    dt, _ = bench.timed_steps(step, pipe.wait, gather, len(xs), 0, (lambda: torch.cuda.stream(pipe.sp)), torch.cuda.synchronize, (lambda: None),
                              (lambda v: v), on_block=lambda i: blocks.append(i))
    assert blocks == [0, 1, 2] and gather.nslots[0] == 1  # block 2 (buffer 0) is the flushed partial one
    # blocks 1 and 2 are still in the double buffer; block 0 was overwritten by block 2: re-run to read every block as it completes
    rows_1, rows_2 = gather.results(1), gather.results(2)
    assert len(rows_1) == 3 * B and len(rows_2) == 1 * B
    for n, rows in ((3, rows_1), (6, rows_2)):
        for i, r in enumerate(rows):
            wb, wc = want[n + i // B]
            cnt = int(wc[i % B])
            assert r.shape == (cnt, 6)
            assert torch.equal(r, wb[i % B, :cnt]), f"block starting at batch {n}, row {i}"
    assert dt > 0


def test_box_gatherer_nccl_blocks_read_as_they_complete(nccl_group):
    """Same path, rows consumed inside on_block (what a serving loop does): all 8 batches, every=3 + partial flush of 2, each equal to
    direct execution; then the own_stream=True form (one collective per batch on the gatherer's side stream)."""
    import bench
    from edge_yolo_amd import dist as eyd
    dev = nccl_group
    B, H, W, md = 3, 192, 192, 300
    model, _ = bench.build_model("yolo11n-test.yaml", torch.float16, dev)
    model.model[-1].head_streams = False
    g = torch.Generator(device=dev).manual_seed(12)
    xs = [torch.rand(B, 3, H, W, generator=g, device=dev).half() for _ in range(8)]
    want = _direct(model, xs)
    pipe = bench.make_pipeline(model, xs[0], bench.pipeline_cuts(len(model.model), 4), {"conf": 0.25, "classes": None}, 0.25, 0.7, md)
    gather = eyd.BoxGatherer(1, B, md, dev, own_stream=False, every=3)
    got, k = [], [0]

    def step():
        j = pipe.submit(xs[k[0]])
        k[0] += 1
        return pipe.outputs(j)

    bench.timed_steps(step, pipe.wait, gather, len(xs), 0, (lambda: torch.cuda.stream(pipe.sp)), torch.cuda.synchronize, (lambda: None), (lambda v: v),
                      on_block=lambda i: got.extend(r.clone() for r in gather.results(i)))
    assert len(got) == len(xs) * B
    for i, r in enumerate(got):
        wb, wc = want[i // B]
        cnt = int(wc[i % B])
        assert r.shape == (cnt, 6) and torch.equal(r, wb[i % B, :cnt]), f"row {i}"
    # side-stream form (bench.py --no-pipeline): every batch its own collective, double buffered
    side = eyd.BoxGatherer(1, B, md, dev, own_stream=True)
    assert side.side is not None and side.every == 1
    for n, x in enumerate(xs[:3]):
        b, c = want[n]
        blk = side(b, c)
        rows = side.results(blk)
        assert len(rows) == B and all(torch.equal(r, b[i, : int(c[i])]) for i, r in enumerate(rows))


def test_stage_count_autotune_next_to_a_process_group(nccl_group):
    """bench.choose_variant over real 3- and 4-stage pipelines sharing one stream set, with the nccl gatherer active (the situation the
    start-up auto-tune exists for): both variants run, one is kept, and it still returns the boxes of direct execution."""
    import bench
    from edge_yolo_amd import dist as eyd
    dev = nccl_group
    B, md = 4, 300
    model, _ = bench.build_model("yolo11n-test.yaml", torch.float16, dev)
    model.model[-1].head_streams = False
    x = torch.rand(B, 3, 256, 256, generator=torch.Generator(device=dev).manual_seed(13), device=dev).half()
    want = _direct(model, [x])[0]
    gather = eyd.BoxGatherer(1, B, md, dev, own_stream=False, every=2)
    streams = [torch.cuda.Stream(device=dev) for _ in range(5)]
    n = len(model.model)
    head_nms = {"conf": 0.25, "classes": None}

    def trial(p):
        return bench.timed_steps(lambda: p.outputs(p.submit()), p.wait, gather, 6, 2, (lambda: torch.cuda.stream(p.sp)), torch.cuda.synchronize,
                                 (lambda: None), (lambda v: v))[0]

    name, pipe, times = bench.choose_variant({str(s): (lambda s=s: bench.make_pipeline(model, x, bench.pipeline_cuts(n, s), head_nms, 0.25, 0.7, md, streams=streams))
                                              for s in (3, 4)}, trial, lambda v: v)
    assert set(times) == {"3", "4"} and name in times and all(t > 0 for t in times.values())
    j = pipe.submit()
    pipe.wait(j)
    torch.cuda.synchronize()
    b, c = pipe.outputs(j)
    assert torch.equal(c, want[1]) and torch.equal(b, want[0])
    rows = gather.results()
    assert len(rows) % B == 0 and len(rows) > 0


def test_bench_force_gather_child_process():
    """`python bench.py --force-gather` as the driver would start it (a child process, N=1): the RCCL gather path inside the timed
    region, one JSON line out."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MASTER_PORT"] = str(31000 + os.getpid() % 2000)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-gather", "--steps", "10", "--warmup", "3", "--gather-every", "4", "--batch", "8",
                        "--imgsz", "320", "--no-roofline", "--no-cpu-baseline", "--no-api"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["steps"] == 10
