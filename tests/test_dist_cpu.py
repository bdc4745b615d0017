"""world_size-2 gloo test (CPU) of the multi-GPU path: image sharding + the one all_gather of packed result rows."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import edge_yolo_amd  # noqa: F401
from edge_yolo_amd import dist as eyd


def _fake_results(b0, b1, max_det):
    """deterministic per-image rows: image i has (i % max_det) + 1 detections."""
    B = b1 - b0
    boxes = torch.zeros(B, max_det, 6)
    count = torch.zeros(B, dtype=torch.int32)
    for j, i in enumerate(range(b0, b1)):
        n = (i * 7) % max_det + 1
        boxes[j, :n] = torch.arange(n * 6, dtype=torch.float32).view(n, 6) + 1000 * i
        count[j] = n
    return boxes, count


def _worker(rank, world, port, n_images, max_det, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = eyd.shard_range(n_images, world, rank)
        g = eyd.BoxGatherer(world, hi - lo, max_det, "cpu")
        out = None
        for step in range(3):  # exercises the double buffering
            boxes, count = _fake_results(lo, hi, max_det)
            boxes += step
            out = g(boxes, count)
        rows = g.results()
        q.put((rank, [r.clone() for r in rows]))
    finally:
        dist.destroy_process_group()


def test_gather_boxes_gloo_world2():
    world, n_images, max_det = 2, 8, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, max_det, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full_b, full_c = _fake_results(0, n_images, max_det)
    for rank in range(world):
        rows = got[rank]
        assert len(rows) == n_images  # every rank ends up with every image's rows, in image order
        for i, r in enumerate(rows):
            n = int(full_c[i])
            assert r.shape == (n, 6)
            assert torch.equal(r, full_b[i, :n] + 2)


def test_shard_range_covers_everything():
    for n in (1, 7, 32, 256):
        for w in (1, 2, 3, 8):
            spans = [eyd.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_pack_unpack_roundtrip():
    b, c = _fake_results(0, 5, 12)
    rows = eyd.unpack_rows(eyd.pack_rows(b, c), 12)
    for i, r in enumerate(rows):
        assert torch.equal(r, b[i, : int(c[i])])


# ---------------------------------------------------------------------------------------------------------------------------------
# bench.py's real control flow (timed_steps: warm-up, barrier-bracketed timed region, gather of every step with `every` batches per
# collective, flush of the partial final block inside the timed region; choose_variant: the start-up auto-tune) with a stub step.
def _bench_worker(rank, world, port, steps, warmup, every, B, max_det, q):
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import contextlib
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def reduce_max(v):
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t)

        # start-up auto-tune: rank 1 sees variant "a" slower than "b", rank 0 the other way round; the max-reduced times decide, so
        # both ranks must pick the same one
        delays = {"a": (0.002, 0.030)[rank], "b": (0.012, 0.010)[rank]}
        name, obj, times = bench.choose_variant({k: (lambda k=k: k) for k in ("a", "b")}, lambda k: (time.sleep(delays[k]), delays[k])[1], reduce_max)
        gather = eyd.BoxGatherer(world, B, max_det, "cpu", own_stream=False, every=every)
        blocks, n = [], [0]

        def step():  # global step counter -> this rank's images of that step
            k = n[0]
            n[0] += 1
            lo = (k * world + rank) * B
            return _fake_results(lo, lo + B, max_det)

        dt, _ = bench.timed_steps(step, lambda: None, gather, steps, warmup, contextlib.nullcontext, lambda: None, dist.barrier, reduce_max,
                                  on_block=lambda k: blocks.append([r.numpy().copy() for r in gather.results(k)]))  # numpy: plain pickles through the queue
        q.put((rank, name, obj, blocks, dt))
    finally:
        dist.destroy_process_group()


def test_bench_control_flow_gloo_world2():
    world, steps, warmup, every, B, max_det = 2, 7, 2, 3, 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, steps, warmup, every, B, max_det, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=180)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] == "b" and got[0][1] == "b"  # same winner on both ranks (max over ranks: a = 30 ms, b = 12 ms)
    total = warmup + steps
    full_b, full_c = _fake_results(0, total * world * B, max_det)
    for rank in range(world):
        blocks = got[rank][2]
        # warm-up: 2 steps -> one partial block (flushed before the timed region); timed: 7 steps -> blocks of 3, 3 and a partial 1
        assert [len(b) for b in blocks] == [2 * world * B, 3 * world * B, 3 * world * B, 1 * world * B]
        rows = [r for b in blocks for r in b]
        assert len(rows) == total * world * B  # every rank ends with all B*world*steps rows (+ warm-up), in global image order
        for i, r in enumerate(rows):
            assert torch.equal(torch.from_numpy(r), full_b[i, : int(full_c[i])]), f"rank {rank} image {i}"
        assert got[rank][3] > 0


# ---------------------------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` as typed (no torch.distributed.run around it): the parent starts its own rank processes, relays rank 0's
# JSON line and exits with the ranks' worst code.  --rehearse-cpu = gloo + stub step (no device work), so this runs without a GPU.
def _run_bench(*args, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=300)


def test_bench_self_launch_spawns_ranks_and_relays_one_json_line():
    import json
    r = _run_bench("--gpus", "2", "--rehearse-cpu", "--steps", "5", "--warmup", "2", "--batch", "3", "--gather-every", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # exactly ONE line on stdout, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2
    assert out["rows_checked"] == (5 + 2) * 2 * 3  # every image of every rank and step reached rank 0, in global order (asserted in the rank)
    assert "REHEARSAL" in out["metric"]  # the stub can never be mistaken for a measurement


def test_bench_self_launch_reports_rank_failure():
    """Without GPUs the real path must fail in the ranks (no CPU fallback) -- and the launcher must turn that into a non-zero exit
    code and no JSON line, not into a hang or a silent success."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a host without GPUs")
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-roofline", "--no-cpu-baseline", "--no-api")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "rank exit codes" in r.stderr


def test_bench_rejects_mismatched_world_size():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-cpu"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
