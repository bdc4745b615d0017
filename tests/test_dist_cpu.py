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
