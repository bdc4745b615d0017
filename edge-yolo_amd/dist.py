"""Multi-GPU sharding of the predict path: images are independent units, so each rank (one process per GPU) runs the
whole path on its own slice of the batch; the only exchange is ONE all_gather per step of the padded result rows
(RCCL over xGMI when the process group is 'nccl'; the same code runs on 'gloo' for the CPU tests).

Message: (B_local, max_det*6 + 1) fp32 per rank — rows x1,y1,x2,y2,conf,cls plus the count in the last column —
7.2 KB per image, i.e. 230 KB per rank at 32 images: latency bound, far below one 153 GB/s xGMI link, so a single
flat all_gather on a side stream (overlapping the next batch) is the right shape; no ring/tree tuning applies.
The reference has no multi-GPU predict at all (engine/predictor.py:306-321 is single device).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, world, rank):
    """Contiguous slice [lo, hi) of `n_items` images owned by `rank` (remainder spread over the first ranks)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_rows(boxes, count, out=None):
    """(B,max_det,6) fp32 + (B,) int -> (B, max_det*6+1) fp32."""
    B, md, _ = boxes.shape
    if out is None:
        out = torch.empty((B, md * 6 + 1), dtype=torch.float32, device=boxes.device)
    out[:, : md * 6].copy_(boxes.reshape(B, md * 6))
    out[:, md * 6].copy_(count)
    return out


def unpack_rows(packed, max_det):
    """(N, max_det*6+1) -> list of (n_i, 6) tensors."""
    n = packed[:, max_det * 6].round().to(torch.int64).tolist()
    rows = packed[:, : max_det * 6].reshape(-1, max_det, 6)
    return [rows[i, : n[i]] for i in range(len(n))]


class BoxGatherer:
    """All-gather of packed result rows, K = `every` batches per collective, double buffered.

    own_stream=True (every forced to 1): pack + collective run on a stream of the gatherer's own; the producer stream only waits for
    the staging copy, so the collective overlaps the next batch.  own_stream=False: pack + collective are enqueued on the caller's
    current stream (for callers that already run the post-processing on a stream of its own: every extra ACTIVE stream costs
    hardware-queue sharing with the forward graphs), and the rows of K consecutive batches are collected on the device and exchanged
    by ONE all_gather (K x 230 KB per rank: still latency-bound); `flush()` sends a partial block.

    Block layout on the wire: rank-major [rank][slot][image].  `results(j)` returns the rows of block j in GLOBAL batch order
    (slot, rank, image) -- i.e. for every batch of the block the images of rank 0, then rank 1, ... -- and only for the slots that
    were filled: unused slots of a partial block carry count 0 and are dropped."""

    def __init__(self, world, batch_local, max_det, device, group=None, own_stream=True, every=1):
        self.world, self.B, self.max_det, self.group = world, batch_local, max_det, group
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if (self.cuda and own_stream) else None
        self.every = 1 if self.side is not None else max(1, int(every))
        self.slot = 0
        self.w = max_det * 6 + 1
        rows = batch_local * self.every
        self.stage = [torch.zeros((rows, self.w), dtype=torch.float32, device=self.device) for _ in range(2)]
        self.gathered = [torch.zeros((world * rows, self.w), dtype=torch.float32, device=self.device) for _ in range(2)]
        self.nslots = [0, 0]  # filled slots of the block each buffer holds
        self.i = 0            # blocks exchanged so far; block k lives in buffer k & 1
        self.copied = [None, None]
        self.done = [None, None]

    def _exchange(self, j):
        if self.cuda:
            dist.all_gather_into_tensor(self.gathered[j], self.stage[j], group=self.group)
        else:
            dist.all_gather(list(self.gathered[j].chunk(self.world)), self.stage[j], group=self.group)

    def __call__(self, boxes, count):
        """Add one batch.  Returns the index of the block that was exchanged by this call, or None while the block is still filling."""
        j = self.i & 1
        if self.side is not None:
            main = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                pack_rows(boxes, count, self.stage[j])
                copied = torch.cuda.Event()
                copied.record(self.side)
                self._exchange(j)
                done = torch.cuda.Event()
                done.record(self.side)
            main.wait_event(copied)  # the next replay may overwrite boxes/count only after they were staged
            self.copied[j], self.done[j] = copied, done
            self.nslots[j] = 1
            self.i += 1
            return self.i - 1
        pack_rows(boxes, count, self.stage[j][self.slot * self.B:(self.slot + 1) * self.B])
        self.slot += 1
        if self.slot == self.every:
            return self.flush()
        return None

    def flush(self):
        """(own_stream=False) exchange the block collected so far, complete or not.  Returns its index, or None if it was empty."""
        if self.side is not None or self.slot == 0:
            return None
        j = self.i & 1
        if self.slot < self.every:
            self.stage[j][self.slot * self.B:, self.max_det * 6].zero_()  # unused slots: count 0 (their rows are stale, never unpacked)
        self.nslots[j] = self.slot
        self.slot = 0
        self._exchange(j)
        if self.cuda:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            self.done[j] = done
        self.i += 1
        return self.i - 1

    def wait(self):
        if self.cuda and self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        elif self.cuda:
            for d in self.done:
                if d is not None:
                    torch.cuda.current_stream(self.device).wait_event(d)

    def results(self, block=None):
        """Rows of an exchanged block (default: the newest), as a list of (n_i, 6) tensors in global batch order (slot, rank, image)."""
        if self.i == 0:
            return []
        k = self.i - 1 if block is None else block
        j = k & 1
        self.wait()
        if self.cuda:
            torch.cuda.current_stream(self.device).synchronize()
        g = self.gathered[j].view(self.world, self.every, self.B, self.w)[:, : self.nslots[j]]
        return unpack_rows(g.permute(1, 0, 2, 3).reshape(-1, self.w), self.max_det)
