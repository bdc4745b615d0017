"""Predict loop for the detect task: preprocess -> inference -> postprocess, mirroring the reference's
engine/predictor.py (`BasePredictor.preprocess/inference/postprocess/stream_inference` :116-304) and
models/yolo/detect/predict.py (`DetectionPredictor.postprocess` :23-41).

MI355X-native execution model: the whole per-batch device work (stem ... head decode, NMS) is launched on one HIP
stream and, for a fixed (batch, H, W, dtype), captured ONCE into a hipGraph and replayed (the reference re-dispatches
~300 ATen ops from Python per batch).  NMS results come back as one fixed-size (B,max_det,6)+(B,) device buffer,
so there is exactly one D2H copy per batch.
"""
import contextlib
import gc
import threading
import time

import numpy as np
import torch

from .results import Results
from ..utils import ops


class Profile:
    """Wall timer with device sync (reference utils/ops.py:17-62)."""

    def __init__(self, device=None):
        self.t, self.dt, self.device = 0.0, 0.0, device

    def __enter__(self):
        self.start = self.time()
        return self

    def __exit__(self, *a):
        self.dt = self.time() - self.start
        self.t += self.dt

    def time(self):
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return time.perf_counter()


@contextlib.contextmanager
def _capture(graph):
    """torch.cuda.graph(graph) with Python's cyclic garbage collector held off for the duration: a collection that runs in the middle
    of a capture may finalise an unrelated captured graph (a replaced predictor, an exhausted pipeline), and destroying a hipGraph while
    any stream is capturing is an error ("operation not permitted when stream is capturing") that kills the process."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was:
            gc.enable()


class GraphRunner:
    """Captures `fn(static_input) -> tuple of tensors` into a HIP graph per input signature and replays it."""

    def __init__(self, fn, warmup=2):
        self.fn, self.warmup, self.graphs = fn, warmup, {}

    def __call__(self, x):
        key = (tuple(x.shape), x.dtype, x.device)
        g = self.graphs.get(key)
        if g is None:
            static_in = x.clone()
            s = torch.cuda.Stream(device=x.device)
            s.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(s):
                for _ in range(self.warmup):  # packs weights, sizes allocator pools
                    out = self.fn(static_in)
            torch.cuda.current_stream(x.device).wait_stream(s)
            torch.cuda.synchronize(x.device)
            graph = torch.cuda.CUDAGraph()
            with _capture(graph):
                out = self.fn(static_in)
            g = self.graphs[key] = (graph, static_in, out)
        graph, static_in, out = g
        if x.data_ptr() != static_in.data_ptr():  # callers that fill static_input() in place skip the copy
            static_in.copy_(x, non_blocking=True)
        graph.replay()
        return out

    def static_input(self, like):
        """The captured graph's input buffer for tensors shaped like `like` (captures on first use)."""
        self(like)
        return self.graphs[(tuple(like.shape), like.dtype, like.device)][1]


class PipelinedRunner:
    """Software pipeline over consecutive batches: stage s of batch i runs on its own HIP stream while stage s-1 of batch i+1 and
    stage s+1 of batch i-1 run on theirs.  Most kernels of this network at batch 32 are latency-bound (small feature maps, a few
    hundred workgroups), so independent work from neighbouring batches fills the chip.  Each stage is a captured hipGraph; as many
    buffer sets as stages alternate, so nothing is copied between the stages.  Every batch still goes through every stage in full --
    only their placement in time overlaps (throughput mode; stages-1 batches of extra latency).

    PipelinedRunner(f1, f2, ..., example): stage functions chained (stage k gets stage k-1's return value), example = input batch.
    """

    def __init__(self, *args, warmup=2, copy_stream=False, streams=None, extra_sets=None):
        """copy_stream=True: batches handed to submit(x) are copied into the stage-0 input buffer on a stream of their own (host
        tensors: one H2D DMA per batch that overlaps the compute of the batches in flight) instead of on the first stage's stream.
        A pinned host batch must stay unchanged until its results are out (the usual contract of an asynchronous copy).
        streams: HIP streams to run the stages on (>= one per stage).  HIP binds streams to the 4 hardware queues in creation order,
        so callers that build several pipelines (bench.py's start-up auto-tune) create ONE set of streams and hand it to all of
        them: a second set would alias queues of the first and run ~25 % slower."""
        *stages, example = args
        dev = example.device
        self.dev, self.n = dev, len(stages)
        # buffer sets: one per stage -- plus one with a copy stream: the upload of batch i is a pipeline stage of its own (it waits for
        # the batch that used its buffer set before), so with n sets stage 0 idles for the whole transfer of every batch
        self.nsets = self.n + ((1 if copy_stream else 0) if extra_sets is None else int(extra_sets))
        if streams is not None and len(streams) < len(stages):
            raise ValueError(f"PipelinedRunner: {len(stages)} stages need {len(stages)} streams, got {len(streams)}")
        self.streams = list(streams[: len(stages)]) if streams is not None else [torch.cuda.Stream(device=dev) for _ in stages]
        # the upload rides on the FIRST stage's stream (in order before that batch's stage-0 graph; it overlaps the other stages of the
        # batches in flight).  A stream of its own would be the fifth active stream: HIP binds streams to 4 hardware queues, two of the
        # five then share one and the batches stop overlapping altogether (measured: 2.05 ms/batch = the unpipelined step, device-resident
        # input included)
        self.copy_stream = self.streams[0] if copy_stream else None
        self.sf, self.sp = self.streams[0], self.streams[-1]
        self.sets = []
        cur = torch.cuda.current_stream(dev)
        for _ in range(self.nsets):
            static_in = example.clone()
            self.sf.wait_stream(cur)
            with torch.cuda.stream(self.sf):
                for _ in range(warmup):
                    v = static_in
                    for f in stages:
                        v = f(v)
            cur.wait_stream(self.sf)
            torch.cuda.synchronize(dev)
            graphs, v = [], static_in
            for f in stages:
                g = torch.cuda.CUDAGraph()
                with _capture(g):
                    v = f(v)
                graphs.append(g)
            self.sets.append(dict(x=static_in, graphs=graphs, out=v, done=[torch.cuda.Event() for _ in stages]))
        self.i = 0
        for st in self.sets:  # events start in the signalled state
            for e in st["done"]:
                e.record(cur)
        for st in self.sets:
            st["post_done"] = st["done"][-1]

    def static_input(self, j=None):
        return self.sets[self.i % self.nsets if j is None else j]["x"]

    def submit(self, x=None, upload=None, side_copy=False):
        """Enqueue one batch (x=None: the batch already sits in static_input()).  Returns the buffer-set index; its outputs
        are valid after `wait(j)` / a device synchronise.  upload(dst, x, j): brings x to the device in place of the plain copy into the
        stage-0 input buffer `dst` of buffer set j; runs on the copy stream when there is one and may return a callable that is then run
        on the FIRST STAGE's stream right before its graph (e.g. raw uint8 bytes over PCIe on the copy stream, the conversion kernel
        on the stage stream).  side_copy=True: the plain copy goes to a stream of its own instead of the first stage's -- right for
        DMA uploads of big host batches (measured, 78.6 MB f16 batches: 2.2 vs 2.8 ms per batch), wrong for everything else: a fifth
        active stream shares one of HIP's 4 hardware queues with a stage and the batches stop overlapping (device-resident input:
        2.05 vs 1.35 ms per batch)."""
        j = self.i % self.nsets
        self.i += 1
        st = self.sets[j]
        cur = torch.cuda.current_stream(self.dev)
        self.streams[0].wait_stream(cur)
        copied, finish = None, None
        if self.copy_stream is not None and x is not None and x.data_ptr() != st["x"].data_ptr():
            cs = self.copy_stream
            if side_copy:
                if getattr(self, "_side_copy", None) is None:
                    self._side_copy = torch.cuda.Stream(device=self.dev)
                cs = self._side_copy
            cs.wait_stream(cur)
            cs.wait_event(st["done"][-1])  # every stage of the batch that used this buffer set n submits ago is done (stage 0 read its input long before)
            with torch.cuda.stream(cs):
                if upload is not None:
                    finish = upload(st["x"], x, j)
                else:
                    st["x"].copy_(x, non_blocking=True)
                if x.is_cuda:
                    # x was allocated on the caller's stream and is read here on the copy stream: without this the caching allocator hands its
                    # block to the caller's next allocation (e.g. the next batch's converted input) as soon as the caller drops x, while
                    # this copy may not have run yet -- the batch would silently receive the next batch's pixels
                    x.record_stream(cs)
                copied = torch.cuda.Event()
                copied.record(cs)
            x = None
        for s, (stream, g) in enumerate(zip(self.streams, st["graphs"])):
            with torch.cuda.stream(stream):
                # stage 0 waits for the LAST stage of the batch that used this buffer set n submits ago; stage s for stage s-1 of this batch
                stream.wait_event(st["done"][-1] if s == 0 else st["done"][s - 1])
                if s == 0 and copied is not None:
                    stream.wait_event(copied)
                    if finish is not None:
                        finish()
                if s == 0 and x is not None and x.data_ptr() != st["x"].data_ptr():
                    if upload is not None:
                        finish = upload(st["x"], x, j)
                        if finish is not None:
                            finish()
                    else:
                        st["x"].copy_(x, non_blocking=True)
                g.replay()
                st["done"][s].record(stream)
        return j

    def outputs(self, j):
        return self.sets[j]["out"]

    def wait(self, j=None):
        cur = torch.cuda.current_stream(self.dev)
        for k in ([j] if j is not None else range(self.nsets)):
            cur.wait_event(self.sets[k]["done"][-1])


class DetectionPredictor:
    def __init__(self, model, device, half=False, conf=0.25, iou=0.7, max_det=300, agnostic_nms=False, classes=None, graph=True, validate_input=None, keep_pred=False,
                 augment=False):
        """This signature is the contract for classes injected through `YOLO.predict(predictor=...)` (reference engine/model.py:505,552):
        the facade constructs `cls(model, device, half=, conf=, iou=, max_det=, agnostic_nms=, classes=, graph=, augment=)` and calls the
        instance with the source; subclasses usually override `preprocess` / `postprocess` (reference BasePredictor hooks).

        validate_input: the reference's value-range check of tensor sources (LoadTensor._single_check, data/loaders.py:560-566: a tensor
        whose max exceeds 1 is divided by 255 with a warning).  None (default) = reference behaviour for HOST tensors (the check runs on
        the host, no device sync) and NO check for DEVICE tensors (it would cost a device reduction + a host sync per call -- a
        deliberate divergence: a 0..255 device tensor is fed to the network as is); True = check device tensors too; False = never.
        augment: scale / flip test-time augmentation (DetectionModel._predict_augment; cfg/default.yaml `augment`)."""
        self.model, self.device, self.half, self.validate_input, self.keep_pred, self.augment = model, device, half, validate_input, keep_pred, bool(augment)
        self.conf, self.iou, self.max_det, self.agnostic_nms, self.classes = conf, iou, max_det, agnostic_nms, classes
        self._lock = threading.Lock()
        self.runner = GraphRunner(self._device_step) if graph else self._device_step

    def close(self):
        """Drop the captured graphs now (the runner and this predictor reference each other: without this they live until the cyclic
        collector runs, possibly in the middle of somebody's capture)."""
        self.runner = None

    # ---- device side (captured)
    def _device_step(self, im):
        if self.augment:  # three forwards + descale / clip / concat (reference tasks.py:372-408), then the ordinary NMS on the concatenated anchors
            pred = self.model(im, augment=True)[0]
            boxes, count, index = ops.nms_device(pred, self.conf, self.iou, self.classes, self.agnostic_nms, self.max_det)
            return boxes, count, index, pred
        # the head decode builds the NMS candidates for (conf, classes) in the same pass: the (B,4+nc,A) prediction tensor is only
        # written when the caller asked for it (keep_pred)
        preds = self.model(im, head_nms={"conf": self.conf, "classes": self.classes, "keep_pred": self.keep_pred})
        pred = preds[0] if isinstance(preds, (list, tuple)) else preds
        boxes, count, index = ops.nms_device(pred, self.conf, self.iou, self.classes, self.agnostic_nms, self.max_det)
        return boxes, count, index, getattr(pred, "pred", pred)

    def preprocess(self, im):
        """Tensor sources: BCHW float in [0,1] (reference LoadTensor, data/loaders.py:516-586).  ndarray / list sources:
        HWC uint8 BGR images, letterboxed to a common stride-32 shape on the host (reference predictor.py:116-161)."""
        if not isinstance(im, torch.Tensor):
            return self._letterbox_batch(im, getattr(self, "imgsz", 640))
        if im.dim() == 3:
            im = im[None]
        if im.dim() != 4 or im.shape[2] % 32 or im.shape[3] % 32:
            raise ValueError(f"input tensor should be BCHW with H,W multiples of 32, got {tuple(im.shape)}")
        if not im.is_floating_point():
            raise TypeError("tensor sources must be floating point images in [0,1] (uint8 HWC BGR images go in as numpy arrays)")
        check = (not im.is_cuda) if self.validate_input is None else bool(self.validate_input)
        if check and im.numel():
            mx = float(im.max())
            if mx > 1.0 + float(torch.finfo(im.dtype).eps):  # reference LoadTensor._single_check (data/loaders.py:560-566): warn and rescale
                import warnings
                warnings.warn(f"torch.Tensor inputs should be normalized 0.0-1.0 but max value is {mx}. Dividing input by 255.")
                im = im.float() / 255.0
        im = im.to(self.device, non_blocking=True)
        return (im.half() if self.half else im.float()).contiguous()

    def _letterbox_batch(self, ims, new_shape=640):
        """reference pre_transform + preprocess (predictor.py:123-161) on the GPU: LetterBox geometry on the host, one
        `ey_letterbox` launch per image (resize + pad 114 + BGR->RGB + CHW + /255) into the batch tensor."""
        from ..data.augment import LetterBox
        ims = ims if isinstance(ims, (list, tuple)) else [ims]
        self._orig = [np.asarray(a) for a in ims]
        same_shapes = len({a.shape for a in self._orig}) == 1
        lb = LetterBox(new_shape, auto=same_shapes, stride=int(max(self.model.stride)) if hasattr(self.model, "stride") else 32)
        return lb.batch(self._orig, self.device, torch.float16 if self.half else torch.float32)

    def __call__(self, source):
        with self._lock, torch.cuda.device(self.device):  # kernels go to the CURRENT device's stream (_lib.stream)
            self._orig = None
            prof = (Profile(self.device), Profile(self.device), Profile(self.device))
            with prof[0]:
                im = self.preprocess(source)
            with prof[1]:
                boxes, count, index, pred = self.runner(im)
            with prof[2]:
                results = self.postprocess(boxes, count, im, source)
            n = len(results)
            for r in results:
                r.speed = {"preprocess": prof[0].dt * 1e3 / n, "inference": prof[1].dt * 1e3 / n, "postprocess": prof[2].dt * 1e3 / n}
            return results

    def postprocess(self, boxes, count, img, source):
        """reference detect/predict.py:23-41: per image rows -> scale_boxes to the original shape -> Results."""
        n = count.tolist()
        boxes = boxes.clone()
        results = []
        names = self.model.names
        # original image == network input (tensor sources; images that already have the network's shape: scale_boxes is gain 1, pad 0):
        # one clip over the whole batch (4 launches, not 4 per image)
        same = self._orig is None or all(tuple(o.shape[:2]) == tuple(img.shape[2:]) for o in self._orig)
        if same:
            ops.clip_boxes(boxes, img.shape[2:])
        for i in range(len(n)):
            det = boxes[i, : n[i]]
            orig = self._orig[i] if self._orig is not None else None
            if not same:
                det[:, :4] = ops.scale_boxes(img.shape[2:], det[:, :4], orig.shape)
            r = Results(orig, path=f"image{i}.jpg", names=names, boxes=det)
            if orig is None:
                r.orig_shape = tuple(img.shape[2:])
                r.boxes.orig_shape = r.orig_shape
            results.append(r)
        return results
