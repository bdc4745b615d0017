"""`YOLO(...)` facade for the detect task, mirroring the reference's engine/model.py (`Model.__init__` :84-151,
`_new` :231-264, `_load` :266-302, `predict` :501-560) and models/yolo/model.py (:11-59)."""
from pathlib import Path

import torch

from .predictor import DetectionPredictor
from ..nn.tasks import DetectionModel, guess_model_task, torch_safe_load_state, yaml_model_load


def _plain(o):
    """YAML dict -> containers torch.load(weights_only=True) accepts (dict / list / str / int / float / bool / None)."""
    if isinstance(o, dict):
        return {str(k): _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    if isinstance(o, (str, int, float, bool)) or o is None:
        return o
    return str(o)


class Model(torch.nn.Module):
    def __init__(self, model="yolo11n.yaml", task=None, verbose=False, nc=None):
        """nc (extension): number of classes for a model built from a YAML (the reference takes it from the dataset YAML at train time,
        engine/trainer.py; checkpoints written by save() carry it)."""
        super().__init__()
        self.predictor = None
        self.model = None
        self.overrides = {}
        self.ckpt_path = None
        self.task = task
        if isinstance(model, dict):  # (extension) an already loaded YAML dict
            self._new(model, task=task, verbose=verbose, nc=nc)
            return
        model = str(model).strip()
        if Path(model).suffix in {".yaml", ".yml"}:
            self._new(model, task=task, verbose=verbose, nc=nc)
        else:
            self._load(model, task=task)

    def _new(self, cfg, task=None, model=None, verbose=False, nc=None):
        cfg_dict = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        self.cfg = cfg
        # the reference raises NotImplementedError here for GFLHeadv2_uniH YAMLs unless task="detect" is passed
        # (engine/model.py:1096-1103 via tasks.py:1198-1210); the argument is accepted but not required
        self.task = task or guess_model_task(cfg_dict)
        if self.task != "detect":
            raise NotImplementedError(f"task '{self.task}': only 'detect' is built")
        self.model = DetectionModel(cfg_dict, nc=nc, verbose=verbose)
        self.overrides["model"] = self.cfg
        self.overrides["task"] = self.task
        self.model.task = self.task
        self.model_name = cfg_dict.get("yaml_file", "model") if isinstance(cfg, dict) else cfg

    def _load(self, weights, task=None):
        """Tensor-only checkpoints written by `save()`: {'yaml': <name or dict>, 'nc': int, 'fused': bool, 'state_dict': {...}}.
        Pickled reference checkpoints (module objects, tasks.py:815-955) are never unpickled here; tools/export_reference_weights.py
        turns one into this format on a machine that has the reference installed."""
        ck = torch_safe_load_state(weights)
        if not isinstance(ck, dict) or "state_dict" not in ck or "yaml" not in ck:
            raise ValueError(f"{weights}: expected a tensor-only checkpoint with 'yaml' and 'state_dict' entries (see YOLO.save)")
        cfg = ck["yaml"]
        if ck.get("nc") is not None:
            cfg = dict(yaml_model_load(cfg)) if not isinstance(cfg, dict) else dict(cfg)
            cfg["nc"] = int(ck["nc"])
        self._new(cfg, task=task)
        if bool(ck.get("fused", False)):
            self.model.fuse()  # the checkpoint holds BN-folded conv weights + biases (saved after predict()): same module layout first
        loaded, own = self.model.load(ck["state_dict"])
        if loaded != own:
            missing = sorted(set(self.model.state_dict()) - set(ck["state_dict"]))[:5]
            raise ValueError(f"{weights}: only {loaded} of {own} model tensors were found in the checkpoint (e.g. missing {missing}); "
                             "was it saved from a different YAML / nc, or fused without the 'fused' flag?")
        self.ckpt_path = weights

    def save(self, filename):
        """Tensor-only checkpoint (readable with torch.load(weights_only=True)).  predict() folds BatchNorm into the convs in place
        (reference AutoBackend does the same, autobackend.py:144-155); the 'fused' flag records which layout the tensors have."""
        m = self.model
        sd = {k: v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu() for k, v in m.state_dict().items()}
        # a model built from a dict (a custom architecture, or any model that went through _load) is saved WITH that dict: a path string
        # would only reload where a file of that name exists under cfg/models and still describes the same graph
        cfg = _plain(m.yaml) if isinstance(self.cfg, dict) else self.cfg
        torch.save({"yaml": cfg, "nc": int(m.yaml["nc"]),
                    "fused": bool(m.convs_folded()), "state_dict": sd}, filename)

    def load(self, weights):
        """Load a flat state_dict (dict or tensor-only file) keyed like the reference's `model.N....` entries."""
        sd = torch_safe_load_state(weights) if isinstance(weights, (str, Path)) else weights
        self.model.load(sd)
        self.predictor = None
        return self

    @property
    def task_map(self):
        """task -> {model class, predictor class} (reference engine/model.py:1062-1064, models/yolo/model.py:24-59)."""
        return {"detect": {"model": DetectionModel, "predictor": DetectionPredictor}}

    @property
    def names(self):
        return self.model.names

    @property
    def device(self):
        return next(self.model.parameters()).device

    def fuse(self):
        self.model.fuse()
        return self

    def _select_device(self, device):
        if device in (None, ""):
            device = "cuda:0"
        if isinstance(device, int):
            device = f"cuda:{device}"
        device = torch.device(device)
        if device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError(f"device '{device}': edge-yolo_amd runs on MI355X (ROCm 'cuda' devices) only; there is no CPU path. "
                               "(BASELINE configs[0], the CPU plumbing case, is the reference's own CPU predictor.)")
        return device

    def predict(self, source=None, stream=False, predictor=None, **kwargs):
        """kwargs (reference cfg/default.yaml:51-65 names): imgsz, half, conf, iou, max_det, agnostic_nms, classes, device, augment.
        predictor (reference engine/model.py:505,552: `(predictor or self._smart_load("predictor"))(overrides=...)`): a predictor CLASS to use
        instead of the task's default; it is constructed with DetectionPredictor's signature (see there) and called with the source."""
        if source is None:
            raise ValueError("predict() needs a source: a BCHW float tensor in [0,1] or HWC uint8 BGR ndarray(s)")
        if predictor is not None and not callable(predictor):
            raise TypeError(f"predict(predictor=...): expected a predictor class (constructed like DetectionPredictor), got {type(predictor).__name__}")
        args = {"conf": 0.25, "iou": 0.7, "max_det": 300, "half": False, "agnostic_nms": False, "classes": None, "device": None, "graph": True, "augment": False}
        unknown = set(kwargs) - set(args) - {"imgsz", "verbose", "batch", "save", "mode"}
        if unknown:
            raise TypeError(f"predict() got unsupported arguments {sorted(unknown)}")
        args.update({k: v for k, v in kwargs.items() if k in args})
        device = self._select_device(args["device"] if args["device"] is not None else (source.device if isinstance(source, torch.Tensor) and source.is_cuda else None))
        pcls = predictor or self.task_map[self.task]["predictor"]
        key = tuple((k, str(v)) for k, v in sorted(args.items())) + (("dev", str(device)), ("predictor", pcls))
        if self.predictor is None or self._pred_key != key:
            if self.predictor is not None:  # options changed: release the old predictor's graphs before anything new is captured
                self.predictor.close()
                self.predictor = None
            # AutoBackend order (reference nn/autobackend.py:144-155): to(device) -> fuse() -> half()/float()
            m = self.model.to(device)
            m.fuse()
            m = m.half() if args["half"] else m.float()
            m.eval()
            self.predictor = pcls(m, device, half=args["half"], conf=args["conf"], iou=args["iou"], max_det=args["max_det"],
                                  agnostic_nms=args["agnostic_nms"], classes=args["classes"], graph=args["graph"], augment=args["augment"])
            self._pred_key = key
        self.predictor.imgsz = kwargs.get("imgsz", 640)  # letterbox target for ndarray sources (reference cfg default 640)
        results = self.predictor(source)
        return iter(results) if stream else results

    __call__ = predict

    def predict_batches(self, batches, stages=4, **kwargs):
        """Throughput form of predict(): a generator over an iterable of BCHW float tensors in [0,1] (all of one shape; host or device,
        pinned host fp16/fp32 tensors upload fastest) -- or of decoded image batches, uint8 tensors (B,h,w,3) in BGR order like a stack
        of cv2 images (3 bytes per pixel over PCIe instead of 6; LetterBox to `imgsz` (default 640, minimum rectangle) + BGR->RGB + CHW + /255
        run on the device in one launch, boxes are scaled back to (h,w) like the reference does for image sources) -- that yields one list of
        Results per batch, in order.  The layer list is cut into
        `stages` pipeline stages (engine/predictor.py::PipelinedRunner: one hipGraph and one HIP stream per stage; batch i's head/NMS
        run beside batch i+1's neck and batch i+2's backbone), host batches are uploaded on a copy stream under the batches in flight,
        and the detection counts come back through a pinned buffer, so the only host waits are on events.  The results of a batch
        arrive `stages` submissions later (one buffer set per stage plus one for the upload in flight).  Same kwargs as predict() (conf, iou, max_det, half, agnostic_nms, classes, device).
        The reference has no counterpart (its stream=True generator still runs one batch at a time)."""
        from .predictor import PipelinedRunner
        from ..utils import ops
        args = {"conf": 0.25, "iou": 0.7, "max_det": 300, "half": False, "agnostic_nms": False, "classes": None, "device": None}
        unknown = set(kwargs) - set(args) - {"imgsz", "verbose", "cuts"}
        if unknown:
            raise TypeError(f"predict_batches() got unsupported arguments {sorted(unknown)}")
        args.update({k: v for k, v in kwargs.items() if k in args})
        pipe, post, pending, nset, dev_ctx = None, None, [], 0, None
        u8, lb, stage_u8, origs, upload = False, None, None, None, None
        try:
            for x in batches:
                if pipe is None:
                    u8 = x.dtype == torch.uint8
                    if u8 and (x.dim() != 4 or x.shape[3] != 3):
                        raise ValueError(f"predict_batches: uint8 batches must be (B,h,w,3) image stacks, got {tuple(x.shape)}")
                    device = self._select_device(args["device"] if args["device"] is not None else (x.device if x.is_cuda else None))
                    dev_ctx = torch.cuda.device(device)
                    dev_ctx.__enter__()
                    m = self.model.to(device)
                    m.fuse()
                    m = (m.half() if args["half"] else m.float()).eval()
                    post = DetectionPredictor(m, device, half=args["half"], conf=args["conf"], iou=args["iou"], max_det=args["max_det"],
                                              agnostic_nms=args["agnostic_nms"], classes=args["classes"], graph=False)
                    n = len(m.model)
                    stages = max(2, min(int(stages), n))
                    cuts = sorted({max(1, round(0.39 * (n - 1))), max(2, round(0.87 * (n - 1))), n - 1})[-(stages - 1):]
                    if kwargs.get("cuts"):
                        cuts = sorted(int(c) for c in kwargs["cuts"])
                    bounds = [0] + cuts + [n]
                    hn = {"conf": args["conf"], "classes": args["classes"]}  # the head stage also builds the NMS candidates (fused decode)
                    fns = [(lambda st, lo=lo, hi=hi: m.forward_layers(st if lo else (st, []), lo, hi, head_nms=hn)) for lo, hi in zip(bounds[:-1], bounds[1:])]
                    last = fns.pop()
                    fns.append(lambda st, last=last: ops.nms_device(last(st)[0][0], args["conf"], args["iou"], args["classes"], args["agnostic_nms"], args["max_det"])[:2])
                    head = m.model[-1]
                    fork, head.head_streams = getattr(head, "head_streams", False), False  # the head is a pipeline stage of its own: no fork inside it
                    dt = torch.float16 if args["half"] else torch.float32
                    if u8:
                        from ..data.augment import LetterBox
                        stride = int(max(m.stride)) if hasattr(m, "stride") else 32
                        lb = LetterBox(kwargs.get("imgsz", 640), auto=True, stride=stride)  # as predict() letterboxes a list of same-shape images
                        example = lb.batch_tensor(x.contiguous().to(device), dt)
                        u8_shape = tuple(x.shape)
                    else:
                        example = post.preprocess(x)
                    try:
                        pipe = PipelinedRunner(*fns, example, copy_stream=True)
                    finally:
                        head.head_streams = fork
                    nset = pipe.nsets
                    counts = [torch.empty(x.shape[0], dtype=torch.int32).pin_memory() for _ in range(nset)]
                    ready = [torch.cuda.Event() for _ in range(nset)]
                    # Image batches in pinned memory are NOT uploaded with a copy: the conversion kernel reads them over PCIe itself (pinned
                    # memory is mapped into the device's address space): 2.2 ms per batch instead of 2.9 with a DMA copy + conversion.
                    # On this stack neither form of transfer overlaps much with the kernels of other streams (a host-reading kernel beside
                    # 400 matmuls: 5.6 ms of transfers stretch them from 11.4 to 14.7 ms; DMA copies wait for them altogether), so a
                    # host batch costs about compute + transfer; 3 bytes per pixel is what helps.
                    if u8:
                        stage_u8 = [None] * nset
                        origs = [None] * nset

                        def upload(dst, src, j):  # ONE conversion launch (LetterBox + BGR->RGB + CHW + /255) on the first stage's stream
                            if not src.is_cuda and src.is_pinned():
                                return lambda: lb.batch_tensor(src, dt, out=dst)
                            if stage_u8[j] is None:
                                stage_u8[j] = torch.empty(u8_shape, dtype=torch.uint8, device=device)
                            stage_u8[j].copy_(src, non_blocking=True)
                            return lambda: lb.batch_tensor(stage_u8[j], dt, out=dst)
                    else:
                        origs = [None] * nset
                if u8:
                    if x.dtype != torch.uint8 or tuple(x.shape) != u8_shape:
                        raise ValueError(f"predict_batches: every batch must be a uint8 tensor of shape {u8_shape}, got {x.dtype} {tuple(x.shape)}")
                    x = x.contiguous()
                else:
                    if x.dim() != 4 or tuple(x.shape) != tuple(pipe.static_input(0).shape):  # checked for EVERY float batch, converted or not
                        raise ValueError(f"predict_batches: every batch must have shape {tuple(pipe.static_input(0).shape)}, got {tuple(x.shape)}")
                    if not x.is_cuda and x.dtype != dt:  # a dtype-changing H2D copy would convert on the host: upload as is, convert on the device
                        x = post.preprocess(x)
                while pending and (len(pending) >= nset or pending[0] == pipe.i % nset):  # the buffer set about to be reused must be read first
                    yield self._finish(pipe, post, pending.pop(0), counts, ready, origs)
                # float host batches: DMA on a stream of its own (measured faster than the host-reading copy kernel for 6-byte pixels);
                # image batches: the conversion kernel reads pinned memory itself; device batches: a device copy ahead of stage 0
                j = pipe.submit(x, upload=upload, side_copy=not u8 and not x.is_cuda)
                origs[j] = x  # (keeps a pinned host batch alive and, for image batches, is what the Results refer to)
                with torch.cuda.stream(pipe.sp):  # counts -> pinned host memory right behind this batch's NMS; the host later waits on the event only
                    counts[j].copy_(pipe.outputs(j)[1], non_blocking=True)
                    ready[j].record(pipe.sp)
                pending.append(j)
            while pending:
                yield self._finish(pipe, post, pending.pop(0), counts, ready, origs)
        finally:
            if dev_ctx is not None:
                dev_ctx.__exit__(None, None, None)

    @staticmethod
    def _finish(pipe, post, j, counts, ready, origs=None):
        ready[j].synchronize()
        boxes, _ = pipe.outputs(j)
        # image sources: boxes go back to the original image frame (reference detect/predict.py:36-39); the originals are views of the host batch
        src = origs[j] if origs is not None else None
        if src is not None and src.dtype != torch.uint8:
            src = None  # (float tensors are network inputs, not images: tensor-source semantics)
        post._orig = [src[i].numpy() if not src.is_cuda else src[i] for i in range(src.shape[0])] if src is not None else None
        return post.postprocess(boxes, counts[j], pipe.static_input(j), None)


class YOLO(Model):
    """YOLO detect model (reference models/yolo/model.py:11-59)."""

    def __init__(self, model="yolo11n.yaml", task=None, verbose=False, nc=None):
        super().__init__(model=model, task=task, verbose=verbose, nc=nc)

    @property
    def task_map(self):
        return {"detect": {"model": DetectionModel, "predictor": DetectionPredictor}}
