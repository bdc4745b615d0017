"""`YOLO(...)` facade for the detect task, mirroring the reference's engine/model.py (`Model.__init__` :84-151,
`_new` :231-264, `_load` :266-302, `predict` :501-560) and models/yolo/model.py (:11-59)."""
from pathlib import Path

import torch

from .predictor import DetectionPredictor
from ..nn.tasks import DetectionModel, guess_model_task, torch_safe_load_state, yaml_model_load


class Model(torch.nn.Module):
    def __init__(self, model="yolo11n.yaml", task=None, verbose=False):
        super().__init__()
        self.predictor = None
        self.model = None
        self.overrides = {}
        self.ckpt_path = None
        self.task = task
        model = str(model).strip()
        if Path(model).suffix in {".yaml", ".yml"}:
            self._new(model, task=task, verbose=verbose)
        else:
            self._load(model, task=task)

    def _new(self, cfg, task=None, model=None, verbose=False):
        cfg_dict = yaml_model_load(cfg)
        self.cfg = cfg
        # the reference raises NotImplementedError here for GFLHeadv2_uniH YAMLs unless task="detect" is passed
        # (engine/model.py:1096-1103 via tasks.py:1198-1210); the argument is accepted but not required
        self.task = task or guess_model_task(cfg_dict)
        if self.task != "detect":
            raise NotImplementedError(f"task '{self.task}': only 'detect' is built")
        self.model = DetectionModel(cfg_dict, verbose=verbose)
        self.overrides["model"] = self.cfg
        self.overrides["task"] = self.task
        self.model.task = self.task
        self.model_name = cfg

    def _load(self, weights, task=None):
        """Tensor-only checkpoints written by `save()`: {'yaml': <name or dict>, 'nc': int, 'state_dict': {...}}.
        Pickled reference checkpoints (module objects, tasks.py:815-955) are never unpickled here."""
        ck = torch_safe_load_state(weights)
        if not isinstance(ck, dict) or "state_dict" not in ck or "yaml" not in ck:
            raise ValueError(f"{weights}: expected a tensor-only checkpoint with 'yaml' and 'state_dict' entries (see YOLO.save)")
        self._new(ck["yaml"], task=task)
        self.model.load(ck["state_dict"])
        self.ckpt_path = weights

    def save(self, filename):
        torch.save({"yaml": self.cfg, "nc": self.model.yaml["nc"], "state_dict": self.model.state_dict()}, filename)

    def load(self, weights):
        """Load a flat state_dict (dict or tensor-only file) keyed like the reference's `model.N....` entries."""
        sd = torch_safe_load_state(weights) if isinstance(weights, (str, Path)) else weights
        self.model.load(sd)
        self.predictor = None
        return self

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
        """kwargs (reference cfg/default.yaml:51-65 names): imgsz, half, conf, iou, max_det, agnostic_nms, classes, device."""
        if source is None:
            raise ValueError("predict() needs a source: a BCHW float tensor in [0,1] or HWC uint8 BGR ndarray(s)")
        args = {"conf": 0.25, "iou": 0.7, "max_det": 300, "half": False, "agnostic_nms": False, "classes": None, "device": None, "graph": True}
        unknown = set(kwargs) - set(args) - {"imgsz", "verbose", "batch", "save", "mode"}
        if unknown:
            raise TypeError(f"predict() got unsupported arguments {sorted(unknown)}")
        args.update({k: v for k, v in kwargs.items() if k in args})
        device = self._select_device(args["device"] if args["device"] is not None else (source.device if isinstance(source, torch.Tensor) and source.is_cuda else None))
        key = tuple((k, str(v)) for k, v in sorted(args.items())) + (("dev", str(device)),)
        if self.predictor is None or self._pred_key != key:
            # AutoBackend order (reference nn/autobackend.py:144-155): to(device) -> fuse() -> half()/float()
            m = self.model.to(device)
            m.fuse()
            m = m.half() if args["half"] else m.float()
            m.eval()
            self.predictor = DetectionPredictor(m, device, half=args["half"], conf=args["conf"], iou=args["iou"], max_det=args["max_det"],
                                                agnostic_nms=args["agnostic_nms"], classes=args["classes"], graph=args["graph"])
            self._pred_key = key
        self.predictor.imgsz = kwargs.get("imgsz", 640)  # letterbox target for ndarray sources (reference cfg default 640)
        results = self.predictor(source)
        return iter(results) if stream else results

    __call__ = predict

    def predict_batches(self, batches, stages=4, **kwargs):
        """Throughput form of predict(): a generator over an iterable of BCHW float tensors in [0,1] (all of one shape) that yields one
        list of Results per batch, in order.  The layer list is cut into `stages` pipeline stages (engine/predictor.py::PipelinedRunner:
        one hipGraph and one HIP stream per stage; batch i's head/NMS run beside batch i+1's neck and batch i+2's backbone), so the
        results of a batch arrive `stages - 1` submissions later.  Same kwargs as predict() (conf, iou, max_det, half, agnostic_nms,
        classes, device).  The reference has no counterpart (its stream=True generator still runs one batch at a time)."""
        from .predictor import PipelinedRunner
        from ..utils import ops
        args = {"conf": 0.25, "iou": 0.7, "max_det": 300, "half": False, "agnostic_nms": False, "classes": None, "device": None}
        unknown = set(kwargs) - set(args) - {"imgsz", "verbose"}
        if unknown:
            raise TypeError(f"predict_batches() got unsupported arguments {sorted(unknown)}")
        args.update({k: v for k, v in kwargs.items() if k in args})
        pipe, post, pending, nset = None, None, [], 0
        for x in batches:
            if pipe is None:
                device = self._select_device(args["device"] if args["device"] is not None else (x.device if x.is_cuda else None))
                m = self.model.to(device)
                m.fuse()
                m = (m.half() if args["half"] else m.float()).eval()
                post = DetectionPredictor(m, device, half=args["half"], conf=args["conf"], iou=args["iou"], max_det=args["max_det"],
                                          agnostic_nms=args["agnostic_nms"], classes=args["classes"], graph=False)
                n = len(m.model)
                stages = max(2, min(int(stages), n))
                cuts = sorted({max(1, round(0.39 * (n - 1))), max(2, round(0.87 * (n - 1))), n - 1})[-(stages - 1):]
                bounds = [0] + cuts + [n]
                fns = [(lambda st, lo=lo, hi=hi: m.forward_layers(st if lo else (st, []), lo, hi)) for lo, hi in zip(bounds[:-1], bounds[1:])]
                last = fns.pop()
                fns.append(lambda st, last=last: ops.nms_device(last(st)[0][0], args["conf"], args["iou"], args["classes"], args["agnostic_nms"], args["max_det"])[:2])
                import edge_yolo_amd.nn.modules.head as _hm
                fork, _hm._HEAD_STREAMS = _hm._HEAD_STREAMS, False  # the head is a pipeline stage of its own: no fork inside it while capturing
                try:
                    pipe = PipelinedRunner(*fns, post.preprocess(x))
                finally:
                    _hm._HEAD_STREAMS = fork
                nset = pipe.n
            im = post.preprocess(x)
            while pending and (len(pending) >= nset or pending[0][0] == pipe.i % nset):  # the buffer set about to be reused must be read first
                yield self._finish(pipe, post, *pending.pop(0))
            pending.append((pipe.submit(im), im))
        while pending:
            yield self._finish(pipe, post, *pending.pop(0))

    @staticmethod
    def _finish(pipe, post, j, im):
        pipe.wait(j)
        boxes, count = pipe.outputs(j)
        post._orig = None
        return post.postprocess(boxes, count, im, None)


class YOLO(Model):
    """YOLO detect model (reference models/yolo/model.py:11-59)."""

    def __init__(self, model="yolo11n.yaml", task=None, verbose=False):
        super().__init__(model=model, task=task, verbose=verbose)

    @property
    def task_map(self):
        return {"detect": {"model": DetectionModel, "predictor": DetectionPredictor}}
