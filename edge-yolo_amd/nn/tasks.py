"""YAML -> model graph and the graph executor, mirroring the reference's ultralytics/nn/tasks.py for the detect
path: `parse_model` (:958-1147), `yaml_model_load` (:1150-1163), `guess_model_scale` (:1166-1181),
`guess_model_task` (:1184-1255), `BaseModel._predict_once/fuse` (:152-242), `DetectionModel` (:320-370).
"""
import ast
import contextlib
import re
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn
import yaml

from . import _ops as ops
from .modules import (C2PSA, C2PSA_LinearAttention, C2f, C3, C3k2, Concat, Conv, DSC3K2_Wavelet, DSConv, DWConv, Detect, E2EDetect, GF2Detect,
                      GFLHeadv2_uniH, SPPF, Upsample)
from .modules import *  # noqa: F401,F403  (registry: YAML names resolve through globals(), as in the reference)
from .modules.conv import _Packed
from .. import _lib as L
from ..utils.ops import make_divisible

CFG_DIR = Path(__file__).resolve().parent.parent / "cfg" / "models"
_CH_MODULES = {Conv, SPPF, C2PSA, C2PSA_LinearAttention, DWConv, C2f, C3k2, DSC3K2_Wavelet, C3, DSConv}
_REPEAT_MODULES = {C2f, C3k2, DSC3K2_Wavelet, C3, C2PSA, C2PSA_LinearAttention}
_HEADS = {Detect, GF2Detect, E2EDetect, GFLHeadv2_uniH}


def guess_model_scale(model_path):
    """'yolo11n-test.yaml' -> 'n' (reference tasks.py:1166-1181)."""
    try:
        return re.search(r"yolo[v]?\d+([nslmx])", Path(model_path).stem).group(1)
    except AttributeError:
        return ""


def yaml_model_load(path):
    """Load a model YAML; 'yolo11n-test.yaml' resolves to the unified 'yolo11-test.yaml' + scale 'n' (tasks.py:1150-1163).
    Bare file names are looked up under edge-yolo_amd/cfg/models/**."""
    path = Path(path)
    unified = Path(re.sub(r"(\d+)([nslmx])(.+)?$", r"\1\3", str(path)))
    for cand in (unified, path):
        for f in ([cand] if cand.exists() else sorted(CFG_DIR.rglob(cand.name))):
            d = yaml.safe_load(open(f, encoding="utf-8"))
            d["scale"] = guess_model_scale(path)
            d["yaml_file"] = str(path)
            return d
    raise FileNotFoundError(f"model YAML '{path}' not found (also searched {CFG_DIR})")


def guess_model_task(model):
    """Reference tasks.py:1184-1255 reduced to this build's scope.  The reference only recognises heads whose
    lower-cased name contains 'detect' and raises for GFLHeadv2_uniH YAMLs unless task='detect' is passed
    (SURVEY §3 quirk); here every registered head is a detect head."""
    cfg = model if isinstance(model, dict) else None
    if cfg is None and isinstance(model, (str, Path)):
        with contextlib.suppress(Exception):
            cfg = yaml_model_load(model)
    if cfg is not None:
        m = cfg["head"][-1][-2]
        if m in {h.__name__ for h in _HEADS}:
            return "detect"
        raise NotImplementedError(f"head '{m}': only the detect task is built")
    return "detect"


def parse_model(d, ch, verbose=False):
    """YAML dict -> (nn.Sequential, savelist).  Channel/repeat rewriting as reference tasks.py:958-1147."""
    legacy = True
    max_channels = float("inf")
    nc, act, scales = (d.get(x) for x in ("nc", "activation", "scales"))
    depth, width = (d.get(x, 1.0) for x in ("depth_multiple", "width_multiple"))
    scale = d.get("scale")
    if scales:
        if not scale:
            scale = tuple(scales.keys())[0]
        depth, width, max_channels = scales[scale]
    if act:
        Conv.default_act = eval(act)  # noqa: S307  (same YAML contract as the reference, tasks.py:974-975)
    ch = [ch]
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        args = list(args)
        if m == "nn.Upsample":
            m = Upsample
        elif isinstance(m, str) and m.startswith("nn."):
            raise NotImplementedError(f"torch module '{m}' is not on the built detection path")
        else:
            if m not in globals():
                raise NotImplementedError(f"module '{m}' is not in the edge-yolo_amd registry (SURVEY.md §8a lists what is)")
            m = globals()[m]
        for j, a in enumerate(args):
            if isinstance(a, str):
                with contextlib.suppress(ValueError):
                    args[j] = locals()[a] if a in locals() else ast.literal_eval(a)
        n = n_ = max(round(n * depth), 1) if n > 1 else n
        if m in _CH_MODULES:
            c1, c2 = ch[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [c1, c2, *args[1:]]
            if m in _REPEAT_MODULES:
                args.insert(2, n)
                n = 1
            if m in {C3k2, DSC3K2_Wavelet}:
                legacy = False
                if scale in "lx":
                    args[3] = True
        elif m is Concat:
            c2 = sum(ch[x] for x in f)
        elif m in _HEADS:
            args.append([ch[x] for x in f])
            m.legacy = legacy
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*(m(*args) for _ in range(n))) if n > 1 else m(*args)
        t = f"{m.__module__}.{m.__name__}".replace("edge_yolo_amd", "ultralytics") if m is not Upsample else "torch.nn.modules.upsampling.Upsample"
        m_.np = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type = i, f, t
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


class BaseModel(nn.Module):
    """forward / predict / _predict_once / fuse of the reference BaseModel (tasks.py:113-317), inference only."""

    def forward(self, x, *args, **kwargs):
        if isinstance(x, dict):
            raise NotImplementedError("training losses are outside the built path (inference forward only)")
        return self.predict(x, *args, **kwargs)

    def predict(self, x, profile=False, visualize=False, augment=False, embed=None, head_nms=None):
        """head_nms (extension, see Detect.forward): dict(conf=, classes=, keep_pred=) -> the head also builds the NMS candidates."""
        if profile or visualize or embed:
            raise NotImplementedError("profile/visualize/embed hooks are not part of the built path")
        if torch.is_tensor(x) and x.is_cuda and x.device.index != torch.cuda.current_device():
            with torch.cuda.device(x.device):  # launches go to the current device's stream (_lib.stream)
                return self._predict_augment(x) if augment else self._predict_once(x, head_nms=head_nms)
        if augment:  # reference tasks.py:147-148
            if head_nms is not None:
                raise ValueError("augment=True returns the concatenated multi-scale predictions; the fused NMS candidate build (head_nms) does not apply")
            return self._predict_augment(x)
        return self._predict_once(x, head_nms=head_nms)

    def _predict_augment(self, x):
        """BaseModel's fallback (reference tasks.py:181-187): models without TTA warn and run single-scale.  DetectionModel overrides."""
        import warnings
        warnings.warn(f"{self.__class__.__name__} does not support 'augment=True', reverting to single-scale prediction.")
        return self._predict_once(x)

    def _predict_once(self, x, profile=False, visualize=False, embed=None, head_nms=None):
        return self.forward_layers((x, []), 0, len(self.model), head_nms=head_nms)[0]

    def forward_features(self, x):
        """_predict_once without the last (head) module: returns the head's input list (for callers that pipeline the head apart)."""
        y = []
        for m in self.model[:-1]:
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            x = m(x)
            y.append(x if m.i in self.save else None)
        h = self.model[-1]
        return [x if j == -1 else y[j] for j in h.f] if not isinstance(h.f, int) else (x if h.f == -1 else y[h.f])

    def forward_layers(self, state, lo, hi, head_nms=None):
        """Layers [lo, hi) of _predict_once (reference tasks.py:152-179) on `state` = (x, saved outputs so far) -> new state: lets a
        caller cut the graph into pipeline stages (each stage a captured hipGraph on its own stream).  head_nms: passed to a Detect
        head in the range (fused NMS candidate build, see Detect.forward)."""
        x, y = state
        y = list(y)
        i = lo
        while i < hi:
            j = self._block_end(i, hi, x)
            if j > i + 1 or (j == i + 1 and i in getattr(self, "_block_of", {}) and self._block_single_ok(i)):
                got = self._run_block(i, j, x, y)
                if got is not None:
                    x, y = got
                    i = j
                    continue
            m = self.model[i]
            if i == 0 and hi > 1 and 0 not in self.save and torch.is_tensor(x) and type(m) is Conv and type(self.model[1]) is Conv and self.model[1].f == -1:
                got = ops.stem_pair(m, self.model[1], x)  # layers 0 + 1 as one kernel: the stem's output never leaves the chip
                if got is not None:
                    x = got
                    y.extend([None, x if 1 in self.save else None])
                    i = 2
                    continue
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j2 == -1 else y[j2] for j2 in m.f]
            if (type(m) is DSC3K2_Wavelet and i + 1 < hi and i not in self.save and type(self.model[i + 1]) is Conv and self.model[i + 1].f == -1
                    and self.model[i + 1].conv.stride == (2, 2) and self.model[i + 1].conv.kernel_size == (3, 3)):
                x = m(x, tail=self.model[i + 1])  # the block's closing 1x1 + the down-sampling conv behind it as one kernel where the shape allows
                y.extend([None, x if i + 1 in self.save else None])
                i += 2
                continue
            x = m(x, nms=head_nms) if (head_nms is not None and isinstance(m, Detect)) else m(x)
            y.append(x if m.i in self.save else None)
            i += 1
        return x, y

    # ---- block programs (nn/_block.py): runs of consecutive small-map layers as ONE launch -------------------------------------
    # Opt-in (default off): measured on MI355X at batch 32 the one-workgroup-per-image block programs cut the step from 116 to 76
    # launches but run ~5x longer than the per-layer kernels they replace (a 1024-thread workgroup has 128 VGPRs per wave and one CU's
    # 64 B/clk vector-memory path: profiles/r02_block_stage_times.txt) -- kept as a tested option, not the default
    block_fusion = False
    block_max_pixels = 1024  # a run is block-executed while its first layer's output map has at most this many pixels (20x20 at 640x640)

    def _block_end(self, i, hi, x):
        """End (exclusive) of the block-executable run of layers starting at i within [i, hi): the maximal stretch of the precomputed
        run (`_block_of`: consecutive layers at the coarsest stride) -- or i when layer i starts none / fusion is off / fp32 mode."""
        runs = getattr(self, "_block_of", None)
        if not self.block_fusion or not runs or i not in runs or not torch.is_tensor(x) or x.dtype != torch.float16 or not x.is_cuda:
            return i
        lo, hi_run, down = runs[i]
        B, _, H, W = x.shape
        ratio = down / self._down[i - 1] if i > 0 else down
        if (H // ratio) * (W // ratio) > self.block_max_pixels or B < 2:
            return i
        return min(hi, hi_run)

    def _block_single_ok(self, i):
        return not isinstance(self.model[i], (Concat, Upsample))

    def _run_block(self, lo, hi, x, y):
        """Layers [lo, hi) through one block program.  Inputs = the current x and the saved outputs of earlier layers the run reads;
        outputs = the final x and the saved outputs of the run.  Returns (x, y) or None (not block-executable: per-layer kernels)."""
        from ._block import BlockCache
        need = set()
        for m in self.model[lo:hi]:
            for f in ([m.f] if isinstance(m.f, int) else m.f):
                if f != -1:  # (-1 = the previous layer: inside the run, or x for its first layer)
                    j = f if f >= 0 else m.i + f
                    if j < lo:
                        need.add(j)
        need = sorted(need)
        if any(y[j] is None for j in need):
            return None
        ins = [x] + [y[j] for j in need]
        if not all(torch.is_tensor(t) for t in ins):
            return None
        keep = [m.i for m in self.model[lo:hi] if m.i in self.save]

        def chain(x0, *extra):
            yy = list(y)
            for j, t in zip(need, extra):
                yy[j] = t
            xx = x0
            for m in self.model[lo:hi]:
                if m.f != -1:
                    xx = yy[m.f] if isinstance(m.f, int) else [xx if j2 == -1 else yy[j2] for j2 in m.f]
                xx = m(xx)
                yy.append(xx if m.i in self.save else None)
            if not torch.is_tensor(xx):
                from ._block import BlockUnsupported
                raise BlockUnsupported("the run does not end in a tensor")
            return [xx] + [yy[i] for i in keep if i != hi - 1]

        caches = self.__dict__.setdefault("_block_caches", {})
        cache = caches.get((lo, hi))
        if cache is None:
            cache = caches[(lo, hi)] = BlockCache(f"layers {lo}-{hi - 1}")
        outs = cache.run(chain, ins)
        if outs is None:
            return None
        y = list(y)
        extra = iter(outs[1:])
        for m in self.model[lo:hi]:
            y.append(outs[0] if (m.i == hi - 1 and m.i in self.save) else (next(extra) if m.i in self.save else None))
        return outs[0], y

    def fuse(self, verbose=False):
        """Fold every Conv/DWConv BatchNorm into its conv parameters (reference tasks.py:214-242).  DSConv keeps its
        BatchNorm as a module, as in the reference; it is folded when the HIP weights are packed."""
        if not self.is_fused():
            for m in self.model.modules():
                if isinstance(m, Conv) and hasattr(m, "bn"):
                    m.fuse_bn()
        return self

    def convs_folded(self):
        """True once fuse() ran: no Conv/DWConv still carries its BatchNorm (DSConv BatchNorms stay modules by design, so the
        reference's count-based is_fused() below says False for EdgeLine YAMLs even after fuse(); checkpoints record this flag)."""
        return not any(isinstance(m, Conv) and hasattr(m, "bn") for m in self.model.modules())

    def is_fused(self, thresh=10):
        bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
        return sum(isinstance(v, bn) for v in self.modules()) < thresh

    def _apply(self, fn, *args, **kwargs):
        self.__dict__["_block_caches"] = {}  # recorded block programs hold packed weights: parameters moved / converted -> re-record
        self = super()._apply(fn, *args, **kwargs)
        m = self.model[-1]
        if isinstance(m, Detect):
            m.stride = fn(m.stride)
            m.anchors = fn(m.anchors)
            m.strides = fn(m.strides)
        return self

    def load(self, weights, verbose=False):
        """Load a flat state_dict (tensor-only file or dict); intersecting keys/shapes only, like reference :276-291."""
        sd = weights["state_dict"] if isinstance(weights, dict) and "state_dict" in weights else weights
        own = self.state_dict()
        csd = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
        self.load_state_dict(csd, strict=False)
        return len(csd), len(own)


class DetectionModel(BaseModel):
    """Detection model built from a YAML (reference tasks.py:320-370)."""

    def __init__(self, cfg="yolo11n.yaml", ch=3, nc=None, verbose=False):
        super().__init__()
        self.yaml = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=ch, verbose=verbose)
        self.names = {i: f"{i}" for i in range(self.yaml["nc"])}
        self.inplace = self.yaml.get("inplace", True)
        self.end2end = getattr(self.model[-1], "end2end", False)
        m = self.model[-1]
        if isinstance(m, Detect):
            # The reference measures strides with a 256x256 zero forward (tasks.py:352-364); the same numbers follow
            # from the graph (product of conv strides / upsample factors on the path to each head input) without
            # needing a device at construction time.
            m.stride = torch.tensor(self._graph_strides(ch))
            self.stride = m.stride
        self._fold_upsample_concat()
        self._plan_blocks()
        self.register_load_state_dict_post_hook(lambda m, _keys: m.__dict__.__setitem__("_block_caches", {}))
        for mod in self.modules():  # initialize_weights (utils/torch_utils.py:410-420)
            if isinstance(mod, nn.BatchNorm2d):
                mod.eps = 1e-3
                mod.momentum = 0.03
        self.eval()

    def _fold_upsample_concat(self):
        """K7: mark Upsample/Concat layers whose every consumer starts with a 1x1 conv as lazy: they then hand a
        VirtualCat to that conv instead of writing the upsampled / concatenated tensor (never changes results)."""
        from .modules import C2PSA, C2PSA_LinearAttention
        consumers = {}
        for m in self.model:
            for j in ([m.f] if isinstance(m.f, int) else m.f):
                consumers.setdefault((m.i + j) if j < 0 else j, []).append(m)

        def takes_virtual(m):
            if isinstance(m, (C2f, DSC3K2_Wavelet, C2PSA, C2PSA_LinearAttention)):
                return True
            return isinstance(m, Conv) and m.conv.kernel_size == (1, 1) and m.conv.groups == 1

        for m in self.model:
            if isinstance(m, Concat) and m.i not in self.save:
                m.lazy = all(takes_virtual(c) for c in consumers.get(m.i, [])) and bool(consumers.get(m.i))
        for m in self.model:
            if isinstance(m, Upsample) and m.i not in self.save:
                cs = consumers.get(m.i, [])
                m.lazy = bool(cs) and all(isinstance(c, Concat) and c.lazy for c in cs)

    def _plan_blocks(self):
        """Runs of >= 2 consecutive non-head layers at the coarsest stride (layers 7-10 and 20-22 of the 24-layer YAMLs): candidates for
        block programs.  `_down[i]` = downsampling factor of layer i's output; `_block_of[i]` = (lo, hi, down) of the run i lies in."""
        down = []
        for m in self.model:
            f = m.f if isinstance(m.f, int) else m.f[0]
            src = 1.0 if m.i == 0 else (down[f] if f != -1 else down[-1])
            if isinstance(m, Conv):
                src *= m.conv.stride[0]
            elif isinstance(m, Upsample):
                src /= 2
            down.append(src)
        self._down = down
        top = max(down[:-1]) if len(down) > 1 else 0
        self._block_of = {}
        i, n = 0, len(self.model) - 1  # (the head is handled inside Detect)
        while i < n:
            if down[i] == top and not isinstance(self.model[i], Upsample):
                j = i
                while j < n and down[j] == top and not isinstance(self.model[j], Upsample):
                    j += 1
                if j - i >= 2:
                    for q in range(i, j):
                        self._block_of[q] = (i, j, top)
                i = j
            else:
                i += 1

    def _graph_strides(self, ch):
        down = []
        for m in self.model:
            f = m.f if isinstance(m.f, int) else m.f[0]
            src = 1.0 if (m.i == 0) else (down[f] if f != -1 else down[-1])
            if isinstance(m, Conv):
                src *= m.conv.stride[0]
            elif isinstance(m, Upsample):
                src /= 2
            if isinstance(m, Detect):
                return [float(down[j]) for j in m.f]
            down.append(src)
        raise RuntimeError("no detect head")

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("edge-yolo_amd builds the inference forward path only")
        return super().train(False)

    def _predict_augment(self, x):
        """Scale / flip test-time augmentation (reference DetectionModel._predict_augment, tasks.py:372-387): forwards at scales 1, 0.83 and
        0.67 (the middle one on the left-right mirrored image), `_descale_pred` (:388-397), `_clip_augmented` (:399-408) and the concat over
        anchors -> ((B, 4+nc, A_total) fp32, None).  Same kernels as the single-scale path, three times, plus ey_scale_img (flip + bilinear
        resize + 0.447 padding) in front and ey_tta_merge (de-scale, de-flip, clip, concat) behind each forward."""
        if self.end2end or self.__class__.__name__ != "DetectionModel":  # the reference's own guard (:374-376)
            return super()._predict_augment(x)
        from . import _ops
        img_hw = tuple(x.shape[-2:])
        gs = int(self.stride.max())
        nl = self.model[-1].nl
        g = sum(4 ** k for k in range(nl))
        scales, flips = (1, 0.83, 0.67), (None, 3, None)
        preds = []
        for si, fi in zip(scales, flips):
            xi = _ops.scale_img(x, float(si), flip_lr=(fi == 3), gs=gs)
            preds.append(self._predict_once(xi)[0])
        # _clip_augmented: the first (full-size) output loses its last A//g anchors (its coarsest level), the last (smallest) its first
        # (A//g)*4^(nl-1) (its finest level); e = 1 excluded level, exactly the reference's index arithmetic
        spans = [(0, p.shape[-1]) for p in preds]
        spans[0] = (0, preds[0].shape[-1] - (preds[0].shape[-1] // g) * 1)
        spans[-1] = ((preds[-1].shape[-1] // g) * 4 ** (nl - 1), preds[-1].shape[-1])
        total = sum(hi - lo for lo, hi in spans)
        out = torch.empty((x.shape[0], preds[0].shape[1], total), dtype=torch.float32, device=x.device)
        off = 0
        for p, (lo, hi), si, fi in zip(preds, spans, scales, flips):
            _ops.tta_merge(p, lo, hi, si, fi, img_hw, out, off)
            off += hi - lo
        return out, None


def torch_safe_load_state(path):
    """Tensor-only checkpoint reader (never unpickles code): torch.load(weights_only=True) or safetensors."""
    path = str(path)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)
