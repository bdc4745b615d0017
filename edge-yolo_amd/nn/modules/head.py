"""Detection heads, HIP-backed: `Detect` (reference head.py:38-189), `GF2Detect` (:194-345), `E2EDetect` (:799-824),
`GFLHeadv2_uniH` (:827-908).  Same constructor signatures / attribute names / state_dict keys.  The towers are MFMA convs and
depthwise kernels; everything after them (DGQP statistics + quality FCs, DFL softmax-expectation, anchor decode,
score modulation) is ONE fused kernel per pyramid level writing the (B, 4+nc, A) fp32 prediction tensor.
"""
import copy
import math

import torch
import torch.nn as nn

from .block import DFL
from .conv import Conv, DWConv, _Packed, fold_bn
from .. import _ops as ops
from ... import _lib as L

__all__ = ("Detect", "GF2Detect", "E2EDetect", "GFLHeadv2_uniH")


class _Plain(_Packed):
    """Weights of a bare nn.Conv2d(+bias) 1x1 tower tail, packed for the MFMA conv."""

    def __init__(self, conv):
        super().__init__()
        self.__dict__["_ref"] = conv  # not registered: the nn.Conv2d stays where the reference keeps it

    def run(self, x, out):
        c = self._ref
        return ops.conv2d(self, [x], lambda: fold_bn(c.weight, c.bias, None), 1, 1, 0, L.ACT_NONE, out=out)


class Detect(nn.Module):
    """YOLO Detect head (reference head.py:38-189)."""

    dynamic = False
    export = False
    format = None
    end2end = False
    max_det = 300
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)
    legacy = False
    # Execution options (instance attributes; they never change results):
    #  head_streams: run the towers of the smaller pyramid levels on a second HIP stream (a parallel hipGraph branch).  A caller
    #                that already runs the head as a pipeline stage of its own (YOLO.predict_batches, bench.py) sets it False.
    #  tower_streams: optional stream index (0 = the caller's stream) per tower in order (level0 box, level0 cls, level1 box, ...)
    #  chain: fuse the last two 1x1 convs of the class tower into one register-only kernel (ey_conv_pw_chain)
    head_streams = True
    tower_streams = None
    chain = True
    #  block_fusion: both towers of a pyramid level whose map has at most block_max_pixels pixels (20x20 at 640x640) run as ONE block
    #                program (nn/_block.py: one persistent workgroup per image) instead of 7 launches; f16 mode only.  Opt-in: measured
    #                slower than the per-layer kernels at batch 32 (see DetectionModel.block_fusion)
    block_fusion = False
    block_max_pixels = 1024

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch)
        self.cv3 = (
            nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
            if self.legacy
            else nn.ModuleList(
                nn.Sequential(nn.Sequential(DWConv(x, x, 3), Conv(x, c3, 1)), nn.Sequential(DWConv(c3, c3, 3), Conv(c3, c3, 1)),
                              nn.Conv2d(c3, self.nc, 1))
                for x in ch)
        )
        self.dfl = DFL(self.reg_max) if self.reg_max > 1 else nn.Identity()
        if self.end2end:  # head.py:76-78
            self.one2one_cv2 = copy.deepcopy(self.cv2)
            self.one2one_cv3 = copy.deepcopy(self.cv3)
        self._tails = {}
        # packed tails / quality-head weights are caches of parameters: drop them whenever parameters are (re)loaded
        self.register_load_state_dict_post_hook(lambda m, _keys: m._reset_caches())

    def _reset_caches(self):
        self.__dict__["_blk"] = {}
        self._tails = {}
        self._stride_f = None
        if hasattr(self, "_qcache"):
            self._qcache = {}

    # ---- towers
    def _tail(self, conv):
        t = self._tails.get(id(conv))
        if t is None:
            t = self._tails[id(conv)] = _Plain(conv)
        return t

    def _apply(self, fn, *a, **k):
        self._reset_caches()
        return super()._apply(fn, *a, **k)

    # branch of the towers the inference decodes: "" (cv2 / cv3 / reg_conf) or "one2one_" (their end2end copies, head.py:76-78,220-221)
    _branch = ""

    def _box_tower(self, i, x, raw):
        """box logits -> raw[:, :64]."""
        b = getattr(self, self._branch + "cv2")[i]
        self._tail(b[2]).run(b[1](b[0](x)), raw[:, :4 * self.reg_max])

    def _cls_tower(self, i, x, raw):
        """class logits -> raw[:, 64:]."""
        c = getattr(self, self._branch + "cv3")[i]
        t = x
        out = raw[:, 4 * self.reg_max:]
        last = c[-2]
        # non-legacy tower (head.py:68-70): ... -> DWConv -> Conv(c3,c3,1)+SiLU -> nn.Conv2d(c3,nc,1): the last two 1x1 convs run as ONE
        # register-only kernel when the shape fits (ey_conv_pw_chain), the intermediate (B,c3,H,W) tensor never exists
        if (self.chain and isinstance(last, nn.Sequential) and len(last) == 2 and isinstance(last[0], DWConv) and isinstance(last[1], Conv)
                and last[1].conv.kernel_size == (1, 1) and isinstance(last[1].act, nn.SiLU) and x.dtype == torch.float16):
            for j in range(len(c) - 2):
                t = c[j](t)
            t = last[0](t)
            tail = c[-1]
            if ops.conv_pw_chain(self._tail(tail), t, last[1].folded, L.ACT_SILU, lambda: fold_bn(tail.weight, tail.bias, None), L.ACT_NONE, out) is not None:
                return
            self._tail(tail).run(last[1](t), out)
            return
        for j in range(len(c) - 1):
            t = c[j](t)  # (self._dw_pw fuses DWConv+Conv into one kernel; measured slower than the two kernels at these widths)
        self._tail(c[-1]).run(t, out)

    def _towers(self, i, x, raw):
        self._box_tower(i, x, raw)
        self._cls_tower(i, x, raw)

    def _towers_block(self, i, x, raw):
        """Both towers of level i as one block program; False when the level is not block-executable (per-layer kernels then)."""
        if not self.block_fusion or x.dtype != torch.float16 or x.shape[2] * x.shape[3] > self.block_max_pixels or x.shape[0] < 2:
            return False
        from .._block import BlockCache
        cache = self.__dict__.setdefault("_blk", {}).get(i)
        if cache is None:
            cache = self._blk[i] = BlockCache(f"head level {i}")

        def chain(t):
            self._towers(i, t, raw)
            return [raw]

        return cache.run(chain, [x], [raw]) is not None

    @staticmethod
    def _dw_pw(blk, t):
        """Sequential(DWConv 3x3 (+BN+SiLU), Conv 1x1 (+BN+SiLU)) of the non-legacy cls tower (reference head.py:68-69) as
        ONE fused depthwise->pointwise kernel; anything else runs module by module."""
        if isinstance(blk, nn.Sequential) and len(blk) == 2 and isinstance(blk[0], DWConv) and isinstance(blk[1], Conv):
            dw, pw = blk[0].conv, blk[1].conv
            k = dw.kernel_size[0]
            if (dw.groups == dw.in_channels == dw.out_channels and dw.stride == (1, 1) and k in (3, 5, 7) and dw.padding == (k // 2, k // 2)
                    and pw.kernel_size == (1, 1) and pw.groups == 1 and isinstance(blk[0].act, nn.SiLU) and isinstance(blk[1].act, nn.SiLU)):
                y = ops.dsconv(blk[1], t, blk[0].folded, blk[1].folded, k, L.ACT_SILU, dw_act=L.ACT_SILU)
                if y is not None:
                    return y
        return blk(t)

    def _quality_params(self, i, device):
        return None

    def forward(self, x, nms=None):
        """nms=None: returns (pred (B,4+nc,A) fp32, raw per-level maps) like the reference.  nms=dict(conf=, classes=None, keep_pred=False)
        (predict pipelines): the decode also builds the NMS candidates for that confidence threshold / class filter in the same pass
        and returns (_ops.Candidates, raw maps) -- `utils.ops.nms_device` takes it in place of `pred`, which is then neither written nor
        re-read (keep_pred=True still writes it, as Candidates.pred)."""
        if self.training:
            raise RuntimeError("edge-yolo_amd implements the inference forward only: call model.eval()")
        if self.reg_max != 16:
            raise NotImplementedError("the decode kernel is built for reg_max=16")
        xs = [L.as_nhwc(t) for t in x]
        if self.end2end:
            return self._forward_end2end(x, xs)
        B, dev, dt = xs[0].shape[0], xs[0].device, xs[0].dtype
        A = sum(t.shape[2] * t.shape[3] for t in xs)
        want_pred = nms is None or nms.get("keep_pred") or not (len(xs) <= 4 and self.nc > 1)
        pred = torch.empty((B, 4 + self.nc, A), dtype=torch.float32, device=dev) if want_pred else None
        a_off = 0
        if getattr(self, "_stride_f", None) is None:
            self._stride_f = [float(s) for s in self.stride]  # host copy once (no D2H inside a captured graph)
        levels = []
        # The towers of the pyramid levels are independent: the small levels run on a second HIP stream (fork / join around them),
        # which a captured hipGraph records as a parallel branch -- the 20x20 and 40x40 towers (a few hundred workgroups per kernel)
        # then run beside the 80x80 ones instead of after them (2.18 -> 2.08 ms/step).  More branches are slower (3 level streams
        # 2.38 ms, 6 tower streams 2.40 ms: the persistent kernels fight for the CUs).  `head_streams = False` keeps one stream.
        fork = self.head_streams and dev.type == "cuda" and len(xs) > 1
        cur = torch.cuda.current_stream(dev) if fork else None
        # default: the first (largest) level on the caller's stream, all smaller levels one after the other on ONE side stream
        smap = list(self.tower_streams) if (self.tower_streams and len(self.tower_streams) == 2 * len(xs)) else [0, 0] + [1] * (2 * len(xs) - 2)
        if fork and (getattr(self, "_side", None) is None or self._side_dev != dev or len(self._side) != max(smap)):
            self._side, self._side_dev = [torch.cuda.Stream(device=dev) for _ in range(max(smap))], dev
        # every buffer a side stream WRITES is allocated before the fork: a block handed out later on the caller's stream could be one
        # that kernels still queued there are using (the caching allocator orders reuse per stream only)
        raws = [L.empty_nhwc(B, (self.no + 7) // 8 * 8, t.shape[2], t.shape[3], dt, dev)[:, :self.no] for t in xs]  # pixel stride kept 16-byte aligned for any nc
        if fork:
            for side in self._side:
                side.wait_stream(cur)
        task = 0
        for i, t in enumerate(xs):
            H, W = t.shape[2:]
            raw = raws[i]
            if smap[task] == smap[task + 1]:  # both towers on one stream: small levels go through one block program
                if fork and smap[task] > 0:
                    with torch.cuda.stream(self._side[smap[task] - 1]):
                        done = self._towers_block(i, t, raw)
                else:
                    done = self._towers_block(i, t, raw)
                if done:
                    task += 2
                    levels.append((raw[:, :4 * self.reg_max], raw[:, 4 * self.reg_max:], self._stride_f[i], self._quality_params(i, dev), a_off))
                    a_off += H * W
                    x[i] = raw
                    continue
            for tower in (self._box_tower, self._cls_tower):
                if fork and smap[task] > 0:
                    with torch.cuda.stream(self._side[smap[task] - 1]):
                        tower(i, t, raw)
                else:
                    tower(i, t, raw)
                task += 1
            levels.append((raw[:, :4 * self.reg_max], raw[:, 4 * self.reg_max:], self._stride_f[i], self._quality_params(i, dev), a_off))
            a_off += H * W
            x[i] = raw
        if fork:
            for side in self._side:
                cur.wait_stream(side)
        if getattr(self, "defer_decode", False):
            # two-stage pipelines (engine/predictor.py::PipelinedRunner) put the decode into the post-processing stage, next to the
            # NMS: the next batch's backbone then starts as soon as the towers are done instead of behind the decode launch
            return levels, x
        if nms is not None and len(levels) <= 4 and self.nc > 1:
            classes = nms.get("classes")
            mask = None
            if classes is not None:
                key = (tuple(classes), dev)
                if getattr(self, "_mask_key", None) != key:  # tiny host->device upload, once per filter (not inside a captured graph)
                    m = torch.zeros(self.nc, dtype=torch.uint8)
                    m[torch.as_tensor(list(classes), dtype=torch.long)] = 1
                    self._mask, self._mask_key = m.to(dev), key
                mask = self._mask
            cand = ops.head_decode_levels(levels, pred if nms.get("keep_pred") else None, nms=(nms["conf"], mask, classes))
            return cand, x
        self._decode(levels, pred)
        return pred if self.export else (pred, x)

    # ---- end2end (NMS-free) inference, reference head.py:93-115 (Detect) / :273-298 (GF2Detect)
    one2many_in_inference = False  # the reference also runs the one2many towers in eval mode and returns their maps next to the result,
    #                                where nothing reads them; True reproduces that ("one2many" is None otherwise)

    def _branch_maps(self, xs, branch):
        B, dev, dt = xs[0].shape[0], xs[0].device, xs[0].dtype
        self.__dict__["_branch"] = branch
        try:
            raws, levels, a_off = [], [], 0
            for i, t in enumerate(xs):
                raw = L.empty_nhwc(B, (self.no + 7) // 8 * 8, t.shape[2], t.shape[3], dt, dev)[:, :self.no]
                self._towers(i, t, raw)
                levels.append((raw[:, :4 * self.reg_max], raw[:, 4 * self.reg_max:], self._stride_f[i], self._quality_params(i, dev), a_off))
                a_off += t.shape[2] * t.shape[3]
                raws.append(raw)
        finally:
            self.__dict__.pop("_branch", None)
        return raws, levels, a_off

    def _forward_end2end(self, x, xs):
        """-> (y (B, min(max_det, A), 6) fp32 rows [x1,y1,x2,y2,score,class], {"one2many": maps or None, "one2one": maps}): the one2one
        towers, the decode with x1y1x2y2 boxes (decode_bboxes: xywh and not end2end, head.py:163-165) and Detect.postprocess (:167-189)
        as the top-k selection kernel -- no NMS follows (utils/ops.py:224-228 only filters these rows)."""
        if len(xs) > 4:
            raise NotImplementedError("end2end heads: at most 4 pyramid levels")
        if getattr(self, "_stride_f", None) is None:
            self._stride_f = [float(s) for s in self.stride]
        raws, levels, A = self._branch_maps(xs, "one2one_")
        pred = torch.empty((xs[0].shape[0], 4 + self.nc, A), dtype=torch.float32, device=xs[0].device)
        ops.head_decode_levels(levels, pred, xyxy=True)
        y = ops.e2e_topk(pred, min(self.max_det, A))
        many = self._branch_maps(xs, "")[0] if self.one2many_in_inference else None
        return y if self.export else (y, {"one2many": many, "one2one": raws})

    def _decode(self, levels, pred):
        # every level is decoded by ONE launch (reference: Detect._inference runs after all towers, head.py:84-90,117-148)
        if len(levels) <= 4:
            ops.head_decode_levels(levels, pred)
        else:
            for box, cls, st, q, off in levels:
                ops.head_decode(box, cls, st, q, pred, off)
        return pred

    def decode(self, levels):
        """second half of forward() for callers that deferred it (`defer_decode`): levels -> pred (B, 4+nc, A) fp32."""
        box0 = levels[0][0]
        A = sum(lv[0].shape[2] * lv[0].shape[3] for lv in levels)
        pred = torch.empty((box0.shape[0], 4 + self.nc, A), dtype=torch.float32, device=box0.device)
        return self._decode(levels, pred)

    def bias_init(self):
        """reference head.py:150-161 (needs self.stride)."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / float(s)) ** 2)
        if self.end2end:  # reference :158-161
            for a, b, s in zip(self.one2one_cv2, self.one2one_cv3, self.stride):
                a[-1].bias.data[:] = 1.0
                b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / float(s)) ** 2)
        self._reset_caches()


class GF2Detect(Detect):
    """GFLv2-style quality head on top of Detect (reference head.py:194-345): q = DGQP(softmax(box logits)),
    score = sigmoid(cls) * clamp(q, 1e-6, 1-1e-6)."""

    def __init__(self, nc=80, ch=()):
        super().__init__(nc, ch)
        self.reg_topk = 4
        self.add_mean = True
        self.reg_channels = 64
        self.apply_quality_in_inference = True
        in_stat = 4 * (self.reg_topk + (1 if self.add_mean else 0))
        self.reg_conf = nn.ModuleList(
            nn.Sequential(nn.Conv2d(in_stat, self.reg_channels, 1, bias=True), nn.ReLU(inplace=True), nn.Conv2d(self.reg_channels, 1, 1, bias=True),
                          nn.Sigmoid()) for _ in ch)
        if self.end2end:  # head.py:220-221
            self.one2one_reg_conf = copy.deepcopy(self.reg_conf)
        self._qcache = {}

    def _quality_params(self, i, device):
        if not self.apply_quality_in_inference:
            return None
        if self.reg_topk != 4 or not self.add_mean:
            raise NotImplementedError("the decode kernel is built for reg_topk=4, add_mean=True (the reference defaults)")
        q = self._qcache.get((self._branch, i, device))
        if q is None:
            m = getattr(self, self._branch + "reg_conf")[i]
            f = lambda t: t.detach().float().to(device).contiguous()  # noqa: E731
            q = self._qcache[(self._branch, i, device)] = (f(m[0].weight).view(m[0].out_channels, -1), f(m[0].bias), f(m[2].weight).view(-1), f(m[2].bias))
        return q


class E2EDetect(GF2Detect):
    """NMS-free head (reference head.py:799-824): GF2Detect with end2end = True -- one2one copies of both towers and of the quality
    head (:76-78,220-221), decoded as x1y1x2y2 and reduced to the max_det best rows by Detect.postprocess (:167-189, :273-298).  The
    class tower is the depthwise structure whatever `legacy` says (:812-822)."""

    end2end = True

    def __init__(self, nc=80, ch=()):
        super().__init__(nc, ch)
        c3 = max(ch[0], min(self.nc, 100))
        self.cv3 = nn.ModuleList(
            nn.Sequential(nn.Sequential(DWConv(x, x, 3), Conv(x, c3, 1)), nn.Sequential(DWConv(c3, c3, 3), Conv(c3, c3, 1)), nn.Conv2d(c3, self.nc, 1))
            for x in ch)
        self.one2one_cv3 = copy.deepcopy(self.cv3)


class GFLHeadv2_uniH(GF2Detect):
    """reference head.py:827-908.  stem/dat/pos/cit are nn.Identity placeholders there (use_* default False);
    forward == GF2Detect.forward through the `super()` calls at :898,:907."""

    def __init__(self, nc=80, ch=(), reg_topk=4, add_mean=True, reg_channels=64, use_dat=False, use_cit=False, use_poscnn=False):
        super().__init__(nc, ch)
        if use_dat or use_cit or use_poscnn:
            raise NotImplementedError("use_dat/use_cit/use_poscnn only add nn.Identity placeholders in the reference; not built")
        self.stem = nn.ModuleList(nn.Identity() for _ in ch)
        self.dat = self.pos_cls = self.pos_reg = self.cit_cls = self.cit_reg = None
        self.reg_topk = reg_topk
        self.add_mean = add_mean
        self.reg_channels = reg_channels
        in_stat = 4 * (self.reg_topk + (1 if self.add_mean else 0))
        self.reg_conf = nn.ModuleList(
            nn.Sequential(nn.Conv2d(in_stat, self.reg_channels, 1, bias=True), nn.ReLU(inplace=True), nn.Conv2d(self.reg_channels, 1, 1, bias=True),
                          nn.Sigmoid()) for _ in ch)
