"""Convolution modules of the detection path, HIP-backed.

Same constructor signatures, attribute names and state_dict keys as the reference's
ultralytics/nn/modules/conv.py (`Conv` :41-59, `DSConv` :87-104, `DWConv` :124-129, `Concat` :345-355,
`autopad` :32-38), so a reference state_dict loads unchanged.  `nn.Conv2d` / `nn.BatchNorm2d` children are
parameter containers only; they are never called.  forward() packs (BN-folded) weights for the HIP kernels on
first use and launches through the C ABI (`_lib`).  Extra keyword arguments of forward (`out=`, `res=`) let the
composite blocks write straight into concat buffers and fuse residual adds; positional use is drop-in.
"""
import ctypes
import math

import torch
import torch.nn as nn

from .. import _ops as ops
from ... import _lib as L

__all__ = ("Conv", "DWConv", "DSConv", "Concat", "Upsample", "autopad")


def autopad(k, p=None, d=1):
    """'same' padding (reference conv.py:32-38)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def _act_code(act):
    if isinstance(act, nn.SiLU):
        return L.ACT_SILU
    if isinstance(act, nn.ReLU):
        return L.ACT_RELU
    if isinstance(act, nn.Sigmoid):
        return L.ACT_SIGMOID
    if isinstance(act, nn.Identity):
        return L.ACT_NONE
    raise NotImplementedError(f"activation {type(act).__name__} has no HIP epilogue (SiLU/ReLU/Sigmoid/Identity do)")


class _Packed(nn.Module):
    """Mixin: per-(dtype, device) cache of packed device weights; dropped whenever parameters move or reload."""

    def __init__(self):
        super().__init__()
        self._cache = {}
        self.register_load_state_dict_post_hook(lambda m, _keys: m._cache.clear())

    def _apply(self, fn, *a, **k):
        self._cache = {}
        return super()._apply(fn, *a, **k)

    def _packed(self, key, build):
        v = self._cache.get(key)
        if v is None:
            with torch.no_grad():
                v = self._cache[key] = build()
        return v


def fold_bn(w, b, bn):
    """fuse_conv_and_bn (reference utils/torch_utils.py:238-265) on fp32 copies: returns (w', b')."""
    w = w.detach().float()
    b = torch.zeros(w.shape[0], device=w.device) if b is None else b.detach().float()
    if bn is None:
        return w, b
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    return w * s.view(-1, 1, 1, 1), s * b + (bn.bias.detach().float() - bn.running_mean.detach().float() * s)


class Conv(_Packed):
    """Conv2d + BatchNorm + activation (reference conv.py:41-59): args (c1, c2, k, s, p, g, d, act)."""

    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()

    # -- fuse(): fold BN into the conv parameters like BaseModel.fuse does (tasks.py:214-242)
    def fuse_bn(self):
        if hasattr(self, "bn"):
            w, b = fold_bn(self.conv.weight, self.conv.bias, self.bn)
            dt = self.conv.weight.dtype
            self.conv.weight = nn.Parameter(w.to(dt), requires_grad=False)
            self.conv.bias = nn.Parameter(b.to(dt), requires_grad=False)
            del self.bn
            self._cache = {}

    def folded(self):
        return fold_bn(self.conv.weight, self.conv.bias, getattr(self, "bn", None))

    def forward(self, x, out=None, res=None):
        c = self.conv
        w, b = None, None
        k, s, p, d, g = c.kernel_size[0], c.stride[0], c.padding[0], c.dilation[0], c.groups
        if c.kernel_size[0] != c.kernel_size[1] or c.stride[0] != c.stride[1] or d != 1:
            raise NotImplementedError("non-square kernels / dilation are outside the EdgeLine-YOLO detection path")
        act = _act_code(self.act)
        if (g == 1 and k == 3 and s == 2 and p == 1 and c.in_channels <= 4 and c.out_channels % 16 == 0 and res is None
                and torch.is_tensor(x) and x.dim() == 4 and x.is_contiguous() and not L.is_nhwc_view(x)):
            # network stem: read the planar NCHW image directly, write NHWC
            return ops.stem_conv(self, x, self.folded, act, x.dtype, out=out)
        if g == 1:
            return ops.conv2d(self, [x], self.folded, k, s, p, act, out=out, res=res)
        if g == c.in_channels == c.out_channels and s == 1 and k in (3, 5, 7) and p == k // 2 and res is None:
            return ops.dwconv(self, x, self.folded, k, act, out=out)
        return ops.conv2d_direct(self, x, self.folded, k, s, p, g, act, out=out, res=res)

    forward_fuse = forward


class DWConv(Conv):
    """Depth-wise convolution (reference conv.py:124-129)."""

    def __init__(self, c1, c2, k=1, s=1, d=1, act=True):
        super().__init__(c1, c2, k, s, g=math.gcd(c1, c2), d=d, act=act)


class DSConv(_Packed):
    """Depthwise-separable conv: dw kxk -> pw 1x1 -> BN -> SiLU (reference conv.py:87-104).  BaseModel.fuse does
    not touch it (tasks.py:224), so its BatchNorm stays a module; it is folded into the packed pw weights here."""

    def __init__(self, c_in, c_out, k=3, s=1, p=None, d=1, bias=False):
        super().__init__()
        if p is None:
            p = (d * (k - 1)) // 2
        self.dw = nn.Conv2d(c_in, c_in, kernel_size=k, stride=s, padding=p, dilation=d, groups=c_in, bias=bias)
        self.pw = nn.Conv2d(c_in, c_out, 1, 1, 0, bias=bias)
        self.bn = nn.BatchNorm2d(c_out)
        self.act = nn.SiLU()

    def _dw_folded(self):
        return fold_bn(self.dw.weight, None, None)[0], (self.dw.bias.detach().float() if self.dw.bias is not None else None)

    def _pw_folded(self):
        return fold_bn(self.pw.weight, self.pw.bias, self.bn)

    def forward(self, x, out=None, res=None):
        k, s, d = self.dw.kernel_size[0], self.dw.stride[0], self.dw.dilation[0]
        if s != 1 or d != 1 or self.dw.padding[0] != k // 2:
            raise NotImplementedError("DSConv with stride/dilation != 1 is outside the EdgeLine-YOLO detection path")
        y = ops.dsconv(self, x, self._dw_folded, self._pw_folded, k, L.ACT_SILU, out=out, res=res)  # one fused kernel
        if y is not None:
            return y
        t = ops.dwconv(self, x, self._dw_folded, k, L.ACT_NONE, tag="dw")
        return ops.conv2d(self, [t], self._pw_folded, 1, 1, 0, L.ACT_SILU, out=out, res=res, tag="pw")


class Concat(nn.Module):
    """Channel concat (reference conv.py:345-355).  Stand-alone it copies slices on the device and returns a tensor;
    inside a DetectionModel whose consumer starts with a 1x1 conv (`lazy=True`, set by the graph builder) it returns a
    VirtualCat that the conv reads in place."""

    lazy = False

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        if self.d != 1:
            raise NotImplementedError("Concat along dim != 1")
        if self.lazy:
            parts = []
            for t in x:
                parts += t.parts if isinstance(t, ops.VirtualCat) else [(t, 0)]
            if len(parts) <= 2:
                return ops.VirtualCat(parts)
        return ops.concat(x)


class Upsample(nn.Module):
    """nn.Upsample(None, 2, 'nearest') of the YAMLs (layers 11, 14)."""

    def __init__(self, size=None, scale_factor=None, mode="nearest"):
        super().__init__()
        if size is not None or int(scale_factor) != 2 or mode != "nearest":
            raise NotImplementedError("only nearest x2 upsampling is on the detection path")
        self.scale_factor, self.mode = scale_factor, mode

    lazy = False

    def forward(self, x):
        if self.lazy and torch.is_tensor(x):
            return ops.VirtualCat([(x, 1)])
        return ops.upsample2x(x)
