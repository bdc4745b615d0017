"""Composite blocks of the detection path, HIP-backed.  Constructor signatures, attribute names and state_dict
keys follow the reference's ultralytics/nn/modules/block.py (line numbers cited per class).  chunk/split/cat
never copy: producers write straight into channel slices of one NHWC buffer (`out=`), residual adds ride in the
conv epilogue (`res=`).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F  # only for softplus/tanh on 4+1 scalar parameters at pack time

from .conv import Conv, DSConv, _Packed, fold_bn
from .. import _ops as ops
from ... import _lib as L

__all__ = ("DFL", "SPPF", "C2f", "C3", "C3k", "C3k2", "Bottleneck", "Attention", "PSABlock", "C2PSA", "LinearAttention",
           "PSABlock_LinearAttention", "C2PSA_LinearAttention", "DSBottleneck", "DSC3k", "DSC3K2_Wavelet")


def _slot(buf, i, c):
    return buf[:, i * c:(i + 1) * c]


class _Chains(nn.Module):
    """Mixin: pointwise chains of a block -- runs of 1x1 convs whose pixels do not interact -- as ONE launch each (nn/_block.py tiled
    programs: recorded from the ordinary module code below, one 256-thread workgroup per 16-pixel tile walks the stages).  The caches
    hold packed weights, so they are dropped whenever parameters move or reload.  `pw_chains = False` keeps one launch per conv."""

    # True, False, or a tuple of chain names.  Measured on MI355X at batch 32 (C2PSA at 256 channels, 20x20): the four-conv tail as one
    # launch 43 us vs 55 us for four launches; the two-conv head 43 us vs 36 us (its 16 MB of outputs bound it, not launches)
    pw_chains = ("proj_ffn_cv2",)

    def _chain(self, name):
        from .._block import BlockCache
        d = self.__dict__.setdefault("_chain_caches", {})
        c = d.get(name)
        if c is None:
            c = d[name] = BlockCache(f"{type(self).__name__}.{name}", tiled=True)
        return c

    def _chains_on(self, x, name=None):
        on = self.pw_chains if isinstance(self.pw_chains, bool) else (name is None or name in self.pw_chains)
        return bool(on) and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float16 and ops.RECORD is None

    def _apply(self, fn, *a, **k):
        self.__dict__["_chain_caches"] = {}
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self.__dict__["_chain_caches"] = {}
        return super()._load_from_state_dict(*a, **k)


class DFL(nn.Module):
    """Integral of the distribution-focal-loss bins (reference block.py:72-90).  Parameter holder: the expectation is
    computed inside the fused head-decode kernel with the fixed weights 0..c1-1."""

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1


class Bottleneck(nn.Module):
    """reference block.py:467-480."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x, out=None):
        return self.cv2(self.cv1(x), out=out, res=x if self.add else None)


class C2f(nn.Module):
    """reference block.py:357-379."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def forward(self, x, out=None, tail=None):
        """tail (extension): the stride-2 3x3 Conv module that consumes this block's output and nothing else does -- cv2 and that conv then
        run as one kernel where the shape allows (ops.pw_conv3s2), and the TAIL's output is returned."""
        B, _, H, W = x.shape  # x may be a VirtualCat (upsample+concat folded into cv1)
        c, n = self.c, len(self.m)
        buf = L.empty_nhwc(B, (2 + n) * c, H, W, x.dtype, x.device)
        self.cv1(x, out=buf[:, :2 * c])
        for i, m in enumerate(self.m):
            m(_slot(buf, 1 + i, c), out=_slot(buf, 2 + i, c))
        return self.cv2(buf, out=out)


class C3(_Packed):
    """reference block.py:382-396."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=((1, 1), (3, 3)), e=1.0) for _ in range(n)))

    def _cv12(self):
        """cv1 and cv2 are both 1x1 convs over the same input: stack their (BN-folded) filters -> one launch."""
        (w1, b1), (w2, b2) = self.cv1.folded(), self.cv2.folded()
        return torch.cat((w1, w2), 0), torch.cat((b1, b2), 0)

    def forward(self, x, out=None, cv12=None):
        """cv12 (extension): the stacked cv1|cv2 output when the caller has already computed it (chained into the producer of x)."""
        x = L.as_nhwc(x)
        B, _, H, W = x.shape
        c_ = self.cv1.conv.out_channels
        # buf = [cv1(x) -> m(.) in place | cv2(x)]: the bottleneck chain rewrites slot 0 in place (its last conv reads a
        # temporary and adds slot 0 pixel-by-pixel as the residual), so cat() never happens
        buf = cv12 if cv12 is not None else ops.conv2d(self, [x], self._cv12, 1, 1, 0, L.ACT_SILU, tag="cv12")
        slot, t, spare = buf[:, :c_], buf[:, :c_], None
        for m in self.m:
            if isinstance(m, DSBottleneck):
                # a DSBottleneck may run as ONE band kernel whose workgroups read halo rows of their neighbours' input: never in place.
                # The chain alternates between slot 0 and a spare buffer (n = 2: slot 0 -> spare -> slot 0)
                if t.data_ptr() == slot.data_ptr():
                    if spare is None:
                        spare = L.empty_nhwc(B, c_, H, W, x.dtype, x.device)
                    t = m(t, out=spare)
                else:
                    t = m(t, out=slot)
            else:
                t = m(t, out=slot)
        if t.data_ptr() != slot.data_ptr():  # odd DSBottleneck count: the chain ended in the spare buffer -> cv3 reads [spare | cv2(x)] as a virtual concat
            return ops.conv2d(self.cv3, [t, buf[:, c_:]], self.cv3.folded, 1, 1, 0, L.ACT_SILU, out=out)
        return self.cv3(buf, out=out)


class C3k(C3):
    """reference block.py:868-876."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5, k=3):
        super().__init__(c1, c2, n, shortcut, g, e)
        c_ = int(c2 * e)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=(k, k), e=1.0) for _ in range(n)))


class C3k2(C2f):
    """reference block.py:857-865."""

    def __init__(self, c1, c2, n=1, c3k=False, e=0.5, g=1, shortcut=True):
        super().__init__(c1, c2, n, shortcut, g, e)
        self.m = nn.ModuleList(C3k(self.c, self.c, 2, shortcut, g) if c3k else Bottleneck(self.c, self.c, shortcut, g) for _ in range(n))


class SPPF(nn.Module):
    """reference block.py:204-223: cv1 -> three chained 5x5 max-pools -> cv2 over the 4-way concat."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        if k != 5:
            raise NotImplementedError("SPPF pooling kernel is built for k=5 (the only value the YAMLs use)")
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)

    def forward(self, x, out=None, cv12=None):
        """cv12 (extension): the stacked cv1|cv2 output when the caller has already computed it (chained into the producer of x)."""
        x = L.as_nhwc(x)
        B, _, H, W = x.shape
        c_ = self.cv1.conv.out_channels
        buf = L.empty_nhwc(B, 4 * c_, H, W, x.dtype, x.device)
        self.cv1(x, out=buf[:, :c_])
        ops.sppf_pool(_slot(buf, 0, c_), _slot(buf, 1, c_), _slot(buf, 2, c_), _slot(buf, 3, c_))
        return self.cv2(buf, out=out)


# ----------------------------------------------------------------------------------------------- attention
class Attention(_Packed):
    """Softmax self-attention of the YOLO11 baseline (reference block.py:1000-1053)."""

    def __init__(self, dim, num_heads=8, attn_ratio=0.5):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.key_dim = int(self.head_dim * attn_ratio)
        self.scale = self.key_dim ** -0.5
        nh_kd = self.key_dim * num_heads
        h = dim + nh_kd * 2
        self.qkv = Conv(dim, h, 1, act=False)
        self.proj = Conv(dim, dim, 1, act=False)
        self.pe = Conv(dim, dim, 3, 1, g=dim, act=False)

    def _proj2(self):
        # proj(a + p) == conv over the virtual concat [a | p] with the weight repeated: one launch, no add kernel
        w, b = self.proj.folded()
        return torch.cat((w, w), 1), b

    def forward(self, x, out=None, res=None):
        x = L.as_nhwc(x)
        B, C, H, W = x.shape
        nh, kd, hd = self.num_heads, self.key_dim, self.head_dim
        qkv = self.qkv(x)
        a = ops.softmax_attention(qkv, nh, kd, hd, self.scale)
        per = 2 * kd + hd
        v = L.empty_nhwc(B, C, H, W, x.dtype, x.device)
        for h in range(nh):  # v.reshape(B, C, H, W): gather the per-head value channels
            ops.copy_slice(qkv[:, h * per + 2 * kd:(h + 1) * per], v[:, h * hd:(h + 1) * hd])
        p = self.pe(v)
        return ops.conv2d(self, [a, p], self._proj2, 1, 1, 0, L.ACT_NONE, out=out, res=res, tag="proj2")


class PSABlock(nn.Module):
    """x + Attention(x); x + FFN(x) (reference block.py:3376-3408)."""

    def __init__(self, c, attn_ratio=0.5, num_heads=None, mlp_ratio=2.0, qkv_bias=True, proj_bias=False, **kwargs):
        super().__init__()
        heads = max(1, (c // 64) if num_heads is None else int(num_heads))
        assert c % heads == 0, f"PSABlock: channels {c} must be divisible by num_heads {heads}"
        self.attn = Attention(c, num_heads=heads, attn_ratio=attn_ratio)
        hidden = int(c * mlp_ratio)
        self.ffn = nn.Sequential(Conv(c, hidden, k=1, s=1, act=True), Conv(hidden, c, k=1, s=1, act=False))

    def forward(self, x, out=None):
        x = self.attn(x, res=x)
        return self.ffn[1](self.ffn[0](x), out=out, res=x)


class C2PSA(nn.Module):
    """reference block.py:1100-1139."""

    def __init__(self, c1, c2, n=1, e=0.5):
        super().__init__()
        assert c1 == c2
        self.c = int(c1 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv(2 * self.c, c1, 1)
        self.m = nn.Sequential(*(PSABlock(self.c, attn_ratio=0.5, num_heads=self.c // 64) for _ in range(n)))

    def forward(self, x, out=None):
        t = self.cv1(x)
        b = t[:, self.c:]
        for i, m in enumerate(self.m):
            b = m(b, out=t[:, self.c:] if i == len(self.m) - 1 else None)
        return self.cv2(t, out=out)


class LinearAttention(_Packed):
    """reference block.py:3348-3373: qkv 1x1 (bias) -> k softmax over head_dim, q softmax over N -> ctx = k^T v ->
    y = q ctx -> proj 1x1.  Extra kwargs are swallowed like the reference's **kwargs."""

    def __init__(self, dim, num_heads, attn_ratio=None, qkv_bias=False, proj_bias=True, **kwargs):
        super().__init__()
        assert dim % num_heads == 0, "LinearAttention: dim must be divisible by num_heads"
        self.dim = dim
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.qkv = nn.Conv2d(dim, 3 * dim, kernel_size=1, bias=qkv_bias)
        self.proj = nn.Conv2d(dim, dim, kernel_size=1, bias=proj_bias)

    def forward(self, x, out=None, res=None):
        qkv = ops.conv2d(self, [x], lambda: fold_bn(self.qkv.weight, self.qkv.bias, None), 1, 1, 0, L.ACT_NONE, tag="qkv")
        y = ops.linear_attention(qkv, self.num_heads)
        bias_fn = (lambda: (self.proj.weight.detach().float(), self.proj.bias.detach().float() if self.proj.bias is not None else None))
        return ops.conv2d(self, [y], bias_fn, 1, 1, 0, L.ACT_NONE, out=out, res=res, tag="proj")


class PSABlock_LinearAttention(nn.Module):
    """reference block.py:3412-3449."""

    def __init__(self, dim, attn_ratio=0.5, num_heads=None, mlp_ratio=2.0, qkv_bias=True, proj_bias=False, fmap="elu", eps=1e-6):
        super().__init__()
        self.attn = LinearAttention(dim=dim, num_heads=num_heads, attn_ratio=attn_ratio, qkv_bias=qkv_bias, proj_bias=proj_bias, fmap=fmap, eps=eps)
        hidden = int(dim * mlp_ratio)
        self.ffn = nn.Sequential(Conv(dim, hidden, k=1, s=1, act=True), Conv(hidden, dim, k=1, s=1, act=False))

    def forward(self, x, out=None):
        x = self.attn(x, res=x)
        return self.ffn[1](self.ffn[0](x), out=out, res=x)


class C2PSA_LinearAttention(_Chains):
    """reference block.py:3452-3497.  f16: 7 launches -> 3: [cv1 -> qkv] | linear attention | [proj(+x) -> ffn -> ffn(+x) -> cv2]."""

    def __init__(self, c1, c2, n=1, e=0.5, attn_ratio=0.5, num_heads=None, mlp_ratio=2.0, fmap="elu"):
        super().__init__()
        assert c1 == c2, "C2PSA_LinearAttention requires c1 == c2"
        self.c = int(c1 * e)
        heads = max(1, (self.c // 64) if num_heads is None else num_heads)
        assert self.c % heads == 0
        self.cv1 = Conv(c1, 2 * self.c, k=1, s=1)
        self.m = nn.Sequential(*[PSABlock_LinearAttention(dim=self.c, attn_ratio=attn_ratio, num_heads=heads, mlp_ratio=mlp_ratio, fmap=fmap)
                                 for _ in range(n)])
        self.cv2 = Conv(2 * self.c, c1, k=1, s=1)

    def forward(self, x, out=None):
        if len(self.m) == 1 and self._chains_on(x):
            y = self._forward_chained(x, out)
            if y is not None:
                return y
        t = self.cv1(x)
        b = t[:, self.c:]
        for i, m in enumerate(self.m):
            b = m(b, out=t[:, self.c:] if i == len(self.m) - 1 else None)
        return self.cv2(t, out=out)

    def _forward_chained(self, x, out):
        blk, c = self.m[0], self.c
        at = blk.attn
        qkv_w = lambda: fold_bn(at.qkv.weight, at.qkv.bias, None)  # noqa: E731
        proj_w = lambda: (at.proj.weight.detach().float(), at.proj.bias.detach().float() if at.proj.bias is not None else None)  # noqa: E731

        def head(x0):  # cv1 -> qkv of the attention branch
            t = self.cv1(x0)
            return [t, ops.conv2d(at, [t[:, c:]], qkv_w, 1, 1, 0, L.ACT_NONE, tag="qkv")]

        def tail(y, t):  # x1 = b + proj(y); x2 = x1 + ffn(x1) (written over b, in place per pixel); cv2([a | x2])
            b = t[:, c:]
            x1 = ops.conv2d(at, [y], proj_w, 1, 1, 0, L.ACT_NONE, res=b, tag="proj")
            blk.ffn[1](blk.ffn[0](x1), out=b, res=x1)
            return [self.cv2(t, out=out)]

        got = self._chain("cv1_qkv").run(head, [L.as_nhwc(x)]) if self._chains_on(x, "cv1_qkv") else None
        t, qkv = got if got is not None else head(L.as_nhwc(x))
        y = ops.linear_attention(qkv, at.num_heads)
        res = self._chain("proj_ffn_cv2").run(tail, [y, t], [out] if out is not None else None) if self._chains_on(x, "proj_ffn_cv2") else None
        if res is None:  # (not block-executable: the same tail, one launch per conv)
            return tail(y, t)[0]
        return res[0]


# ----------------------------------------------------------------------------------------------- DS / wavelet
class DSBottleneck(nn.Module):
    """reference block.py:1467-1503."""

    def __init__(self, c1, c2, shortcut=True, e=0.5, k1=3, k2=5, d2=1):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = DSConv(c1, c_, k1, s=1, p=None, d=1)
        self.cv2 = DSConv(c_, c2, k2, s=1, p=None, d=d2)
        self.add = shortcut and c1 == c2

    def forward(self, x, out=None):
        y = ops.dsb_pair(self.cv1, self.cv2, x, self.add, out=out)  # both DSConvs + the residual as one band kernel on the small maps
        if y is not None:
            return y
        return self.cv2(self.cv1(x), out=out, res=x if self.add else None)


class DSC3k(C3):
    """reference block.py:1506-1562."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5, k1=3, k2=5, d2=1):
        super().__init__(c1, c2, n, shortcut, g, e)
        c_ = int(c2 * e)
        self.m = nn.Sequential(*(DSBottleneck(c_, c_, shortcut=shortcut, e=1.0, k1=k1, k2=k2, d2=d2) for _ in range(n)))


class _PywtDWT2D(nn.Module):
    """Single-level 2-D Haar analysis (reference block.py:3582-3642).  The pywt filter bank is the constant
    1/sqrt(2) pair; the kernel applies the squared taps exactly as the reference's depthwise conv does."""

    def __init__(self, wave="haar", mode="symmetric"):
        super().__init__()
        if wave not in ("haar", "db1"):
            raise NotImplementedError(f"wavelet '{wave}': only the Haar bank is on the detection path")
        self.wave_name, self.mode = wave, mode

    def forward(self, x):
        y = ops.dwt_haar(x)
        c = x.shape[1]
        return y[:, :c], y[:, c:2 * c], y[:, 2 * c:3 * c], y[:, 3 * c:]


class _WaveletEnhancer(_Packed):
    """reference block.py:3645-3710:  b + tanh(gamma) * fuse(cat[b, w0 up(f_ll LL), w1 up(f_h LH), w2 up(f_h HL), w3 up(f_h HH)]).

    A 1x1 conv commutes with bilinear upsampling, so the 3c-channel concat never exists:
        fuse(cat[...]) = SiLU( W_b b + up2x( sum_i w_i W_i P_i ) + bias ),
    i.e. a half-resolution 1x1 conv Z over the four processed sub-bands P (2c -> c, sub-band weights w_i and the BN
    scale folded into its weights) whose bilinear x2 upsample is added before the activation in the epilogue of the
    full-resolution 1x1 conv over b, which also applies tanh(gamma) and the residual b."""

    fused_z = True  # f16: the half-resolution branch as one kernel (ey_wavelet_z); False = dwt + 4-group conv + 1x1 conv (three launches)

    def __init__(self, c, use_ds=False, alpha0=(0.5, 0.2, 0.2, 0.1), wave="haar", mode="symmetric"):
        super().__init__()
        self.c = c
        self.dwt = _PywtDWT2D(wave=wave, mode=mode)
        self.f_ll = Conv(c, c // 2, k=1, s=1)
        self.f_h = (DSConv if use_ds else Conv)(c, c // 2, k=3, s=1)
        self.fuse = Conv(3 * c, c, k=1, s=1)
        self.alpha = nn.Parameter(torch.tensor(alpha0, dtype=torch.float32))
        self.gamma = nn.Parameter(torch.tensor(0.0))

    def _band_weights(self):
        w = F.softplus(self.alpha.detach().float())
        return w / (w.sum() + 1e-6)  # block.py:3697-3698

    def _subband_sets(self):
        wl, bl = self.f_ll.folded()
        wh, bh = self.f_h.folded()
        w3 = torch.zeros_like(wh)
        w3[:, :, 1:2, 1:2] = wl  # 1x1 == 3x3 with only the centre tap (pad 1)
        return (w3, bl), (wh, bh)

    def _fuse_b(self):
        w, b = self.fuse.folded()
        return w[:, :self.c].contiguous(), b

    def _fuse_z(self):
        w, _ = self.fuse.folded()
        bw = self._band_weights().to(w.device)
        h = self.c // 2
        wz = w[:, self.c:].clone()
        for i in range(4):
            wz[:, i * h:(i + 1) * h] *= bw[i]
        return wz.contiguous(), None

    def forward(self, b, out=None, then=None):
        """then (extension): dict(mod=, folded_fn=, act=, tag=) of a 1x1 conv that reads this module's output -- where the shapes allow both run as
        one launch (ops.conv_pw_pair) and (y, y_then) is returned; y_then is None when the caller has to run that conv itself."""
        b = L.as_nhwc(b)
        B, c, H, W = b.shape
        if H < 2 or W < 2:
            raise ValueError(f"_WaveletEnhancer: feature map {H}x{W} is too small for a 2x2 Haar step")
        if isinstance(self.f_h, DSConv):
            raise NotImplementedError("use_ds=True sub-band path is not built (no YAML enables it)")
        h = c // 2
        g = self._packed("tanh_gamma", lambda: float(torch.tanh(self.gamma.detach().float())))  # host scalar, cached (graph capture)
        if self.fused_z:  # f16: DWT + the four sub-band convs + Z in ONE kernel, only Z touches HBM
            Z = ops.wavelet_z(self, b, self._subband_sets, self._fuse_z)
            if Z is not None:
                if then is not None:
                    return ops.conv_pw_pair(dict(mod=self, srcs=[b], folded_fn=self._fuse_b, act=L.ACT_SILU, out=out, res=b, addz=Z, out_scale=g, tag="b"), then)
                return ops.conv2d(self, [b], self._fuse_b, 1, 1, 0, L.ACT_SILU, out=out, res=b, addz=Z, out_scale=g, tag="b")
        sub = ops.dwt_haar(b)  # (B,4c,H/2,W/2): LL|LH|HL|HH
        P = L.empty_nhwc(B, 2 * c, H // 2, W // 2, b.dtype, b.device)
        # ONE launch for the four sub-band convs: group 0 = f_ll (a 1x1 conv written as a centre-tap 3x3) on LL, groups 1-3 =
        # the shared f_h on LH, HL, HH (weight set min(g, 1)); the groups are channel-offset slices of `sub` and `P`
        ops.conv2d(self, [sub[:, :c]], self._subband_sets, 3, 1, 1, L.ACT_SILU, out=P[:, :h], ngroup=4, src_gstride=c, y_gstride=h, w_sets=2,
                   tag="sub")
        Z = ops.conv2d(self, [P], self._fuse_z, 1, 1, 0, L.ACT_NONE, tag="z")
        y = ops.conv2d(self, [b], self._fuse_b, 1, 1, 0, L.ACT_SILU, out=out, res=b, addz=Z, out_scale=g, tag="b")
        return (y, None) if then is not None else y


class DSC3K2_Wavelet(nn.Module):
    """reference block.py:3749-3788: cv1 -> chunk(a, b) -> b = wave(b) -> n x (DSC3k | DSBottleneck) -> cat -> cv2."""

    def __init__(self, c1, c2, n=1, dsc3k=False, e=0.5, g=1, shortcut=True, k1=3, k2=7, d2=1, **kwargs):
        super().__init__()
        use_ds = bool(kwargs.get("use_ds", False))
        wave = kwargs.get("wave", "haar")
        mode = kwargs.get("mode", "symmetric")
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1, 1)
        if dsc3k:
            self.m = nn.ModuleList(DSC3k(self.c, self.c, n=2, shortcut=shortcut, g=g) for _ in range(n))
        else:
            self.m = nn.ModuleList(DSBottleneck(self.c, self.c, shortcut=shortcut, e=1.0, k1=k1, k2=k2, d2=d2) for _ in range(n))
        self.wave = _WaveletEnhancer(self.c, use_ds=use_ds, wave=wave, mode=mode)

    def forward(self, x, out=None, tail=None):
        """tail (extension): the stride-2 3x3 Conv module that consumes this block's output and nothing else does -- cv2 and that conv then
        run as one kernel where the shape allows (ops.pw_conv3s2), and the TAIL's output is returned."""
        B, _, H, W = x.shape  # x may be a VirtualCat (upsample+concat folded into cv1)
        c, n = self.c, len(self.m)
        t = self.cv1(x)  # [a | b]
        buf = L.empty_nhwc(B, (1 + n) * c, H, W, x.dtype, x.device)  # [wave(b) | m_0 | ...]
        pre = None
        if n and isinstance(self.m[0], C3):  # the DSC3k behind the enhancer starts with a 1x1 over the enhancer's output: chained into its tail conv
            _, pre = self.wave(t[:, c:], out=_slot(buf, 0, c), then=dict(mod=self.m[0], folded_fn=self.m[0]._cv12, act=L.ACT_SILU, tag="cv12"))
        else:
            self.wave(t[:, c:], out=_slot(buf, 0, c))
        for i, m in enumerate(self.m):
            if i == 0 and pre is not None:
                m(_slot(buf, i, c), out=_slot(buf, 1 + i, c), cv12=pre)
            else:
                m(_slot(buf, i, c), out=_slot(buf, 1 + i, c))
        # cat(a, wave(b), m...) is never built: cv2 reads the two buffers as one virtual concat
        if tail is not None:
            y = ops.pw_conv3s2(self.cv2, tail, [t[:, :c], buf])
            return y if y is not None else tail(ops.conv2d(self.cv2, [t[:, :c], buf], self.cv2.folded, 1, 1, 0, L.ACT_SILU))
        return ops.conv2d(self.cv2, [t[:, :c], buf], self.cv2.folded, 1, 1, 0, L.ACT_SILU, out=out)
