"""Tensor-level wrappers over the C ABI: take logical-NCHW torch tensors (NHWC memory), hand plain pointers /
strides to libedgeyolo_hip.so on the current HIP stream.  No arithmetic happens in Python or in torch here."""
import ctypes

import torch

from .. import _lib as L

TRACE = None  # set by edge-yolo_amd/profiling.py: per-launch HIP-event timing + algorithmic bytes/FLOPs
RECORD = None  # set by nn/_block.py while a block program is being recorded: wrappers append a stage instead of launching


def _no_block(what):
    if RECORD is not None:
        from ._block import BlockUnsupported
        raise BlockUnsupported(what)


class _NoTrace:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NT = _NoTrace()


def _tr(kernel, nbytes, flops=0, note="", kernels=1):
    return TRACE.launch(kernel, nbytes, flops, note, kernels) if TRACE is not None else _NT


def _nb(*tensors):
    """algorithmic bytes: every listed tensor view touched exactly once."""
    return sum(t.numel() * t.element_size() for t in tensors if t is not None)


def _dev_key(x, tag=""):
    return (tag, x.dtype, x.device)


class VirtualCat:
    """Channel concat of NHWC tensors that is never written: `parts` = [(tensor, up)], `up`=1 meaning "read through a
    nearest x2 upsample".  Produced by Upsample / Concat when the graph knows the consumer is a 1x1 conv (layers
    11-12, 14-15, 18, 21 of the YAMLs); the conv kernel walks the sources directly (K7 folded into K1)."""

    def __init__(self, parts):
        self.parts = parts
        t, u = parts[0]
        self.shape = (t.shape[0], sum(p.shape[1] for p, _ in parts), t.shape[2] << u, t.shape[3] << u)
        self.dtype, self.device = t.dtype, t.device
        self.is_cuda = t.is_cuda

    def dim(self):
        return 4

    def materialize(self):
        B, c, H, W = self.shape
        out = L.empty_nhwc(B, c, H, W, self.dtype, self.device)
        c0 = 0
        for t, u in self.parts:
            copy_slice(L.as_nhwc(t), out[:, c0:c0 + t.shape[1]], up=u)
            c0 += t.shape[1]
        return out


def as_tensor(x):
    return x.materialize() if isinstance(x, VirtualCat) else x


def pack_conv_weight(w_oihw, dtype, device):
    """fp32 OIHW -> packed MFMA layout (host side, ey_conv_pack_weight) -> device."""
    w = w_oihw.detach().float().cpu().contiguous()
    co, ci, k, _ = w.shape
    code = L.dtype_code(dtype)
    nbytes = L.lib().ey_conv_packed_bytes(code, co, ci, k)
    buf = torch.empty(nbytes, dtype=torch.uint8)
    L.check(L.lib().ey_conv_pack_weight(code, co, ci, k, w.data_ptr(), buf.data_ptr(), nbytes), "conv_pack_weight")
    return buf.to(device)


def igemm_ok(srcs, k, s, p):
    if k not in (1, 3) or s not in (1, 2) or p != k // 2:
        return False
    for t in srcs:
        es = t.element_size()
        if t.shape[1] % 8 or (L.cstride(t) * es) % 16 or t.data_ptr() % 16:
            return False
    return True


def conv2d(mod, srcs, folded_fn, k, s, p, act, out=None, res=None, tag="", up=None, addz=None, out_scale=1.0,
           ngroup=1, src_gstride=0, y_gstride=0, group_C=None, w_sets=1, _build_only=False):
    """y = res + out_scale*act(conv(cat(srcs)) + bias + up2x(addz)).  srcs: list of 1-2 logical-NCHW tensors
    (source i is read through a nearest x2 upsample when up[i]).  With ngroup>1 `srcs[0]`/`out` are the group-0
    slices and *_gstride the element offsets between groups (channel count per group = group_C)."""
    if len(srcs) == 1 and isinstance(srcs[0], VirtualCat):
        vc = srcs[0]
        if k == 1 and len(vc.parts) <= 2 and up is None:
            srcs, up = [t for t, _ in vc.parts], [u for _, u in vc.parts]
        else:
            srcs = [vc.materialize()]
    x0 = srcs[0]
    L.require_device(x0, "conv2d")
    srcs = [L.as_nhwc(t) for t in srcs]
    up = up or [0] * len(srcs)
    B = srcs[0].shape[0]
    H = srcs[0].shape[2] << up[0]
    W = srcs[0].shape[3] << up[0]
    for t, u in zip(srcs, up):
        if (t.shape[0], t.shape[2] << u, t.shape[3] << u) != (B, H, W) or t.dtype != x0.dtype:
            raise ValueError(f"conv2d: sources disagree: {[tuple(t.shape) for t in srcs]} up={up}")
    cin = sum(t.shape[1] for t in srcs)
    if len(srcs) > 2 or not igemm_ok(srcs, k, s, p):
        if _build_only:
            return None
        _no_block("conv shape outside the MFMA kernel")
        if len(srcs) != 1 or up[0] or addz is not None or ngroup != 1 or out_scale != 1.0:
            raise NotImplementedError("this conv shape needs the generic direct kernel, which takes a single plain source")
        return conv2d_direct(mod, srcs[0], folded_fn, k, s, p, 1, act, out=out, res=res, tag=tag)
    dtype, dev = x0.dtype, x0.device

    def build():
        if w_sets == 1:
            w, b = folded_fn()
            ws, bs = [w], [b]
        else:  # several weight sets (group g uses set min(g, w_sets-1)): packed back to back, biases stacked
            ws, bs = zip(*folded_fn())
        for w in ws:
            if w.shape[1] != cin or w.shape[2] != k or w.shape[0] != ws[0].shape[0]:
                raise ValueError(f"conv2d: weight {tuple(w.shape)} does not match Cin={cin} k={k}")
        packs = [pack_conv_weight(w, dtype, dev) for w in ws]
        bias = None if bs[0] is None else torch.cat([b.to(dev).float() for b in bs]).contiguous()
        return torch.cat(packs), bias, ws[0].shape[0], packs[0].numel() // x0.element_size()

    wp, bias, cout, wset_elems = mod._packed(_dev_key(x0, "igemm" + tag), build)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    if out is None:
        out = L.empty_nhwc(B, cout * (ngroup if y_gstride == 0 and ngroup > 1 else 1), Ho, Wo, dtype, dev)
    elif not L.is_nhwc_view(out) or tuple(out.shape) != (B, cout, Ho, Wo) or out.dtype != dtype:
        raise ValueError(f"conv2d: out= must be an NHWC view of shape {(B, cout, Ho, Wo)} {dtype}, got {tuple(out.shape)} {out.dtype}")
    d = L.ConvDesc()
    d.dtype = L.dtype_code(dtype)
    d.B, d.H, d.W, d.Ho, d.Wo, d.Cout = B, H, W, Ho, Wo, cout
    d.k, d.stride, d.pad, d.act, d.nsrc = k, s, p, act, len(srcs)
    for i, t in enumerate(srcs):
        d.src[i] = t.data_ptr()
        d.src_C[i] = group_C if (group_C and i == 0) else t.shape[1]
        d.src_cstride[i] = L.cstride(t)
        d.src_up[i] = up[i]
    d.w = wp.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.y, d.y_cstride = out.data_ptr(), L.cstride(out)
    if res is not None:
        res = L.as_nhwc(res)
        if tuple(res.shape) != (B, cout, Ho, Wo) or res.dtype != dtype:
            raise ValueError("conv2d: residual shape/dtype mismatch")
        d.res, d.res_cstride = res.data_ptr(), L.cstride(res)
    d.out_scale = float(out_scale)
    if addz is not None:
        addz = L.as_nhwc(addz)
        if tuple(addz.shape[:2]) != (B, cout) or addz.dtype != dtype:
            raise ValueError("conv2d: addz batch/channel/dtype mismatch")
        d.addz, d.addz_cstride, d.addz_H, d.addz_W = addz.data_ptr(), L.cstride(addz), addz.shape[2], addz.shape[3]
    d.ngroup, d.src_gstride, d.y_gstride = ngroup, src_gstride, y_gstride
    d.w_gstride, d.w_gmax = (wset_elems, w_sets - 1) if w_sets > 1 else (0, 0)
    if _build_only:  # (descriptor, output, tensors the descriptor points into)
        return d, out, (srcs, wp, bias, res, addz)
    if RECORD is not None:
        RECORD.conv(d, srcs, up, out, res, addz, wp, bias, 2.0 * ngroup * B * Ho * Wo * cout * cin * k * k, w_sets * cout * cin * k * k * x0.element_size())
        return out
    if TRACE is not None:
        M = B * Ho * Wo
        es = x0.element_size()
        var = L.lib().ey_conv_variant(d.dtype, cout, cin, k, s, int(len(srcs) == 1 and not up[0]), M, ngroup)
        tn = "f16" if es == 2 else "f32"
        name = (f"conv_small_kernel<{tn},{var % 1000 // 10},{var % 10}>" if var >= 3000 else
                f"conv3_halo_kernel<{tn},{var % 1000 // 10},{s}>" if var >= 2000 else
                f"conv_ws_kernel<{tn},{var % 1000 // 10},{var % 10},{k}>" if var >= 1000 else f"conv_igemm_kernel<{tn},{var // 10},{var % 10}>")
        nbytes = ngroup * (_nb(*srcs) + M * cout * es * (2 if res is not None else 1) + _nb(addz)) + cout * cin * k * k * es
        rec = _tr(name, nbytes, 2.0 * ngroup * M * cout * cin * k * k,
                  note=f"{cin}->{cout} k{k}s{s} {H}x{W} g{ngroup}{' +res' if res is not None else ''}{' +addz' if addz is not None else ''}{' 2src' if len(srcs) > 1 else ''}")
        with rec:
            L.check(L.lib().ey_conv2d(ctypes.byref(d), L.stream()), "ey_conv2d")
            lv = L.lib().ey_conv_last_variant()
            if lv >= 9000:
                rec.kernel = f"conv3p_kernel<{lv % 1000 // 10}>"
            elif lv >= 8000:
                rec.kernel = f"conv3s_kernel<{lv % 1000 // 100},{lv % 100 // 10},{lv % 10}>"
            elif lv >= 7000:
                rec.kernel = f"conv3r_kernel<{lv % 1000 // 10},{lv % 10}>"
            elif lv >= 6000:
                rec.kernel = f"conv3_tile_kernel<{tn},{lv % 1000 // 10},{lv % 10}>"
            elif lv >= 5000:
                rec.kernel = f"conv_pwr_kernel<{tn},{lv % 1000 // 10},{lv % 10}>"
            elif lv >= 4000:
                rec.kernel = f"conv_pw_kernel<{tn},{lv % 1000 // 10}>"
            elif lv >= 3000:
                rec.kernel = f"conv_pwn_kernel<{tn},{lv % 1000 // 10},{lv % 10}>"
        return out
    L.check(L.lib().ey_conv2d(ctypes.byref(d), L.stream()), "ey_conv2d")
    return out


def conv_pw_chain(mod, x, fold1, act1, fold2, act2, out):
    """y = act2(W2 . act1(W1 . x + b1) + b2) in ONE kernel (two chained 1x1 convs, registers only).  Returns None when the shape is
    outside the fused kernel (the caller then runs the two convs)."""
    L.require_device(x, "conv_pw_chain")
    if RECORD is not None:
        return None  # (block programs run the two convs as two stages)
    x = L.as_nhwc(x)
    B, cin, H, W = x.shape
    if x.dtype != torch.float16 or cin % 8 or (L.cstride(x) * 2) % 16 or x.data_ptr() % 16:
        return None

    def build():
        w1, b1 = fold1()
        w2, b2 = fold2()
        cmid, cout = w1.shape[0], w2.shape[0]
        klen = L.lib().ey_conv_chain_klen(cmid)
        if not klen or w1.shape[2] != 1 or w2.shape[2] != 1 or w2.shape[1] != cmid or not ((cmid == 80 and 64 < cout <= 80 and cout % 4 == 0 and 72 <= cin <= 96) or (cmid == 64 and 1 <= cout <= 16 and 40 <= cin <= 64)):
            return False  # cached: this pair runs as two convs
        perm = (ctypes.c_int * klen)()
        L.check(L.lib().ey_conv_chain_kperm(cmid, perm, klen), "ey_conv_chain_kperm")
        idx = torch.tensor(list(perm), dtype=torch.long)
        w2f = w2.detach().float().cpu()
        w2p = torch.zeros((cout, klen, 1, 1))
        w2p[:, idx >= 0] = w2f[:, idx[idx >= 0]]  # second contraction in the order the first GEMM leaves its results in the registers
        dev = x.device
        return (pack_conv_weight(w1, x.dtype, dev), b1.to(dev).float().contiguous() if b1 is not None else None, pack_conv_weight(w2p, x.dtype, dev),
                b2.to(dev).float().contiguous() if b2 is not None else None, cmid, cout)

    packed = mod._packed(_dev_key(x, "pwchain"), build)
    if packed is False:
        return None
    w1p, b1, w2p, b2, cmid, cout = packed
    if not L.is_nhwc_view(out) or tuple(out.shape) != (B, cout, H, W) or out.dtype != x.dtype:
        raise ValueError("conv_pw_chain: out= must be an NHWC view of the output shape")
    M = B * H * W
    with _tr("conv_pw2_kernel", _nb(x, out) + (cmid * cin + cout * cmid) * 2, 2.0 * M * (cmid * cin + cout * cmid), note=f"{cin}->{cmid}->{cout} {H}x{W}"):
        L.check(L.lib().ey_conv_pw_chain(L.dtype_code(x.dtype), B, H, W, cin, cmid, cout, x.data_ptr(), L.cstride(x), w1p.data_ptr(),
                                         b1.data_ptr() if b1 is not None else None, act1, w2p.data_ptr(), b2.data_ptr() if b2 is not None else None, act2,
                                         out.data_ptr(), L.cstride(out), L.stream()), "ey_conv_pw_chain")
    return out


def conv_pw_pair(a, b):
    """Two chained 1x1 convs as one launch where the shapes allow (ey_conv_pw_pair): `a` / `b` = dict(mod=, folded_fn=, act=, tag=[, srcs=, out=,
    res=, addz=, out_scale=]); b reads a's output.  Returns (y_a, y_b); y_b is None when only the first conv was run (caller runs b itself)."""
    first = dict(mod=a["mod"], srcs=a["srcs"], folded_fn=a["folded_fn"], k=1, s=1, p=0, act=a["act"], out=a.get("out"), res=a.get("res"), addz=a.get("addz"),
                 out_scale=a.get("out_scale", 1.0), tag=a.get("tag", ""))
    x = a["srcs"][0]
    if RECORD is None and len(a["srcs"]) == 1 and torch.is_tensor(x) and x.dtype == torch.float16 and x.is_cuda:
        b1 = conv2d(_build_only=True, **first)
        if b1 is not None:
            d1, y1, keep1 = b1
            b2 = conv2d(b["mod"], [y1], b["folded_fn"], 1, 1, 0, b["act"], out=b.get("out"), tag=b.get("tag", ""), _build_only=True)
            if b2 is not None:
                d2, y2, keep2 = b2
                B, cmid, H, W = y1.shape
                try:
                    with _tr(f"conv_pwc_kernel<f16,{cmid // 16}>", _nb(x, y1, y2, a.get("res"), a.get("addz")), 2.0 * B * H * W * cmid * (x.shape[1] + y2.shape[1]),
                             note=f"{x.shape[1]}->{cmid}->{y2.shape[1]} k1 {H}x{W}"):
                        L.check(L.lib().ey_conv_pw_pair(ctypes.byref(d1), ctypes.byref(d2), L.stream()), "ey_conv_pw_pair")
                    return y1, y2
                except NotImplementedError:  # EY_EUNSUPPORTED: returned before anything is launched
                    pass
    return conv2d(**first), None


def conv2d_direct(mod, x, folded_fn, k, s, p, g, act, out=None, res=None, tag=""):
    L.require_device(x, "conv2d_direct")
    _no_block("direct conv")
    if res is not None:
        raise NotImplementedError("residual add is only fused into the MFMA conv")
    x = L.as_nhwc(x)
    B, cin, H, W = x.shape

    def build():
        w, b = folded_fn()
        return w.to(x.device).contiguous(), (b.to(x.device).contiguous() if b is not None else None)

    w, bias = mod._packed(_dev_key(x, "direct" + tag), build)
    cout = w.shape[0]
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    if out is None:
        out = L.empty_nhwc(B, cout, Ho, Wo, x.dtype, x.device)
    d = L.ConvDirectDesc()
    d.dtype = L.dtype_code(x.dtype)
    d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, cin, Ho, Wo, cout
    d.k, d.stride, d.pad, d.groups, d.act = k, s, p, g, act
    d.x, d.x_cstride = x.data_ptr(), L.cstride(x)
    d.w_oihw = w.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.y, d.y_cstride = out.data_ptr(), L.cstride(out)
    with _tr("conv_direct_kernel", _nb(x, out), 2.0 * B * Ho * Wo * cout * (cin // g) * k * k):
        L.check(L.lib().ey_conv2d_direct(ctypes.byref(d), L.stream()), "ey_conv2d_direct")
    return out


def stem_conv(mod, x, folded_fn, act, out_dtype, out=None):
    """3x3/s2 conv straight from the NCHW-contiguous image (layer 0)."""
    L.require_device(x, "stem_conv")
    _no_block("stem conv")
    B, cin, H, W = x.shape

    def build():
        w, b = folded_fn()
        return w.to(x.device).contiguous(), b.to(x.device).contiguous()

    w, bias = mod._packed(("stem", x.device), build)
    cout = w.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = L.empty_nhwc(B, cout, Ho, Wo, out_dtype, x.device)
    with _tr("stem_kernel", _nb(x, out), 2.0 * B * Ho * Wo * cout * cin * 9):
        L.check(L.lib().ey_stem_conv(L.dtype_code(x.dtype), L.dtype_code(out_dtype), B, cin, H, W, cout, act, x.data_ptr(), w.data_ptr(),
                                     bias.data_ptr(), out.data_ptr(), L.cstride(out), L.stream()), "ey_stem_conv")
    return out


def _conv_pack(mod, x0, cin, k, tag=""):
    """Packed MFMA weights + fp32 bias of a plain Conv module, under the cache key (and with the tuple layout) ops.conv2d uses."""
    def build():
        w, b = mod.folded()
        if w.shape[1] != cin or w.shape[2] != k:
            raise ValueError(f"conv weight {tuple(w.shape)} does not match Cin={cin} k={k}")
        pk = pack_conv_weight(w, x0.dtype, x0.device)
        return pk, (None if b is None else b.to(x0.device).float().contiguous()), w.shape[0], pk.numel() // x0.element_size()

    return mod._packed(_dev_key(x0, "igemm" + tag), build)


def pw_conv3s2(cv2, conv3, srcs, out=None):
    """A block's closing 1x1 Conv over a two-part virtual concat + the stride-2 3x3 Conv that follows it (layers 2 -> 3 of the n-scale
    backbones: DSC3K2_Wavelet.cv2 / C3k2.cv2, reference block.py:357-396,3783-3788, then conv.py:41-59) as one launch; the 64-channel map
    between them stays in LDS.  Bit-identical to the two launches.  Returns None (nothing launched) outside the fused kernel's shapes."""
    if RECORD is not None or len(srcs) != 2 or any(isinstance(t, VirtualCat) for t in srcs):
        return None
    x0 = srcs[0]
    L.require_device(x0, "pw_conv3s2")
    if x0.dtype != torch.float16:
        return None
    srcs = [L.as_nhwc(t) for t in srcs]
    B, _, H, W = srcs[0].shape
    c1, c3 = cv2.conv, conv3.conv
    from .modules.conv import _act_code
    if (tuple(srcs[1].shape[0:1] + srcs[1].shape[2:]) != (B, H, W) or c1.kernel_size != (1, 1) or c1.stride != (1, 1) or c1.groups != 1 or c1.out_channels != 64
            or c3.kernel_size != (3, 3) or c3.stride != (2, 2) or c3.padding != (1, 1) or c3.dilation != (1, 1) or c3.groups != 1 or c3.in_channels != 64
            or c3.out_channels != 64 or c1.in_channels != srcs[0].shape[1] + srcs[1].shape[1] or any(t.shape[1] > 32 or t.shape[1] % 8 for t in srcs)
            or _act_code(cv2.act) != L.ACT_SILU or _act_code(conv3.act) != L.ACT_SILU
            or any((L.cstride(t) * 2) % 16 or t.data_ptr() % 16 for t in srcs)):
        return None
    w1, b1 = _conv_pack(cv2, x0, c1.in_channels, 1)[:2]
    w2, b2 = _conv_pack(conv3, x0, 64, 3)[:2]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = L.empty_nhwc(B, 64, Ho, Wo, x0.dtype, x0.device)
    elif not L.is_nhwc_view(out) or tuple(out.shape) != (B, 64, Ho, Wo) or out.dtype != x0.dtype:
        raise ValueError("pw_conv3s2: out= must be an NHWC view of the output shape")
    c0n, c1n = srcs[0].shape[1], srcs[1].shape[1]
    try:
        with _tr("pw3_kernel", _nb(srcs[0], srcs[1], out), 2.0 * B * (H * W * 64 * (c0n + c1n) + Ho * Wo * 64 * 576), note=f"{c0n}+{c1n}->64 k1 -> 64 k3s2 {H}x{W}"):
            L.check(L.lib().ey_conv_pw_conv3s2(L.dtype_code(x0.dtype), B, H, W, srcs[0].data_ptr(), c0n, L.cstride(srcs[0]), srcs[1].data_ptr(), c1n, L.cstride(srcs[1]), 64,
                                               w1.data_ptr(), b1.data_ptr(), L.ACT_SILU, 64, w2.data_ptr(), b2.data_ptr(), L.ACT_SILU, out.data_ptr(), L.cstride(out),
                                               L.stream()), "ey_conv_pw_conv3s2")
    except NotImplementedError:  # EY_EUNSUPPORTED: returned before anything is launched
        return None
    return out


def stem_pair(m0, m1, x, out=None):
    """Layers 0 + 1 (Conv 3->16 k3 s2 + Conv 16->32 k3 s2, both BN + SiLU; reference conv.py:41-59) as one launch from the NCHW-contiguous f16
    image: the stem's output -- the largest tensor of the forward, with one consumer -- stays in LDS.  Bit-identical to the two launches.
    Returns None (nothing launched) when the modules / shape are outside the fused kernel."""
    L.require_device(x, "stem_pair")
    if RECORD is not None or x.dtype != torch.float16 or x.dim() != 4 or not x.is_contiguous() or L.is_nhwc_view(x):
        return None
    c0, c1 = m0.conv, m1.conv
    B, cin, H, W = x.shape
    if (cin != 3 or c0.in_channels != 3 or c0.out_channels != 16 or c1.in_channels != 16 or c1.out_channels != 32 or W % 8 or x.data_ptr() % 16
            or any(c.kernel_size != (3, 3) or c.stride != (2, 2) or c.padding != (1, 1) or c.dilation != (1, 1) or c.groups != 1 for c in (c0, c1))):
        return None

    def build0():
        w, b = m0.folded()
        return w.to(x.device).contiguous(), b.to(x.device).contiguous()

    w0, b0 = m0._packed(("stem", x.device), build0)
    w1, b1 = _conv_pack(m1, x, 16, 3)[:2]
    Hs, Ws = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Ho, Wo = (Hs - 1) // 2 + 1, (Ws - 1) // 2 + 1
    if out is None:
        out = L.empty_nhwc(B, 32, Ho, Wo, x.dtype, x.device)
    elif not L.is_nhwc_view(out) or tuple(out.shape) != (B, 32, Ho, Wo) or out.dtype != x.dtype:
        raise ValueError("stem_pair: out= must be an NHWC view of the output shape")
    from .modules.conv import _act_code
    try:
        with _tr("stem_pair_kernel", _nb(x, out), 2.0 * B * (Hs * Ws * 16 * 27 + Ho * Wo * 32 * 144), note=f"3->16->32 {H}x{W}"):
            L.check(L.lib().ey_stem_pair(B, H, W, x.data_ptr(), w0.data_ptr(), b0.data_ptr(), _act_code(m0.act), 32, w1.data_ptr(), b1.data_ptr(), _act_code(m1.act),
                                         out.data_ptr(), L.cstride(out), L.stream()), "ey_stem_pair")
    except NotImplementedError:  # EY_EUNSUPPORTED: returned before anything is launched
        return None
    return out


def dwconv(mod, x, folded_fn, k, act, out=None, tag=""):
    L.require_device(x, "dwconv")
    x = L.as_nhwc(x)
    B, c, H, W = x.shape
    es = x.element_size()
    if c % 8 or (L.cstride(x) * es) % 16 or x.data_ptr() % 16:
        return conv2d_direct(mod, x, folded_fn, k, 1, k // 2, c, act, out=out, tag=tag)

    def build():
        w, b = folded_fn()  # (C,1,k,k)
        wk = w.view(c, k, k).permute(1, 2, 0).contiguous().to(device=x.device, dtype=x.dtype)  # [k][k][C]
        return wk, (b.to(x.device).contiguous() if b is not None else None)

    wk, bias = mod._packed(_dev_key(x, "dw" + tag), build)
    if out is None:
        out = L.empty_nhwc(B, c, H, W, x.dtype, x.device)
    if RECORD is not None:
        RECORD.dw(x, wk, bias, k, act, out)
        return out
    with _tr(f"dwconv_kernel<{k}>", _nb(x, out), 2.0 * x.numel() * k * k, note=f"C{c} {H}x{W}"):
        L.check(L.lib().ey_dwconv(L.dtype_code(x.dtype), B, H, W, c, k, act, x.data_ptr(), L.cstride(x), wk.data_ptr(),
                                  bias.data_ptr() if bias is not None else None, out.data_ptr(), L.cstride(out), L.stream()), "ey_dwconv")
    return out


def _dsconv_pack(mod, x, dw_fn, pw_fn, k):
    """Device operands of one fused DSConv, cached on the module per (dtype, device): depthwise weights [k][k][C], depthwise bias,
    packed pointwise weights (BN folded), bias, Cout, Toeplitz form of the depthwise weights (C 16 / 32) or None."""
    c = x.shape[1]

    def build():
        wd, bd = dw_fn()
        wp, bp = pw_fn()
        wkf = wd.detach().float().view(c, k, k).permute(1, 2, 0).contiguous().cpu()  # [k][k][C] fp32 host
        wk = wkf.to(device=x.device, dtype=x.dtype)
        tz = None
        if x.dtype == torch.float16 and c in (16, 32) and wp.shape[0] <= 32:  # Toeplitz-MFMA depthwise stage (ey_dsconv_tz)
            nb = L.lib().ey_dsconv_toeplitz_bytes(c, k)
            buf = torch.empty(nb, dtype=torch.uint8)
            w16 = wkf.half().float().contiguous()  # the f16-rounded weights the other kernels use (kept alive across the call)
            L.check(L.lib().ey_dsconv_pack_toeplitz(c, k, w16.data_ptr(), buf.data_ptr(), nb), "ey_dsconv_pack_toeplitz")
            tz = buf.to(x.device)
        return (wk, (bd.to(x.device).float().contiguous() if bd is not None else None), pack_conv_weight(wp, x.dtype, x.device),
                (bp.to(x.device).contiguous() if bp is not None else None), wp.shape[0], tz)

    return mod._packed(_dev_key(x, "dsfused"), build)


def dsb_pair(cv1, cv2, x, add, out=None):
    """DSBottleneck.forward (reference block.py:1496-1503) as one launch: y = [x +] cv2(cv1(x)) for two DSConv modules (k 3 -> 5 / 7, equal
    widths of 32 / 64 channels, f16, small maps).  Bit-identical to the two ey_dsconv launches.  Returns None when the shape is outside
    the fused kernel (nothing launched): the caller then runs the two DSConvs."""
    L.require_device(x, "dsb_pair")
    if RECORD is not None or x.dtype != torch.float16:
        return None
    x = L.as_nhwc(x)
    B, c, H, W = x.shape
    k1, k2 = cv1.dw.kernel_size[0], cv2.dw.kernel_size[0]
    if (c not in (32, 64) or k1 != 3 or k2 not in (5, 7) or cv1.pw.out_channels != c or cv2.pw.out_channels != c or cv2.dw.in_channels != c
            or any(m.dw.stride != (1, 1) or m.dw.dilation != (1, 1) or m.dw.padding != (m.dw.kernel_size[0] // 2,) * 2 or m.dw.bias is not None for m in (cv1, cv2))
            or (L.cstride(x) * 2) % 16 or x.data_ptr() % 16):
        return None
    if out is None:
        out = L.empty_nhwc(B, c, H, W, x.dtype, x.device)
    elif not L.is_nhwc_view(out) or tuple(out.shape) != (B, c, H, W):
        raise ValueError("dsb_pair: out= must be an NHWC view of the output shape")
    if (L.cstride(out) * 2) % 16 or out.data_ptr() % 16:
        return None
    base = x.untyped_storage().data_ptr()
    if out.untyped_storage().data_ptr() == base:
        # same buffer (channel slots of one concat buffer): fine while the channel ranges are disjoint; an overlap (in-place update) would race --
        # a workgroup reads halo rows of x that lie in other workgroups' bands
        cs = L.cstride(x)
        if L.cstride(out) != cs or abs(((x.data_ptr() - base) // 2) % cs - ((out.data_ptr() - base) // 2) % cs) < c:
            return None
    wk1, _, wp1, b1, _, _ = _dsconv_pack(cv1, x, cv1._dw_folded, cv1._pw_folded, k1)
    wk2, _, wp2, b2, _, _ = _dsconv_pack(cv2, x, cv2._dw_folded, cv2._pw_folded, k2)
    try:
        with _tr(f"dsb_pair_kernel<{k1},{k2}>", _nb(x, out), 2.0 * B * H * W * c * (k1 * k1 + k2 * k2 + 2 * c), note=f"C{c} {H}x{W}{' +res' if add else ''}"):
            L.check(L.lib().ey_dsb_pair(L.dtype_code(x.dtype), B, H, W, c, k1, k2, L.ACT_SILU, x.data_ptr(), L.cstride(x), wk1.data_ptr(), wp1.data_ptr(),
                                        b1.data_ptr() if b1 is not None else None, wk2.data_ptr(), wp2.data_ptr(), b2.data_ptr() if b2 is not None else None,
                                        1 if add else 0, out.data_ptr(), L.cstride(out), L.stream()), "ey_dsb_pair")
    except NotImplementedError:  # EY_EUNSUPPORTED: returned before anything is launched
        return None
    return out


def dsconv(mod, x, dw_fn, pw_fn, k, act, out=None, res=None, dw_act=0):
    """Fused depthwise->pointwise: y = res + act(pw1x1(dw_act(dw_kxk(x) + dw_bias)) + bias).  Returns None when the shape is
    outside the fused kernel (caller then runs the two-kernel form)."""
    L.require_device(x, "dsconv")
    x = L.as_nhwc(x)
    B, c, H, W = x.shape
    es = x.element_size()
    if c % 8 or c > 256 or (L.cstride(x) * es) % 16 or x.data_ptr() % 16:
        return None

    wk, dwb, wp, bias, cout, tz = _dsconv_pack(mod, x, dw_fn, pw_fn, k)
    if out is None:
        out = L.empty_nhwc(B, cout, H, W, x.dtype, x.device)
    elif not L.is_nhwc_view(out) or tuple(out.shape) != (B, cout, H, W):
        raise ValueError("dsconv: out= must be an NHWC view of the output shape")
    if res is not None:
        res = L.as_nhwc(res)
    if RECORD is not None:  # two stages: depthwise (rounded to the storage type like the reference's intermediate tensor) -> pointwise 1x1
        t = L.empty_nhwc(B, c, H, W, x.dtype, x.device)
        RECORD.dw(x, wk, dwb, k, dw_act, t)
        d = L.ConvDesc()
        d.dtype, d.B, d.H, d.W, d.Ho, d.Wo, d.Cout = L.dtype_code(x.dtype), B, H, W, H, W, cout
        d.k, d.stride, d.pad, d.act, d.nsrc = 1, 1, 0, act, 1
        d.src_C[0] = c
        d.y_cstride, d.out_scale, d.ngroup = L.cstride(out), 1.0, 1
        RECORD.conv(d, [t], [0], out, res, None, wp, bias, 2.0 * B * H * W * c * cout, cout * c * es)
        return out
    try:
      rec = _tr(f"dsconv_kernel<{k}>", _nb(x, out, res), 2.0 * B * H * W * c * (k * k + cout), note=f"C{c}->{cout} {H}x{W}{' +res' if res is not None else ''}")
      with rec:
        tail = (dwb.data_ptr() if dwb is not None else None, dw_act, wp.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), L.cstride(out),
                res.data_ptr() if res is not None else None, L.cstride(res) if res is not None else 0, L.stream())
        if tz is not None:
            L.check(L.lib().ey_dsconv_tz(L.dtype_code(x.dtype), B, H, W, c, cout, k, act, x.data_ptr(), L.cstride(x), wk.data_ptr(), tz.data_ptr(), *tail), "ey_dsconv_tz")
        else:
            L.check(L.lib().ey_dsconv(L.dtype_code(x.dtype), B, H, W, c, cout, k, act, x.data_ptr(), L.cstride(x), wk.data_ptr(), *tail), "ey_dsconv")
        if TRACE is not None:
            rec.kernel = {2: f"dsconv_strip_kernel<{k}>", 3: f"dsconv_tz_kernel<{k}>"}.get(L.lib().ey_dsconv_last_variant(), rec.kernel)
    except NotImplementedError:  # EY_EUNSUPPORTED is returned before anything is launched (tile does not fit LDS): two-kernel form
        return None
    return out


def dwt_haar(x, out=None):
    """(B,C,H,W) -> (B,4C,H/2,W/2) with channel blocks LL|LH|HL|HH."""
    L.require_device(x, "dwt_haar")
    x = L.as_nhwc(x)
    B, c, H, W = x.shape
    if out is None:
        out = L.empty_nhwc(B, 4 * c, H // 2, W // 2, x.dtype, x.device)
    if RECORD is not None:
        if c % 8 or (L.cstride(x) * x.element_size()) % 16 or x.data_ptr() % 16:
            _no_block("dwt alignment")
        RECORD.dwt(x, out)
        return out
    with _tr("dwt_kernel", _nb(x, out), 4.0 * x.numel()):
        L.check(L.lib().ey_dwt_haar(L.dtype_code(x.dtype), B, H, W, c, x.data_ptr(), L.cstride(x), out.data_ptr(), L.cstride(out), L.stream()), "ey_dwt_haar")
    return out


def wavelet_z(mod, b, sets_fn, z_fn):
    """Half-resolution branch of _WaveletEnhancer in ONE kernel (ey_wavelet_z): b (B,c,H,W) -> Z (B,c,H/2,W/2).  sets_fn() ->
    ((w_ll3x3, b_ll), (w_h, b_h)) BN-folded fp32; z_fn() -> (w_z (c,2c,1,1), None).  Returns None when the shape is outside the fused
    kernel (fp32 mode, c not in {16,32,64,128}, while a block program is recorded): the caller runs dwt + the two convs."""
    L.require_device(b, "wavelet_z")
    b = L.as_nhwc(b)
    B, c, H, W = b.shape
    if RECORD is not None or b.dtype != torch.float16 or c not in (16, 32, 64, 128) or H < 2 or W < 2 or (L.cstride(b) * 2) % 16 or b.data_ptr() % 16:
        return None

    def build():
        (wl, bl), (wh, bh) = sets_fn()
        wz, _ = z_fn()
        packs = [pack_conv_weight(w, b.dtype, b.device) for w in (wl, wh)]
        return (torch.cat(packs), packs[0].numel() // 2, torch.cat([bl.to(b.device).float(), bh.to(b.device).float()]).contiguous(),
                pack_conv_weight(wz, b.dtype, b.device))

    wsub, wset, bias, wz = mod._packed(_dev_key(b, "wavelet_z"), build)
    z = L.empty_nhwc(B, c, H // 2, W // 2, b.dtype, b.device)
    M = B * (H // 2) * (W // 2)
    flops = 2.0 * M * ((c // 2) * c + 3 * (c // 2) * 9 * c + c * 2 * c) + 4.0 * b.numel()
    with _tr("wavelet_z_kernel", _nb(b, z) + ((c // 2) * 10 * c + 2 * c * c) * 2, flops, note=f"C{c} {H}x{W}"):
        L.check(L.lib().ey_wavelet_z(L.dtype_code(b.dtype), B, H, W, c, b.data_ptr(), L.cstride(b), wsub.data_ptr(), wset, bias.data_ptr(), wz.data_ptr(),
                                     z.data_ptr(), L.cstride(z), L.stream()), "ey_wavelet_z")
    return z


def sppf_pool(x, y1, y2, y3):
    L.require_device(x, "sppf_pool")
    B, c, H, W = x.shape
    cs = L.cstride(y1)
    if L.cstride(y2) != cs or L.cstride(y3) != cs:
        raise ValueError("sppf_pool: outputs must share one pixel stride")
    if RECORD is not None:
        RECORD.pool(x, y1, y2, y3)
        return
    with _tr("sppf_kernel", _nb(x, y1, y2, y3)):
        L.check(L.lib().ey_sppf_pool(L.dtype_code(x.dtype), B, H, W, c, x.data_ptr(), L.cstride(x), y1.data_ptr(), y2.data_ptr(), y3.data_ptr(), cs,
                                     L.stream()), "ey_sppf_pool")


def copy_slice(src, dst, up=0):
    """dst[b,c,y,x] = src[b,c,y>>up,x>>up] (both NHWC views)."""
    B, c, H, W = dst.shape
    _no_block("copy / concat / upsample")
    with _tr("copy_kernel", _nb(src, dst)):
        L.check(L.lib().ey_copy_nhwc(L.dtype_code(dst.dtype), B, H, W, c, up, src.data_ptr(), L.cstride(src), dst.data_ptr(), L.cstride(dst), L.stream()),
                "ey_copy_nhwc")
    return dst


def concat(xs):
    xs = [L.as_nhwc(as_tensor(t)) for t in xs]
    L.require_device(xs[0], "concat")
    B, _, H, W = xs[0].shape
    out = L.empty_nhwc(B, sum(t.shape[1] for t in xs), H, W, xs[0].dtype, xs[0].device)
    c0 = 0
    for t in xs:
        if (t.shape[0], t.shape[2], t.shape[3]) != (B, H, W) or t.dtype != out.dtype:
            raise ValueError(f"concat: incompatible inputs {[tuple(t.shape) for t in xs]}")
        copy_slice(t, out[:, c0:c0 + t.shape[1]])
        c0 += t.shape[1]
    return out


def upsample2x(x):
    x = L.as_nhwc(as_tensor(x))
    L.require_device(x, "upsample2x")
    B, c, H, W = x.shape
    return copy_slice(x, L.empty_nhwc(B, c, 2 * H, 2 * W, x.dtype, x.device), up=1)


def to_nchw_contiguous(x):
    """NHWC view -> plain contiguous NCHW tensor (what a foreign caller expects from .contiguous())."""
    x = L.as_nhwc(x)
    B, c, H, W = x.shape
    out = torch.empty((B, c, H, W), dtype=x.dtype, device=x.device)
    L.check(L.lib().ey_nhwc_to_nchw(L.dtype_code(x.dtype), B, c, H, W, x.data_ptr(), L.cstride(x), out.data_ptr(), L.stream()), "ey_nhwc_to_nchw")
    return out


def linear_attention(qkv, heads, out=None):
    """qkv (B,3C,H,W) [q|k|v] -> (B,C,H,W)."""
    L.require_device(qkv, "linear_attention")
    qkv = L.as_nhwc(qkv)
    B, c3, H, W = qkv.shape
    c = c3 // 3
    if out is None:
        out = L.empty_nhwc(B, c, H, W, qkv.dtype, qkv.device)
    if RECORD is not None:
        RECORD.linattn(qkv, heads, out)
        return out
    with _tr("linattn_kernel", _nb(qkv, out), 4.0 * B * H * W * c * (c // heads)):
        L.check(L.lib().ey_linear_attention(L.dtype_code(qkv.dtype), B, H * W, c, heads, qkv.data_ptr(), L.cstride(qkv), out.data_ptr(), L.cstride(out),
                                            L.stream()), "ey_linear_attention")
    return out


def softmax_attention(qkv, heads, kd, hd, scale, out=None):
    L.require_device(qkv, "softmax_attention")
    _no_block("softmax attention")
    qkv = L.as_nhwc(qkv)
    B, _, H, W = qkv.shape
    if out is None:
        out = L.empty_nhwc(B, heads * hd, H, W, qkv.dtype, qkv.device)
    with _tr("softattn_kernel", _nb(qkv, out), 2.0 * B * heads * (H * W) ** 2 * (kd + hd)):
        L.check(L.lib().ey_softmax_attention(L.dtype_code(qkv.dtype), B, H * W, heads, kd, hd, float(scale), qkv.data_ptr(), L.cstride(qkv),
                                             out.data_ptr(), L.cstride(out), L.stream()), "ey_softmax_attention")
    return out


def head_decode(box, cls, stride, q, pred, a_off):
    """One pyramid level of the fused DGQP + DFL + decode; q = (w1[hid,20], b1, w2[hid], b2) fp32 device tensors or None."""
    L.require_device(box, "head_decode")
    _no_block("head decode")
    B, _, H, W = box.shape
    nc = cls.shape[1]
    qa = [t.data_ptr() for t in q] if q is not None else [None] * 4
    hid = q[0].shape[0] if q is not None else 0
    with _tr("head_decode_kernel", _nb(box, cls) + B * H * W * (4 + nc) * 4, 2.0 * B * H * W * (hid * 21 + 200)):
        L.check(L.lib().ey_head_decode(L.dtype_code(box.dtype), B, H, W, nc, float(stride), box.data_ptr(), L.cstride(box), cls.data_ptr(), L.cstride(cls),
                                       qa[0], qa[1], qa[2], qa[3], hid, pred.data_ptr(), pred.shape[2], a_off, L.stream()), "ey_head_decode")


class Candidates:
    """NMS candidates written by the fused head decode (ey_head_decode_levels_nms): sort keys + best class per anchor + (cx,cy,w,h),
    for the conf threshold / class filter they were built with.  `utils.ops.nms_device` accepts it in place of `pred`."""

    def __init__(self, buf, B, nc, A, conf, classes, pred=None):
        self.buf, self.B, self.nc, self.A, self.conf, self.classes, self.pred = buf, B, nc, A, float(conf), (tuple(classes) if classes is not None else None), pred
        self.shape = (B, 4 + nc, A)
        self.device = buf.device


def nms_candidates(cand, iou_thres, max_det, max_nms, max_wh, agnostic):
    """Selection + greedy suppression on a Candidates buffer -> (boxes (B,max_det,6), count (B,), index (B,max_det))."""
    dev = cand.device
    boxes = torch.empty((cand.B, max_det, 6), dtype=torch.float32, device=dev)
    count = torch.empty((cand.B,), dtype=torch.int32, device=dev)
    index = torch.empty((cand.B, max_det), dtype=torch.int32, device=dev)
    # predict mode = the three-kernel fast path (csrc/nms_fast.inc.h), bracketed as one operator
    alg = cand.B * ((cand.A + 255) // 256 * 256 * 12 + cand.A * 16)  # keys + class ids + boxes (the tail of the buffer is scratch)
    with _tr("nms_fast(nf_select+nf_mask+nf_resolve)", alg + _nb(boxes), kernels=3):
        L.check(L.lib().ey_nms_candidates(cand.B, cand.nc, cand.A, cand.buf.data_ptr(), cand.buf.numel(), float(iou_thres), int(max_det), int(max_nms), float(max_wh),
                                          int(bool(agnostic)), boxes.data_ptr(), count.data_ptr(), index.data_ptr(), L.stream()), "ey_nms_candidates")
    return boxes, count, index


def head_decode_levels(levels, pred, nms=None, xyxy=False):
    """levels: list (<= 4) of (box, cls, stride, q-or-None, a_off) -> every level of the fused DGQP + DFL + decode in ONE launch.
    nms = (conf_thres, class_mask uint8 tensor or None, classes): also build the NMS candidates in the same pass (pred may then be None);
    returns a Candidates object.  xyxy: rows 0-3 = x1,y1,x2,y2 (end2end heads, reference head.py:163-165)."""
    n = len(levels)
    box0, cls0 = levels[0][0], levels[0][1]
    L.require_device(box0, "head_decode")
    B, nc = box0.shape[0], cls0.shape[1]
    q0 = levels[0][3]
    hid = q0[0].shape[0] if q0 is not None else 0
    IA, FA, PA = ctypes.c_int * n, ctypes.c_float * n, ctypes.c_void_p * n
    Hs, Ws = IA(*[lv[0].shape[2] for lv in levels]), IA(*[lv[0].shape[3] for lv in levels])
    st = FA(*[float(lv[2]) for lv in levels])
    boxp, clsp = PA(*[lv[0].data_ptr() for lv in levels]), PA(*[lv[1].data_ptr() for lv in levels])
    boxcs, clscs = IA(*[L.cstride(lv[0]) for lv in levels]), IA(*[L.cstride(lv[1]) for lv in levels])
    offs = IA(*[int(lv[4]) for lv in levels])
    qa = [PA(*[(lv[3][j].data_ptr() if lv[3] is not None else None) for lv in levels]) for j in range(4)]
    nbytes = sum(_nb(lv[0], lv[1]) + B * lv[0].shape[2] * lv[0].shape[3] * (4 + nc) * 4 for lv in levels)
    flops = sum(2.0 * B * lv[0].shape[2] * lv[0].shape[3] * (hid * 21 + 200) for lv in levels)
    if nms is not None:
        conf, mask, classes = nms
        A = sum(lv[0].shape[2] * lv[0].shape[3] for lv in levels)
        nb = L.lib().ey_nms_candidates_bytes(B, A)
        buf = torch.empty(nb, dtype=torch.uint8, device=box0.device)
        in_bytes = sum(_nb(lv[0], lv[1]) for lv in levels)
        with _tr("head_decode_kernel", in_bytes + nb + (B * A * (4 + nc) * 4 if pred is not None else 0), flops, note=f"{n} levels + NMS keys"):
            L.check(L.lib().ey_head_decode_levels_nms(L.dtype_code(box0.dtype), B, n, Hs, Ws, st, boxp, boxcs, clsp, clscs, nc, qa[0], qa[1], qa[2], qa[3], hid,
                                                      pred.data_ptr() if pred is not None else None, A, offs, float(conf), mask.data_ptr() if mask is not None else None,
                                                      buf.data_ptr(), nb, L.stream()), "ey_head_decode_levels_nms")
        return Candidates(buf, B, nc, A, conf, classes, pred)
    with _tr("head_decode_kernel", nbytes, flops, note=f"{n} levels"):
        fn = L.lib().ey_head_decode_levels_xyxy if xyxy else L.lib().ey_head_decode_levels
        L.check(fn(L.dtype_code(box0.dtype), B, n, Hs, Ws, st, boxp, boxcs, clsp, clscs, nc, qa[0], qa[1], qa[2], qa[3], hid,
                   pred.data_ptr(), pred.shape[2], offs, L.stream()), "ey_head_decode_levels")


def scale_img(x, ratio, flip_lr=False, gs=32):
    """scale_img(x.flip(3) if flip_lr else x, ratio, gs=gs) of the reference (utils/torch_utils.py:423-432; ratio 1 and no flip -> x itself)
    on a contiguous NCHW image batch: one kernel (flip + bilinear resize + 0.447 padding)."""
    import math
    L.require_device(x, "scale_img")
    if ratio == 1.0 and not flip_lr:
        return x
    x = x.contiguous()
    B, Cc, H, W = x.shape
    if ratio == 1.0:
        hs, ws, Hp, Wp = H, W, H, W
    else:
        hs, ws = int(H * ratio), int(W * ratio)
        Hp, Wp = (math.ceil(v * ratio / gs) * gs for v in (H, W))
    y = torch.empty((B, Cc, Hp, Wp), dtype=x.dtype, device=x.device)
    with _tr("scale_img_kernel", _nb(x, y), 8.0 * y.numel()):
        L.check(L.lib().ey_scale_img(L.dtype_code(x.dtype), B, Cc, H, W, x.data_ptr(), hs, ws, Hp, Wp, int(bool(flip_lr)), 0.447, y.data_ptr(), L.stream()), "ey_scale_img")
    return y


def tta_merge(pred, lo, hi, scale, flip, img_hw, out, out_off):
    """out[:, :, out_off : out_off + hi - lo] = _descale_pred(pred, flip, scale, img_hw)[:, :, lo:hi] (reference tasks.py:388-408)."""
    L.require_device(pred, "tta_merge")
    B, no, A = pred.shape
    if pred.dtype != torch.float32 or out.dtype != torch.float32 or not pred.is_contiguous() or not out.is_contiguous():
        raise TypeError("tta_merge: contiguous fp32 (B, no, A) tensors")
    with _tr("tta_merge_kernel", 8 * B * no * (hi - lo)):
        L.check(L.lib().ey_tta_merge(B, no, A, pred.data_ptr(), lo, hi, float(scale), int(flip or 0), int(img_hw[0]), int(img_hw[1]), out.data_ptr(), out.shape[2],
                                     out_off, L.stream()), "ey_tta_merge")
    return out


def copy_bytes(dst, src):
    """dst (device, contiguous) <- src (contiguous; device or PINNED HOST memory, which the device reads over PCIe itself) with a plain
    copy kernel in the current stream: unlike an H2D hipMemcpyAsync it runs beside kernels of other streams (engine/model.py)."""
    L.require_device(dst, "copy_bytes")
    n = dst.numel() * dst.element_size()
    if not (dst.is_contiguous() and src.is_contiguous() and src.numel() * src.element_size() == n and n % 16 == 0 and (src.is_cuda or src.is_pinned())
            and dst.data_ptr() % 16 == 0 and src.data_ptr() % 16 == 0):
        raise ValueError("copy_bytes: contiguous 16-byte aligned tensors of equal byte size (source on the device or in pinned host memory)")
    L.check(L.lib().ey_copy_linear(src.data_ptr(), dst.data_ptr(), n, L.stream()), "ey_copy_linear")
    return dst


def e2e_topk(pred, k, want_index=False):
    """Detect.postprocess (reference head.py:167-189): pred fp32 (B,4+nc,A) with x1y1x2y2 rows -> (B,k,6) fp32 rows
    [x1,y1,x2,y2,score,class], the k best (anchor, class) pairs per image in descending score order."""
    L.require_device(pred, "e2e_topk")
    if pred.dtype != torch.float32 or not pred.is_contiguous():
        raise ValueError("e2e_topk: pred must be a contiguous float32 (B,4+nc,A) tensor")
    B, no, A = pred.shape
    nc = no - 4
    if not 0 < k <= A:
        raise ValueError(f"e2e_topk: k={k} must be in 1..A={A}")
    out = torch.empty((B, k, 6), dtype=torch.float32, device=pred.device)
    index = torch.empty((B, k), dtype=torch.int32, device=pred.device) if want_index else None
    nb = L.lib().ey_e2e_topk_workspace_bytes(B, nc, A)
    ws = torch.empty(nb, dtype=torch.uint8, device=pred.device)
    with _tr("e2e_topk(score+select)", pred.numel() * 4 + out.numel() * 4):
        L.check(L.lib().ey_e2e_topk(B, nc, A, pred.data_ptr(), k, out.data_ptr(), index.data_ptr() if want_index else None, ws.data_ptr(), nb, L.stream()), "ey_e2e_topk")
    return (out, index) if want_index else out


def nms(pred, conf_thres, iou_thres, max_det, max_nms, max_wh, agnostic, class_mask=None, multi_label=False, return_workspace=False):
    """pred fp32 (B,4+nc,A) contiguous -> (boxes (B,max_det,6) fp32, count (B,) int32, index (B,max_det) int32).
    return_workspace (tests): also the workspace tensor (keys, class ids, the fast path's per-image scratch with its n_sel / n_total / done words)."""
    L.require_device(pred, "nms")
    if pred.dtype != torch.float32 or not pred.is_contiguous():
        raise ValueError("nms: pred must be a contiguous float32 (B,4+nc,A) tensor")
    B, no, A = pred.shape
    dev = pred.device
    multi_label = bool(multi_label) and no - 4 > 1  # reference ops.py:240
    boxes = torch.empty((B, max_det, 6), dtype=torch.float32, device=dev)
    count = torch.empty((B,), dtype=torch.int32, device=dev)
    index = torch.empty((B, max_det), dtype=torch.int32, device=dev)
    nbytes = L.lib().ey_nms_workspace_bytes_ml(B, no - 4, A) if multi_label else L.lib().ey_nms_workspace_bytes(B, A)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with _tr("nms(score+sort_greedy)", _nb(pred, boxes)):
        L.check(L.lib().ey_nms(B, no - 4, A, pred.data_ptr(), float(conf_thres), float(iou_thres), int(max_det), int(max_nms), float(max_wh), int(bool(agnostic)), int(bool(multi_label)),
                               class_mask.data_ptr() if class_mask is not None else None, boxes.data_ptr(), count.data_ptr(), index.data_ptr(),
                               ws.data_ptr(), nbytes, L.stream()), "ey_nms")
    return (boxes, count, index, ws) if return_workspace else (boxes, count, index)
