"""Block programs (csrc/block.hip, `ey_block_*` in include/edgeyolo_hip.h): a chain of layers on small feature maps as ONE launch.

A program is RECORDED by running the ordinary module code of the chain once with `_ops.RECORD` set: every operator wrapper then
appends a stage (same pointers, strides, packed weights and epilogue options it would have launched with) instead of launching,
and allocates its output as usual.  Tensors the chain reads from outside / hands back are "external": they are addressed as
(index, byte offset) and bound per call, so a recorded program serves every later call with the same shapes; everything else
(intermediates, packed weights) belongs to the program.  Operators or shapes the block kernel does not take raise
`BlockUnsupported` while recording -- the caller then keeps the per-layer kernels for that chain (never a different result)."""
import ctypes

import torch

from .. import _lib as L


class BlockUnsupported(Exception):
    pass


def _img_stride(t):
    B, c, H, W = t.shape
    return t.stride(0) if B > 1 else H * W * L.cstride(t)


class BlockRecorder:
    def __init__(self, dtype):
        if dtype != torch.float16:
            raise BlockUnsupported("block programs are f16 only")
        self.stages, self.refs, self.keep, self.desc = [], [], [], []
        self.flops, self.wbytes = 0.0, 0

    # -- helpers
    def _new(self, op, src, out_hw):
        x = src
        if x.dtype != torch.float16 or not x.is_cuda:
            raise BlockUnsupported("f16 device tensors only")
        st = L.BlockStage()
        st.op = op
        st.H, st.W = x.shape[2], x.shape[3]
        st.Ho, st.Wo = out_hw
        st.nsrc, st.ngroup, st.out_scale = 1, 1, 1.0
        if st.H * st.W > 4096:
            raise BlockUnsupported("map too large for a per-image workgroup")
        return st

    def _ref(self, st, field, t, idx=None):
        """tensor reference: absolute pointer now, rebased onto an external tensor in finish()"""
        ptr = t.data_ptr()
        if idx is None:
            setattr(st, field, ptr)
            setattr(st, field + "_ext", -1)
            setattr(st, field + "_img", _img_stride(t))
            setattr(st, field + "_cs", L.cstride(t))
        else:
            getattr(st, field)[idx] = ptr
            getattr(st, field + "_ext")[idx] = -1
            getattr(st, field + "_img")[idx] = _img_stride(t)
            getattr(st, field + "_cs")[idx] = L.cstride(t)
        self.refs.append((len(self.stages), field, idx))
        self.keep.append(t)

    def _push(self, st):
        self.stages.append(st)
        names = {L.BLK_CONV: "conv", L.BLK_DW: "dw", L.BLK_DWT: "dwt", L.BLK_POOL: "pool", L.BLK_LINATTN: "linattn"}
        cin = st.src_C[0] + (st.src_C[1] if st.nsrc > 1 else 0)
        self.desc.append(f"{names[st.op]:7s} k{st.k} s{st.stride} {cin:4d}->{st.Cout:4d} {st.H}x{st.W}->{st.Ho}x{st.Wo} g{st.ngroup}"
                         f"{' +res' if st.has_res else ''}{' +addz' if st.has_addz else ''}")

    # -- operators
    def conv(self, d, srcs, up, out, res, addz, wp, bias, flops, wbytes):
        if any(up):
            raise BlockUnsupported("upsampled conv source")
        if d.Cout % 8 or d.dtype != L.F16:
            raise BlockUnsupported("conv shape")
        st = self._new(L.BLK_CONV, srcs[0], (d.Ho, d.Wo))
        st.k, st.stride, st.act, st.nsrc = d.k, d.stride, d.act, len(srcs)
        for i, t in enumerate(srcs):
            self._ref(st, "src", t, i)
            st.src_C[i] = d.src_C[i]
        st.w = wp.data_ptr()
        st.bias = bias.data_ptr() if bias is not None else None
        st.w_g, st.w_gmax = d.w_gstride, d.w_gmax
        st.Cout = d.Cout
        self._ref(st, "y", out)
        st.y_cs = d.y_cstride
        if res is not None:
            st.has_res = 1
            self._ref(st, "res", res)
        if addz is not None:
            st.has_addz = 1
            self._ref(st, "addz", addz)
            st.addz_H, st.addz_W = d.addz_H, d.addz_W
        st.out_scale = d.out_scale
        st.ngroup, st.src_g, st.y_g = max(1, d.ngroup), d.src_gstride, d.y_gstride
        if st.ngroup > 1:  # group views: the image stride is that of the whole tensor the group-0 slice lives in
            st.src_img[0] = _img_stride(srcs[0])
        self.keep += [wp, bias]
        self.flops += flops
        self.wbytes += wbytes
        self._push(st)

    def dw(self, x, wk, bias, k, act, out):
        st = self._new(L.BLK_DW, x, (x.shape[2], x.shape[3]))
        st.k, st.stride, st.act = k, 1, act
        self._ref(st, "src", x, 0)
        st.src_C[0] = x.shape[1]
        st.w = wk.data_ptr()
        st.bias = bias.data_ptr() if bias is not None else None
        st.Cout = x.shape[1]
        self._ref(st, "y", out)
        self.keep += [wk, bias]
        self.flops += 2.0 * x.numel() * k * k
        self.wbytes += wk.numel() * 2
        self._push(st)

    def dwt(self, x, out):
        st = self._new(L.BLK_DWT, x, (x.shape[2] // 2, x.shape[3] // 2))
        self._ref(st, "src", x, 0)
        st.src_C[0] = x.shape[1]
        st.Cout = 4 * x.shape[1]
        self._ref(st, "y", out)
        self.flops += 4.0 * x.numel()
        self._push(st)

    def pool(self, x, y1, y2, y3):
        c = x.shape[1]
        es = x.element_size()
        if y2.data_ptr() - y1.data_ptr() != c * es or y3.data_ptr() - y2.data_ptr() != c * es or L.cstride(y2) != L.cstride(y1):
            raise BlockUnsupported("pool outputs are not consecutive channel slots of one buffer")
        st = self._new(L.BLK_POOL, x, (x.shape[2], x.shape[3]))
        self._ref(st, "src", x, 0)
        st.src_C[0] = c
        st.Cout = 3 * c
        self._ref(st, "y", y1)
        self.keep += [y2, y3]
        self._push(st)

    def linattn(self, qkv, heads, out):
        c = qkv.shape[1] // 3
        if heads % 2 or c != 64 * heads:
            raise BlockUnsupported("linear attention: even number of 64-channel heads only")
        st = self._new(L.BLK_LINATTN, qkv, (qkv.shape[2], qkv.shape[3]))
        self._ref(st, "src", qkv, 0)
        st.src_C[0] = 3 * c
        st.Cout, st.heads = c, heads
        self._ref(st, "y", out)
        self.flops += 4.0 * qkv.shape[0] * qkv.shape[2] * qkv.shape[3] * c * 64
        self._push(st)

    # -- finish: external tensors -> (index, byte offset); compile; upload
    def finish(self, ins, outs, tag="", tiled=False):
        """tiled=True: the chain must be pointwise (1x1 convs on one map): it then runs as one workgroup per 16-pixel tile over the whole
        chip (ey_block_run_tiles) instead of one workgroup per image."""
        if not self.stages:
            raise BlockUnsupported("empty chain")
        ext = []  # (storage ptr, nbytes, "in" | "out")

        def add(t, kind):
            s = t.untyped_storage()
            for e in ext:
                if e[0] == s.data_ptr():
                    return
            ext.append((s.data_ptr(), s.nbytes(), kind))

        for t in ins:
            add(t, "in")
        for t in outs:
            add(t, "out")
        if len(ext) > 8:
            raise BlockUnsupported("more than 8 external tensors")
        for si, field, idx in self.refs:
            st = self.stages[si]
            ptr = getattr(st, field)[idx] if idx is not None else getattr(st, field)
            for e, (base, nbytes, _) in enumerate(ext):
                if base <= ptr < base + nbytes:
                    if idx is not None:
                        getattr(st, field)[idx] = ptr - base
                        getattr(st, field + "_ext")[idx] = e
                    else:
                        setattr(st, field, ptr - base)
                        setattr(st, field + "_ext", e)
                    break
        n = len(self.stages)
        arr = (L.BlockStage * n)(*self.stages)
        nbytes = L.lib().ey_block_program_bytes(n)
        host = torch.empty(nbytes, dtype=torch.uint8)
        try:
            L.check(L.lib().ey_block_compile(arr, n, host.data_ptr(), nbytes), "ey_block_compile")
        except L.HipLibraryError as e:
            raise BlockUnsupported(str(e)) from e
        dev = ins[0].device
        cs = (L.BlockStage * n).from_buffer_copy(host.numpy().tobytes())
        if tiled and not L.lib().ey_block_tileable(cs, n):
            raise BlockUnsupported("not a pointwise chain")
        prog = BlockProgram(host.to(dev), n, ins, outs, ext, [t for t in self.keep if t is not None], self.flops, self.wbytes, tag)
        prog.tiled, prog.HW, prog.tile_lds = bool(tiled), (cs[0].Ho, cs[0].Wo), int(cs[0].tile_lds_bytes)
        prog.desc = [f"{d}  tile {cs[i].mt}x{cs[i].nti}{' lds' if cs[i].lds else ''}" if cs[i].op == L.BLK_CONV else d for i, d in enumerate(self.desc)]
        return prog


def _sig(t):
    return (tuple(t.shape), tuple(t.stride()), t.dtype, t.data_ptr() - t.untyped_storage().data_ptr(), t.untyped_storage().nbytes())


class BlockProgram:
    def __init__(self, prog, n, ins, outs, ext, keep, flops, wbytes, tag):
        self.prog, self.n, self.keep, self.flops, self.wbytes, self.tag = prog, n, keep, flops, wbytes, tag
        self.B = ins[0].shape[0]
        self.in_sig = [_sig(t) for t in ins]
        self.out_sig = [_sig(t) for t in outs]
        self.ext_kind = [k for _, _, k in ext]
        self.ext_bytes = [nb for _, nb, _ in ext]
        base = [p for p, _, _ in ext]
        self.in_ext = [base.index(t.untyped_storage().data_ptr()) for t in ins]
        self.out_ext = [base.index(t.untyped_storage().data_ptr()) for t in outs]
        self.alg_bytes = sum(t.numel() * t.element_size() for t in list(ins) + list(outs)) + wbytes
        self.first_outs = list(outs)  # the tensors allocated while recording serve the first run
        self.timing = None
        self.tiled, self.HW, self.tile_lds = False, (0, 0), 0

    def matches(self, ins):
        return len(ins) == len(self.in_sig) and all(_sig(t) == s for t, s in zip(ins, self.in_sig))

    def run(self, ins, outs=None):
        """Launch.  ins: tensors with the recorded shapes / strides (any addresses).  outs: pre-allocated outputs with the recorded
        layout, or None -> fresh ones are allocated (the recording's own on the first run).  Returns the outputs."""
        from . import _ops
        if not self.matches(ins):
            raise ValueError("BlockProgram.run: inputs differ from the recorded signature")
        ptrs = [None] * len(self.ext_kind)
        for t, e in zip(ins, self.in_ext):
            ptrs[e] = t.untyped_storage().data_ptr()
        if outs is None:
            if self.first_outs is not None:
                outs, self.first_outs = self.first_outs, None
            else:
                stores, outs = {}, []
                for (shape, stride, dtype, off, nbytes), e in zip(self.out_sig, self.out_ext):
                    if e not in stores:
                        stores[e] = torch.empty(nbytes // 2, dtype=torch.float16, device=self.prog.device)
                    outs.append(stores[e].as_strided(shape, stride, off // 2))
        else:
            self.first_outs = None
            if [_sig(t) for t in outs] != self.out_sig:
                raise ValueError("BlockProgram.run: outputs differ from the recorded signature")
        for t, e in zip(outs, self.out_ext):
            ptrs[e] = t.untyped_storage().data_ptr()
        arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
        if self.timing is not None:  # developer tool (tools/block_stage_times.py): per-stage timestamps of workgroup 0
            L.check(L.lib().ey_block_run_timed(self.prog.data_ptr(), self.n, self.B, arr, len(ptrs), self.timing.data_ptr(), L.stream()), "ey_block_run_timed")
            return outs
        if self.tiled:
            with _ops._tr(f"block_tile_kernel<{self.tag}>", self.alg_bytes, self.flops, note=f"{self.n} stages"):
                L.check(L.lib().ey_block_run_tiles(self.prog.data_ptr(), self.n, self.B, self.HW[0], self.HW[1], self.tile_lds, arr, len(ptrs), L.stream()), "ey_block_run_tiles")
            return outs
        with _ops._tr(f"block_kernel<{self.tag}>", self.alg_bytes, self.flops, note=f"{self.n} stages"):
            L.check(L.lib().ey_block_run(self.prog.data_ptr(), self.n, self.B, arr, len(ptrs), L.stream()), "ey_block_run")
        return outs


class BlockCache:
    """Programs of one chain keyed by the input signature; `None` marks a signature the block kernel does not take."""

    def __init__(self, tag, tiled=False):
        self.tag, self.progs, self.tiled = tag, [], tiled
        self.unsupported = set()

    def clear(self):
        self.progs, self.unsupported = [], set()

    def run(self, fn, ins, outs=None):
        """fn(*ins) -> list of output tensors (the chain in ordinary module code; must write into `outs` when given).
        Returns the outputs, or None when the chain is not block-executable (the caller then runs fn itself)."""
        from . import _ops
        if _ops.RECORD is not None:
            return None  # no nesting
        key = tuple(_sig(t)[:3] for t in ins)
        if key in self.unsupported:
            return None
        for p in self.progs:
            if p.matches(ins):
                return p.run(ins, outs)
        rec = None
        try:
            rec = BlockRecorder(ins[0].dtype)
            _ops.RECORD = rec
            try:
                got = fn(*ins)
            finally:
                _ops.RECORD = None
            prog = rec.finish(ins, got, self.tag, tiled=self.tiled)
        except BlockUnsupported:
            self.unsupported.add(key)
            return None
        self.progs.append(prog)
        if len(self.progs) > 16:
            self.progs.pop(0)
        return prog.run(ins, outs if outs is not None else None)
