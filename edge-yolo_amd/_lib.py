"""ctypes binding of csrc/libedgeyolo_hip.so (C ABI: include/edgeyolo_hip.h) + torch-tensor <-> NHWC-view glue.

PyTorch is plumbing here (device memory, streams); all arithmetic on the path happens in the HIP library.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libedgeyolo_hip.so")

F16, F32 = 0, 1
ACT_NONE, ACT_SILU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3


class HipLibraryError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("Cout", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("act", C.c_int32), ("nsrc", C.c_int32),
        ("src", C.c_void_p * 2), ("src_C", C.c_int32 * 2), ("src_cstride", C.c_int32 * 2), ("src_up", C.c_int32 * 2),
        ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("y_cstride", C.c_int32),
        ("res", C.c_void_p), ("res_cstride", C.c_int32), ("out_scale", C.c_float),
        ("addz", C.c_void_p), ("addz_cstride", C.c_int32), ("addz_H", C.c_int32), ("addz_W", C.c_int32), ("ngroup", C.c_int32),
        ("src_gstride", C.c_int64), ("y_gstride", C.c_int64), ("w_gstride", C.c_int64), ("w_gmax", C.c_int32),
    ]


class ConvDirectDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Ho", C.c_int32),
        ("Wo", C.c_int32), ("Cout", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("groups", C.c_int32),
        ("act", C.c_int32), ("x", C.c_void_p), ("x_cstride", C.c_int32), ("w_oihw", C.c_void_p), ("bias", C.c_void_p),
        ("y", C.c_void_p), ("y_cstride", C.c_int32),
    ]


class BlockStage(C.Structure):
    """ey_block_stage (include/edgeyolo_hip.h): one stage of a block program."""
    _fields_ = [
        ("op", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("k", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32), ("nsrc", C.c_int32),
        ("src", C.c_int64 * 2), ("src_ext", C.c_int32 * 2), ("src_img", C.c_int64 * 2), ("src_cs", C.c_int32 * 2), ("src_C", C.c_int32 * 2),
        ("w", C.c_void_p), ("bias", C.c_void_p), ("w_g", C.c_int64), ("w_gmax", C.c_int32),
        ("y", C.c_int64), ("y_ext", C.c_int32), ("y_img", C.c_int64), ("y_cs", C.c_int32), ("Cout", C.c_int32),
        ("has_res", C.c_int32), ("res", C.c_int64), ("res_ext", C.c_int32), ("res_img", C.c_int64), ("res_cs", C.c_int32),
        ("has_addz", C.c_int32), ("addz", C.c_int64), ("addz_ext", C.c_int32), ("addz_img", C.c_int64), ("addz_cs", C.c_int32), ("addz_H", C.c_int32),
        ("addz_W", C.c_int32), ("out_scale", C.c_float),
        ("ngroup", C.c_int32), ("src_g", C.c_int64), ("y_g", C.c_int64), ("heads", C.c_int32),
        ("kpad", C.c_int32), ("nt_pack", C.c_int32), ("mt", C.c_int32), ("nti", C.c_int32), ("lds", C.c_int32), ("tile_nti", C.c_int32), ("tile_src_lds", C.c_int32 * 2), ("tile_src_lcs", C.c_int32 * 2), ("tile_y_lds", C.c_int32), ("tile_y_lcs", C.c_int32), ("tile_res_lds", C.c_int32), ("tile_res_lcs", C.c_int32), ("tile_lds_bytes", C.c_int32), ("zsy", C.c_float), ("zsx", C.c_float),
    ]


BLK_CONV, BLK_DW, BLK_DWT, BLK_POOL, BLK_LINATTN = range(5)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
# name -> (restype, argtypes): every symbol include/edgeyolo_hip.h declares
SIGNATURES = {
    "ey_last_error": (C.c_char_p, []),
    "ey_version": (_i, []),
    "ey_abi_sizeof": (_sz, [_i]),
    "ey_tune_set": (_i, [C.c_char_p, C.c_long]),
    "ey_tune_get": (C.c_long, [C.c_char_p]),
    "ey_conv_packed_bytes": (_sz, [_i, _i, _i, _i]),
    "ey_conv_pack_weight": (_i, [_i, _i, _i, _i, _vp, _vp, _sz]),
    "ey_conv2d": (_i, [C.POINTER(ConvDesc), _vp]),
    "ey_conv_pw_pair": (_i, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _vp]),
    "ey_conv_variant": (_i, [_i, _i, _i, _i, _i, _i, C.c_long, _i]),
    "ey_conv_last_variant": (_i, []),
    "ey_conv_pack_nt": (_i, [_i]),
    "ey_conv_chain_klen": (_i, [_i]),
    "ey_conv_chain_kperm": (_i, [_i, _vp, _i]),
    "ey_conv_pw_chain": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "ey_conv2d_direct": (_i, [C.POINTER(ConvDirectDesc), _vp]),
    "ey_stem_conv": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "ey_dwconv": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "ey_dsconv": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "ey_dsconv_toeplitz_bytes": (_sz, [_i, _i]),
    "ey_dsconv_pack_toeplitz": (_i, [_i, _i, _vp, _vp, _sz]),
    "ey_dsconv_tz": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "ey_dsconv_last_variant": (_i, []),
    "ey_conv_pw_conv3s2": (_i, [_i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "ey_stem_pair": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "ey_dsb_pair": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "ey_dwt_haar": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "ey_wavelet_z": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, C.c_long, _vp, _vp, _vp, _i, _vp]),
    "ey_sppf_pool": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "ey_copy_nhwc": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "ey_nchw_to_nhwc": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "ey_nhwc_to_nchw": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "ey_letterbox": (_i, [_i, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ey_copy_linear": (_i, [_vp, _vp, _sz, _vp]),
    "ey_letterbox_batch": (_i, [_i, _vp, _i, _i, _i, _i, C.c_long, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ey_linear_attention": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "ey_softmax_attention": (_i, [_i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _vp]),
    "ey_head_decode": (_i, [_i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "ey_head_decode_levels": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "ey_head_decode_levels_xyxy": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "ey_e2e_topk_workspace_bytes": (_sz, [_i, _i, _i]),
    "ey_e2e_topk": (_i, [_i, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "ey_block_stage_sizeof": (_sz, []),
    "ey_block_program_bytes": (_sz, [_i]),
    "ey_block_compile": (_i, [C.POINTER(BlockStage), _i, _vp, _sz]),
    "ey_block_run": (_i, [_vp, _i, _i, C.POINTER(C.c_void_p), _i, _vp]),
    "ey_block_tileable": (_i, [C.POINTER(BlockStage), _i]),
    "ey_block_run_tiles": (_i, [_vp, _i, _i, _i, _i, _i, C.POINTER(C.c_void_p), _i, _vp]),
    "ey_block_run_timed": (_i, [_vp, _i, _i, C.POINTER(C.c_void_p), _i, _vp, _vp]),
    "ey_nms_candidates_bytes": (_sz, [_i, _i]),
    "ey_head_decode_levels_nms": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _f, _vp, _vp, _sz, _vp]),
    "ey_nms_candidates": (_i, [_i, _i, _i, _vp, _sz, _f, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "ey_nms_workspace_bytes": (_sz, [_i, _i]),
    "ey_nms_workspace_bytes_ml": (_sz, [_i, _i, _i]),
    "ey_scale_img": (_i, [_i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "ey_tta_merge": (_i, [_i, _i, _i, _vp, _i, _i, _f, _i, _i, _i, _vp, C.c_long, _i, _vp]),
    "ey_nms": (_i, [_i, _i, _i, _vp, _f, _f, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


def lib():
    """Load the HIP library or fail loudly (there is no fallback implementation)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(f"{LIB_PATH} is not built: run `make -C {os.path.dirname(LIB_PATH)}` "
                                  "(or `python -c 'import __graft_entry__ as g; g.build()'`). edge-yolo_amd has no CPU/PyTorch fallback.")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:
            raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.ey_abi_sizeof(0) != C.sizeof(ConvDesc) or L.ey_abi_sizeof(1) != C.sizeof(ConvDirectDesc) or L.ey_block_stage_sizeof() != C.sizeof(BlockStage):
            raise HipLibraryError(f"{LIB_PATH}: struct layout differs from this binding (stale build?): rebuild with make -C csrc")
        _lib = L
    return _lib


def check(code, what=""):
    if code != 0:
        msg = lib().ey_last_error().decode(errors="replace")
        if code == -2:
            raise NotImplementedError(f"{what}: {msg}")
        raise HipLibraryError(f"{what} failed ({code}): {msg}")


def stream():
    """HIP stream every launch goes to: torch's current stream of the CURRENT device.  Operands must live on that device --
    `require_device` (called by every op on its first operand) raises otherwise; the engine entry points (BaseModel.predict,
    DetectionPredictor, predict_batches, nms_device) enter `torch.cuda.device(<operand device>)` so `device='cuda:1'` works."""
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt):
    if dt == torch.float16:
        return F16
    if dt == torch.float32:
        return F32
    raise TypeError(f"edge-yolo_amd computes in float16 or float32 storage, got {dt}")


def require_device(x, what):
    if not x.is_cuda:
        raise HipLibraryError(f"{what}: tensor is on '{x.device}'. edge-yolo_amd runs on MI355X only (no CPU fallback); "
                              "move the model/input to 'cuda'.")
    if x.device.index != torch.cuda.current_device():
        raise HipLibraryError(f"{what}: tensor is on '{x.device}' but the current device is cuda:{torch.cuda.current_device()}; kernels are launched on "
                              "the current device's stream: wrap the call in `with torch.cuda.device(x.device):` (model.predict / YOLO.predict do)")


def empty_nhwc(B, Cc, H, W, dtype, device):
    """Logical NCHW tensor with NHWC memory (== torch channels_last), the layout every kernel uses."""
    return torch.empty((B, H, W, Cc), dtype=dtype, device=device).permute(0, 3, 1, 2)


def is_nhwc_view(x):
    """True if logical-NCHW `x` is a (possibly channel-sliced) dense NHWC view: offset(b,c,y,x) = ((b*H+y)*W+x)*cs + c."""
    B, Cc, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    if Cc > 1 and sc != 1:
        return False
    if sw < Cc:
        return False
    if H > 1 and sh != W * sw:
        return False
    if B > 1 and sb != H * W * sw:
        return False
    return True


def as_nhwc(x):
    """Return an NHWC view of logical-NCHW x (no copy when it already is one)."""
    if hasattr(x, "materialize"):  # nn._ops.VirtualCat reaching an operator that needs a real tensor
        x = x.materialize()
    if x.dim() != 4:
        raise ValueError(f"expected a BCHW tensor, got shape {tuple(x.shape)}")
    if is_nhwc_view(x):
        return x
    from .nn import _ops
    _ops._no_block("layout conversion")  # (a block program is being recorded: the chain must stay NHWC)
    B, Cc, H, W = x.shape
    y = empty_nhwc(B, Cc, H, W, x.dtype, x.device)
    if x.is_contiguous():
        check(lib().ey_nchw_to_nhwc(dtype_code(x.dtype), B, Cc, H, W, x.data_ptr(), y.data_ptr(), Cc, stream()), "nchw_to_nhwc")
    else:
        y.copy_(x)  # arbitrary-stride foreign tensor: torch does the gather (boundary plumbing, not the hot path)
    return y


def cstride(x):
    """Pixel stride (elements) of an NHWC view."""
    B, Cc, H, W = x.shape
    if W > 1:
        return x.stride(3)
    if H > 1:
        return x.stride(2)
    if B > 1:
        return x.stride(0)
    return max(Cc, x.stride(3))
