"""Inference-time pre-transform: `LetterBox` with the reference's constructor and geometry (ultralytics/data/augment.py:1477-1591),
executed on the GPU.  The reference letterboxes on the host with cv2 (resize INTER_LINEAR + copyMakeBorder 114) and then
`BasePredictor.preprocess` (engine/predictor.py:123-133) stacks, flips BGR->RGB, transposes to BCHW, uploads the float batch
and divides by 255.  Here the raw uint8 image is uploaded (1/2 - 1/4 of the bytes of the float batch) and ONE kernel per image
(`ey_letterbox`) does resize + pad + channel flip + layout + normalisation straight into the NCHW batch tensor."""
import numpy as np
import torch

from .. import _lib as L


class LetterBox:
    def __init__(self, new_shape=(640, 640), auto=False, scaleFill=False, scaleup=True, center=True, stride=32):
        self.new_shape = new_shape
        self.auto, self.scaleFill, self.scaleup, self.stride, self.center = auto, scaleFill, scaleup, stride, center

    def geometry(self, shape):
        """augment.py:1559-1585 for an image of `shape` (h, w) -> ((new_w, new_h), top, bottom, left, right, ratio)."""
        new_shape = (self.new_shape, self.new_shape) if isinstance(self.new_shape, int) else tuple(self.new_shape)
        r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
        if not self.scaleup:
            r = min(r, 1.0)
        ratio = r, r
        new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
        dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
        if self.auto:
            dw, dh = dw % self.stride, dh % self.stride
        elif self.scaleFill:
            dw, dh = 0.0, 0.0
            new_unpad = (new_shape[1], new_shape[0])
            ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
        if self.center:
            dw /= 2
            dh /= 2
        top, bottom = (int(round(dh - 0.1)) if self.center else 0), int(round(dh + 0.1))
        left, right = (int(round(dw - 0.1)) if self.center else 0), int(round(dw + 0.1))
        return new_unpad, top, bottom, left, right, ratio

    def batch(self, images, device, dtype=torch.float32, swap_rb=True):
        """list of HWC uint8 images (ndarray or tensor, BGR) -> (B,3,H,W) `dtype` tensor in [0,1] on `device`: LetterBox of every
        image + the reference's preprocess.  All images must letterbox to one (H,W) (the reference np.stack()s them)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise L.HipLibraryError("LetterBox.batch runs on the HIP path only: device must be a ROCm GPU")
        geo, out_hw = [], None
        for a in images:
            h, w = int(a.shape[0]), int(a.shape[1])
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError(f"LetterBox: expected HWC images with 3 channels, got {tuple(a.shape)}")
            (nw, nh), top, bottom, left, right, _ = self.geometry((h, w))
            hw = (nh + top + bottom, nw + left + right)
            if out_hw is None:
                out_hw = hw
            elif hw != out_hw:
                raise ValueError(f"LetterBox: images letterbox to different shapes {out_hw} vs {hw} (the reference cannot stack them either)")
            geo.append((h, w, nh, nw, top, left))
        H, W = out_hw
        out = torch.empty((len(images), 3, H, W), dtype=dtype, device=device)
        code = L.dtype_code(dtype)
        keep = []
        with torch.cuda.device(device):
            for i, (a, (h, w, nh, nw, top, left)) in enumerate(zip(images, geo)):
                t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
                if t.dtype != torch.uint8:
                    raise TypeError("LetterBox: images must be uint8")
                t = t.contiguous().to(device, non_blocking=True)
                keep.append(t)
                L.check(L.lib().ey_letterbox(code, t.data_ptr(), h, w, 3 * w, out[i].data_ptr(), H, W, nh, nw, top, left, 114, int(swap_rb), L.stream()), "ey_letterbox")
        return out

    def batch_tensor(self, images, dtype, swap_rb=True, out=None, device=None):
        """A decoded batch in ONE launch: images = uint8 tensor (B,h,w,3) (BGR like cv2), contiguous -> (B,3,H,W) `dtype` in [0,1]
        (LetterBox of every image with one geometry + the reference's preprocess, engine/predictor.py:123-133).  out: write there.
        images may live in PINNED HOST memory: the kernel then reads the bytes over PCIe itself (pinned memory is mapped into the
        device's address space) -- an ordinary kernel in the caller's stream, which, unlike an H2D copy, runs beside kernels of other
        streams (measured on this stack: a hipMemcpyAsync H2D does not overlap with kernels of another stream)."""
        host = torch.is_tensor(images) and not images.is_cuda and images.is_pinned()
        if not (torch.is_tensor(images) and (images.is_cuda or host) and images.dtype == torch.uint8 and images.dim() == 4 and images.shape[3] == 3
                and images.is_contiguous()):
            raise ValueError("LetterBox.batch_tensor: expected a contiguous uint8 tensor of shape (B,h,w,3), on the device or in pinned host memory")
        if host and (out is None and device is None):
            raise ValueError("LetterBox.batch_tensor: a pinned host batch needs `out` or `device`")
        dev = images.device if images.is_cuda else (out.device if out is not None else torch.device(device))
        B, h, w, _ = images.shape
        (nw, nh), top, bottom, left, right, _ = self.geometry((h, w))
        H, W = nh + top + bottom, nw + left + right
        if out is None:
            out = torch.empty((B, 3, H, W), dtype=dtype, device=dev)
        elif tuple(out.shape) != (B, 3, H, W) or out.dtype != dtype or not out.is_contiguous() or out.device != dev:
            raise ValueError(f"LetterBox.batch_tensor: out must be a contiguous {dtype} tensor of shape {(B, 3, H, W)} on {dev}")
        with torch.cuda.device(dev):
            L.check(L.lib().ey_letterbox_batch(L.dtype_code(dtype), images.data_ptr(), B, h, w, 3 * w, 3 * w * h, out.data_ptr(), H, W, nh, nw, top, left, 114,
                                               int(swap_rb), L.stream()), "ey_letterbox_batch")
        return out

    def __call__(self, labels=None, image=None):
        """Reference call form `LetterBox(...)(image=img)` -> letterboxed uint8 HWC ndarray (same channel order), computed by the GPU kernel."""
        if labels:
            raise NotImplementedError("LetterBox with labels is a training transform (out of scope: predict path only)")
        if not torch.cuda.is_available():
            raise L.HipLibraryError("LetterBox runs on the HIP path only and no ROCm GPU is visible")
        t = self.batch([image], torch.device("cuda", torch.cuda.current_device()), torch.float32, swap_rb=False)[0]
        return torch.round(t * 255).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
