"""CPU oracle (test infrastructure only) for the GPU pre-processing kernel `ey_letterbox`: LetterBox
(reference ultralytics/data/augment.py:1556-1591) followed by BasePredictor.preprocess (engine/predictor.py:123-133).

The geometry (`letterbox_geometry`) follows augment.py line by line.  The pixel stage needs `cv2.resize(INTER_LINEAR)` and
`cv2.copyMakeBorder`, i.e. OpenCV (opencv-python>=4.6.0, requirements.txt) which is NOT installed in the build image:
`resize_linear_u8` restates OpenCV's published 8-bit algorithm (modules/imgproc/src/resize.cpp: coefficient tables of
`resizeGeneric_`, `HResizeLinear`, the uchar `VResizeLinear` specialisation, and the INTER_LINEAR -> INTER_AREA switch for an
exact 2x decimation).  **Parity unpinned** at that boundary: none of the reference's tests hold a resized image, and OpenCV
wheels may dispatch to IPP/SIMD variants whose rounding is not guaranteed to be this one.  What IS pinned: the geometry
(pure Python arithmetic, checked against values computed by hand from augment.py) and the identity / pad / channel-order /
normalisation behaviour, which do not depend on OpenCV internals."""
import numpy as np


def letterbox_geometry(shape, new_shape=(640, 640), auto=False, scale_fill=False, scaleup=True, center=True, stride=32):
    """augment.py:1559-1585 -> (new_unpad (w,h), top, bottom, left, right, ratio)."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride
    elif scale_fill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    if center:
        dw /= 2
        dh /= 2
    top, bottom = (int(round(dh - 0.1)) if center else 0), int(round(dh + 0.1))
    left, right = (int(round(dw - 0.1)) if center else 0), int(round(dw + 0.1))
    return new_unpad, top, bottom, left, right, ratio


def _coef(n_dst, n_src, clamp_frac):
    scale = 1.0 / (float(n_dst) / float(n_src))  # cv::resize: inv_scale = dsize/ssize; scale = 1/inv_scale (double)
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:  # x direction: the fraction is zeroed where the window leaves the image
        lo, hi = s < 0, s >= n_src - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, n_src - 1, s))
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, a0, a1


def resize_linear_u8(img, new_w, new_h):
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for uint8 HWC images (restated, see module docstring)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    sh, sw = img.shape[:2]
    if (sw, sh) == (new_w, new_h):
        return img.copy()
    src = img.astype(np.int64)
    if sw == 2 * new_w and sh == 2 * new_h:  # INTER_LINEAR -> INTER_AREA fast path
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = _coef(new_w, sw, True)
    sy, b0, b1 = _coef(new_h, sh, False)
    x1 = np.minimum(sx + 1, sw - 1)
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    hrow = src[:, sx, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]  # (sh, new_w, 3) int, scale 2048
    r0, r1 = hrow[y0], hrow[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_u8(img, new_shape=(640, 640), auto=False, stride=32, pad=114):
    """LetterBox.__call__(image=img) -> padded uint8 HWC image (BGR order kept)."""
    (nw, nh), top, bottom, left, right, _ = letterbox_geometry(img.shape[:2], new_shape, auto=auto, stride=stride)
    if img.shape[:2][::-1] != (nw, nh):
        img = resize_linear_u8(img, nw, nh)
    out = np.full((nh + top + bottom, nw + left + right, 3), pad, np.uint8)
    out[top:top + nh, left:left + nw] = img
    return out


def preprocess(imgs, new_shape=(640, 640), auto=False, stride=32, half=False):
    """predictor.py:123-133: letterbox each image, stack, BGR->RGB, BHWC->BCHW, to float (half first when the model is fp16), /255."""
    import torch
    im = np.stack([letterbox_u8(x, new_shape, auto=auto, stride=stride) for x in imgs])
    im = np.ascontiguousarray(im[..., ::-1].transpose((0, 3, 1, 2)))
    t = torch.from_numpy(im)
    t = t.half() if half else t.float()
    return t / 255
