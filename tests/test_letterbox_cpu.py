"""CPU: the LetterBox oracle (oracle/letterbox.py) and the product-side geometry (edge-yolo_amd/data/augment.py).
Geometry expectations are worked out by hand from the reference formula (data/augment.py:1559-1585); the resize restates
OpenCV's 8-bit INTER_LINEAR algorithm (cv2 is not installed: parity unpinned at that boundary, see the oracle docstring),
so it is held to algorithm-independent properties here."""
import numpy as np
import pytest
import torch

import edge_yolo_amd  # noqa: F401
from edge_yolo_amd import _lib as L
from edge_yolo_amd.data.augment import LetterBox
from oracle import letterbox as olb

GEO = [  # (shape hw, new_shape, auto) -> ((new_w,new_h), top, bottom, left, right)
    ((480, 640), 640, False, ((640, 480), 80, 80, 0, 0)),
    ((480, 640), 640, True, ((640, 480), 0, 0, 0, 0)),
    ((1080, 1920), 640, False, ((640, 360), 140, 140, 0, 0)),
    ((1080, 1920), 640, True, ((640, 360), 12, 12, 0, 0)),       # dh = 280 % 32 = 24 -> 12 / 12
    ((500, 375), 640, False, ((480, 640), 0, 0, 80, 80)),
    ((333, 500), 640, False, ((640, 426), 107, 107, 0, 0)),      # r = 1.28: 426.24 -> 426; dh = 214 -> 107 / 107
    ((427, 640), 640, True, ((640, 427), 10, 11, 0, 0)),         # dh = 213 % 32 = 21 -> 10.5: round(10.4) / round(10.6)
    ((100, 100), (320, 640), False, ((320, 320), 0, 0, 160, 160)),
]


@pytest.mark.parametrize("shape,new,auto,want", GEO)
def test_geometry_matches_reference_formula(shape, new, auto, want):
    for geo in (olb.letterbox_geometry(shape, new, auto=auto), LetterBox(new, auto=auto).geometry(shape)):
        assert (geo[0], geo[1], geo[2], geo[3], geo[4]) == want


def test_resize_properties():
    r = np.random.default_rng(3)
    img = r.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(olb.resize_linear_u8(img, 53, 37), img)                       # identity
    flat = np.full((20, 30, 3), 77, np.uint8)
    assert np.all(olb.resize_linear_u8(flat, 71, 45) == 77)                              # constants survive the fixed-point path
    up = olb.resize_linear_u8(img, 106, 74)
    assert up.min() >= img.min() and up.max() <= img.max()                               # convex combination
    even = img[:36, :52]
    box = ((even[0::2, 0::2].astype(int) + even[0::2, 1::2] + even[1::2, 0::2] + even[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(olb.resize_linear_u8(even, 26, 18), box)                       # exact 2x decimation = 2x2 box (cv::resize rule)
    ramp = np.repeat(np.arange(0, 256, 4, dtype=np.uint8)[None, :, None], 8, 0).repeat(3, 2)  # horizontal ramp
    big = olb.resize_linear_u8(ramp, 128, 8).astype(int)
    assert np.all(np.diff(big[0, :, 0]) >= 0)                                            # monotone stays monotone


def test_preprocess_layout_and_normalisation():
    r = np.random.default_rng(4)
    img = r.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    t = olb.preprocess([img], (64, 64))
    assert t.shape == (1, 3, 64, 64) and t.dtype == torch.float32
    assert torch.all(t[0, :, :8] == 114 / 255) and torch.all(t[0, :, 56:] == 114 / 255)  # 8 rows of padding top and bottom
    assert torch.equal(t[0, 0, 8:56], torch.from_numpy(img[..., 2].astype(np.float32)) / 255)  # plane 0 = R = source channel 2 (BGR)
    assert torch.equal(t[0, 2, 8:56], torch.from_numpy(img[..., 0].astype(np.float32)) / 255)
    h = olb.preprocess([img], (64, 64), half=True)
    assert h.dtype == torch.float16 and torch.equal(h, (t * 255).half() / 255)


def test_letterbox_needs_the_gpu():
    with pytest.raises(L.HipLibraryError):
        LetterBox(64).batch([np.zeros((8, 8, 3), np.uint8)], "cpu")
