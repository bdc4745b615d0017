"""GPU: `ey_letterbox` (resize + pad + BGR->RGB + CHW + /255 in one kernel) against the CPU oracle, bit-exact (integer pixel
stage; IEEE division for the normalisation), and the ndarray source path of predict()."""
import numpy as np
import pytest
import torch

import edge_yolo_amd
from edge_yolo_amd.data.augment import LetterBox
from oracle import letterbox as olb

pytestmark = pytest.mark.gpu

CASES = [((480, 640), 640, False), ((480, 640), 640, True), ((333, 500), 640, False), ((1080, 1920), 640, True), ((97, 61), 128, False),
         ((720, 1280), 640, False), ((1280, 1280), 640, False), ((50, 40), 320, False), ((641, 639), 640, False)]


@pytest.mark.parametrize("shape,new,auto", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_letterbox_matches_oracle(shape, new, auto, dtype):
    r = np.random.default_rng(shape[0] * 7 + shape[1])
    imgs = [r.integers(0, 256, (*shape, 3), dtype=np.uint8) for _ in range(2)]
    got = LetterBox(new, auto=auto).batch(imgs, "cuda:0", dtype)
    want = olb.preprocess(imgs, (new, new), auto=auto, half=dtype == torch.float16)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert torch.equal(got.cpu(), want)


def test_call_form_returns_uint8_image():
    r = np.random.default_rng(11)
    img = r.integers(0, 256, (120, 200, 3), dtype=np.uint8)
    out = LetterBox(256)(image=img)
    assert out.dtype == np.uint8 and np.array_equal(out, olb.letterbox_u8(img, (256, 256)))


def test_predict_accepts_ndarray_sources():
    r = np.random.default_rng(5)
    img = r.integers(0, 256, (240, 320, 3), dtype=np.uint8)
    model = edge_yolo_amd.YOLO("yolo11n-test.yaml")
    res = model.predict([img, img], imgsz=320, conf=0.25)
    assert len(res) == 2 and res[0].orig_shape == (240, 320)
    x = olb.preprocess([img, img], (320, 320), auto=True).cuda()  # same shapes -> auto (minimum rectangle), as the reference
    res2 = model.predict(x, conf=0.25)
    assert x.shape[2:] == (256, 320)
    a, b = res[0].boxes.data.cpu(), res2[0].boxes.data.cpu()
    assert a.shape == b.shape
    # ndarray path scales boxes back to the original image: here gain = 1 and the padding (8,0) is removed
    b = b.clone()
    b[:, [1, 3]] -= 8
    b[:, :4] = b[:, :4].clamp(min=0)
    b[:, [0, 2]] = b[:, [0, 2]].clamp(max=320)
    b[:, [1, 3]] = b[:, [1, 3]].clamp(max=240)
    torch.testing.assert_close(a, b, rtol=0, atol=1e-3)
