"""-m gpu: scale / flip test-time augmentation (reference DetectionModel._predict_augment, nn/tasks.py:372-408; predict(augment=True))
and custom predictor injection (reference engine/model.py:505,552) through the C ABI, vs goldens produced by the reference itself
(tests/golden/make_golden.py::augment_case -> augment_n.npz) and vs the CPU oracle."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model as om, nms as onms  # noqa: E402
import synthdata as synth  # noqa: E402


def _model(dtype):
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd.nn.tasks import DetectionModel
    m = DetectionModel("yolo11n-test.yaml")
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict(sd)
    m = m.cuda().fuse()
    return (m.half() if dtype == torch.float16 else m.float()).eval(), sd


def test_scale_img_kernel_vs_aten():
    """ey_scale_img == F.pad(F.interpolate(x.flip(3), size, bilinear, align_corners=False), value=0.447) (utils/torch_utils.py:423-432)."""
    import math
    import torch.nn.functional as F
    from edge_yolo_amd.nn import _ops
    for (b, h, w), ratio, flip in (((2, 64, 64), 0.83, True), ((1, 96, 160), 0.67, False), ((1, 96, 160), 0.83, True), ((2, 64, 96), 1.0, True)):
        x = synth.synth_images(b, h, w, seed=8)
        got = _ops.scale_img(x.cuda(), ratio, flip_lr=flip, gs=32).cpu()
        xi = x.flip(3) if flip else x
        if ratio != 1.0:
            s = (int(h * ratio), int(w * ratio))
            xi = F.interpolate(xi, size=s, mode="bilinear", align_corners=False)
            hp, wp = (math.ceil(v * ratio / 32) * 32 for v in (h, w))
            xi = F.pad(xi, [0, wp - s[1], 0, hp - s[0]], value=0.447)
        assert got.shape == xi.shape
        torch.testing.assert_close(got, xi, rtol=0, atol=2e-6)
    xh = synth.synth_images(1, 64, 64, seed=9).half()
    got = _ops.scale_img(xh.cuda(), 0.67, flip_lr=False, gs=32).cpu()
    want = F.pad(F.interpolate(xh.float(), size=(42, 42), mode="bilinear", align_corners=False), [0, 22, 0, 22], value=0.447).half()
    torch.testing.assert_close(got.float(), want.float(), rtol=0, atol=1e-3)


@pytest.mark.parametrize("tag,shape", [("64", (2, 64, 64)), ("96x160", (1, 96, 160))])
def test_predict_augment_fp32_vs_reference_golden(golden_dir, tag, shape):
    """m(x, augment=True): (B, 4+nc, A_total) within the north-star fp32 bar (1e-3) of the reference's own output; the predict-time NMS
    on the HIP tensor is bit-exact vs the oracle NMS on the same tensor and reproduces the reference's rows."""
    from edge_yolo_amd.utils import ops
    g = np.load(os.path.join(golden_dir, "augment_n.npz"))
    m, _ = _model(torch.float32)
    x = synth.synth_images(*shape, seed=4).cuda()
    y, none = m(x, augment=True)
    torch.cuda.synchronize()
    assert none is None and tuple(y.shape) == g[f"y_{tag}"].shape
    np.testing.assert_allclose(y.cpu().numpy(), g[f"y_{tag}"], rtol=1e-4, atol=1e-3)
    det = ops.non_max_suppression(y, 0.25, 0.7, max_det=300)
    ref = onms.non_max_suppression(y.cpu().numpy(), 0.25, 0.7, max_det=300)
    for i, (a, b) in enumerate(zip(det, ref)):
        np.testing.assert_array_equal(a.cpu().numpy(), b)
        want = g[f"det_{tag}_{i}"]
        assert a.shape[0] == want.shape[0]
        np.testing.assert_allclose(a.cpu().numpy(), want, rtol=1e-4, atol=2e-3)


def test_predict_augment_through_the_facade_and_graph(cfg_dir):
    """YOLO.predict(augment=True): eager == captured-graph replay == model(x, augment=True) + NMS; f16 stays close to the fp32 oracle."""
    import edge_yolo_amd
    from edge_yolo_amd.utils import ops
    y = edge_yolo_amd.YOLO("yolo11n-test.yaml")
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in y.model.state_dict().items()})
    y.model.load_state_dict(sd)
    x = synth.synth_images(2, 128, 160, seed=6)
    r0 = y.predict(x, augment=True, graph=False, device="cuda:0")
    r1 = y.predict(x, augment=True, graph=True, device="cuda:0")
    r2 = y.predict(x, augment=True, graph=True, device="cuda:0")  # replay
    pred, _ = y.model(x.cuda(), augment=True)
    want = ops.non_max_suppression(pred, 0.25, 0.7, max_det=300)
    for a, b, c, w in zip(r0, r1, r2, want):
        w = w.clone()
        ops.clip_boxes(w[None], (128, 160))
        for r in (a, b, c):
            np.testing.assert_array_equal(r.boxes.data.cpu().numpy(), w.cpu().numpy())
    plain = y.predict(x, device="cuda:0")
    assert any(len(a) != len(p) or not np.array_equal(a.boxes.data.cpu().numpy(), p.boxes.data.cpu().numpy()) for a, p in zip(r0, plain))  # TTA changes the result
    # oracle (fp32) on the same input
    o = om.OracleModel(os.path.join(cfg_dir, "yolo11n-test.yaml"), sd)
    yo, _ = o.forward_augment(x)
    np.testing.assert_allclose(pred.cpu().numpy(), yo.numpy(), rtol=1e-4, atol=1e-3)
    yh = y.predict(x, augment=True, half=True, device="cuda:0")
    assert len(yh) == 2 and all(r.boxes.data.shape[1] == 6 for r in yh)
    ph, _ = y.model(x.cuda().half(), augment=True)
    assert float((ph[:, 4:].cpu() - yo[:, 4:]).abs().max()) < 2e-2 and float((ph[:, :4].cpu() - yo[:, :4]).abs().max()) < 0.015 * 160


def test_augment_guards():
    """end2end heads fall back to single scale with a warning (reference tasks.py:374-376); augment + fused candidate build is refused."""
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd.nn.tasks import DetectionModel, yaml_model_load
    d = dict(yaml_model_load("yolo11n-test.yaml"))
    d["head"] = [list(r) for r in d["head"]]
    d["head"][-1][2] = "E2EDetect"
    m = DetectionModel(d)
    m.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    m = m.cuda().fuse().float().eval()
    x = synth.synth_images(1, 64, 64).cuda()
    with pytest.warns(UserWarning, match="augment"):
        ya = m(x, augment=True)
    yb = m(x)
    assert torch.equal(ya[0], yb[0])
    m2, _ = _model(torch.float32)
    with pytest.raises(ValueError, match="augment"):
        m2(x, augment=True, head_nms={"conf": 0.25, "classes": None})


def test_custom_predictor_injection():
    """Model.predict(predictor=Cls) (reference engine/model.py:505,552): the injected class is constructed and used; switching back to
    the default rebuilds; a non-class is a TypeError."""
    import edge_yolo_amd
    from edge_yolo_amd.engine.predictor import DetectionPredictor
    calls = []

    class TopOne(DetectionPredictor):
        """keeps only the best row per image (a postprocess override, the usual reason to inject a predictor)."""

        def postprocess(self, boxes, count, img, source):
            calls.append(int(count.sum()))
            res = super().postprocess(boxes, count, img, source)
            for r in res:
                r.boxes.data = r.boxes.data[:1]
            return res

    y = edge_yolo_amd.YOLO("yolo11n-test.yaml")
    y.model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in y.model.state_dict().items()}))
    x = synth.synth_images(2, 128, 128)
    plain = y.predict(x, device="cuda:0")
    assert isinstance(y.predictor, DetectionPredictor) and not isinstance(y.predictor, TopOne)
    top = y.predict(x, device="cuda:0", predictor=TopOne)
    assert isinstance(y.predictor, TopOne) and len(calls) == 1
    for a, b in zip(plain, top):
        assert len(b.boxes.data) == 1 and torch.equal(b.boxes.data[0], a.boxes.data[0])
    y.predict(x, device="cuda:0", predictor=TopOne)
    assert len(calls) == 2  # same class, same options: the predictor is reused
    again = y.predict(x, device="cuda:0")
    assert not isinstance(y.predictor, TopOne) and all(torch.equal(a.boxes.data, b.boxes.data) for a, b in zip(plain, again))
    with pytest.raises(TypeError, match="predictor"):
        y.predict(x, device="cuda:0", predictor="not a class")
