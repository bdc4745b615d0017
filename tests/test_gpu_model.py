"""-m gpu: whole-model parity.  fp32 against the reference-produced goldens (tests/golden) and the oracle; fp16
against the oracle with the documented throughput-mode tolerance; predict() surface; hipGraph replay == eager."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om, nms as onms
import synthdata as synth  # noqa: E402


@pytest.fixture(scope="module")
def E():
    import edge_yolo_amd
    return edge_yolo_amd


def _build(E, name, dtype, gain=1.9):
    from edge_yolo_amd.nn.tasks import DetectionModel
    m = DetectionModel(name)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes, gain=gain)
    m.load_state_dict(sd)
    m = m.to("cuda")
    m.fuse()
    m = m.half() if dtype == torch.float16 else m.float()
    return m.eval(), sd


@pytest.mark.parametrize("name,tag", [("yolo11n-test.yaml", "edgeline_n_64"), ("yolo11n.yaml", "yolo11n_64"),
                                      ("yolo11n-GF2Detect.yaml", "gf2detect_n_64"), ("yolo11n-lineattention.yaml", "lineattention_n_64"),
                                      ("yolo11n-DSC3K2_Wavelet.yaml", "dsc3k2_wavelet_n_64"), ("yolo11n-tune.yaml", "tune_n_64")])
def test_fp32_vs_reference_golden_64(E, golden_dir, name, tag):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    m, _ = _build(E, name, torch.float32)
    b = g["y"].shape[0]
    y, raw = m(synth.synth_images(b, 64, 64).cuda())
    # north star: box coords / scores within 1e-3 in fp32
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=1e-4, atol=1e-3)
    for i, r in enumerate(raw):
        np.testing.assert_allclose(r.float().cpu().numpy(), g[f"raw{i}"], rtol=1e-4, atol=2e-4)


def test_fp32_layers_vs_reference_golden(E, golden_dir, cfg_dir):
    """per-layer outputs of EdgeLine-n against the reference captures (localises any mismatch)."""
    g = np.load(os.path.join(golden_dir, "edgeline_n_64.npz"))
    m, _ = _build(E, "yolo11n-test.yaml", torch.float32)
    x = synth.synth_images(2, 64, 64).cuda()
    y = []
    for layer in m.model:
        if layer.f != -1:
            x = y[layer.f] if isinstance(layer.f, int) else [x if j == -1 else y[j] for j in layer.f]
        x = layer(x)
        y.append(x if layer.i in m.save else None)
        if torch.is_tensor(x):
            np.testing.assert_allclose(x.float().cpu().numpy(), g[f"layer{layer.i}"], rtol=1e-4, atol=2e-4, err_msg=f"layer {layer.i} {layer.type}")


def test_fp32_nonsquare(E, golden_dir):
    g = np.load(os.path.join(golden_dir, "edgeline_n_96x160.npz"))
    m, _ = _build(E, "yolo11n-test.yaml", torch.float32)
    y, _ = m(synth.synth_images(1, 96, 160).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("name,tag", [("yolo11n-test.yaml", "edgeline_n_640"), ("yolo11n.yaml", "yolo11n_640")])
def test_fp32_640_and_postprocess(E, golden_dir, name, tag):
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    m, _ = _build(E, name, torch.float32)
    y, _ = m(synth.synth_images(1, 640, 640).cuda())
    step = int(g["step"])
    np.testing.assert_allclose(y[:, :, ::step].cpu().numpy(), g["y_sub"], rtol=1e-4, atol=1e-3)
    from edge_yolo_amd.utils import ops
    det = ops.non_max_suppression(y, 0.25, 0.7)[0].cpu().numpy()
    # NMS on OUR fp32 output must equal the oracle NMS on the same tensor bit for bit ...
    want, _ = onms.non_max_suppression(y.cpu().numpy(), 0.25, 0.7, return_idx=True)
    np.testing.assert_array_equal(det, want[0])
    # ... and agree with the reference's post-NMS rows (computed from ITS forward output) within the fp32 bar
    # (rows whose scores differ by less than fp32 noise may swap places or flip one suppression: match rows as a set)
    ref = g["det0"]
    assert det.shape == ref.shape
    hit = sum(bool((np.abs(ref - r).max(1) < 2e-3 + 1e-4 * np.abs(r).max()).any()) for r in det)
    assert hit >= 0.99 * len(det), f"only {hit}/{len(det)} post-NMS rows match the reference's"


@pytest.mark.parametrize("name", ["yolo11n-test.yaml", "yolo11n.yaml"])
def test_fp16_vs_oracle(E, cfg_dir, name):
    """Throughput mode.  fp16 storage: scores within 2e-2, boxes within 1.5% of the image size (documented in DESIGN.md)."""
    m, sd = _build(E, name, torch.float16)
    x = synth.synth_images(2, 320, 320)
    want, _ = om.OracleModel(os.path.join(cfg_dir, name), sd)(x)
    y, _ = m(x.cuda().half())
    y = y.cpu()
    assert y.dtype == torch.float32
    assert float((y[:, 4:] - want[:, 4:]).abs().max()) < 2e-2
    assert float((y[:, :4] - want[:, :4]).abs().max()) < 0.015 * 320


def test_predict_surface_and_graph(E):
    """YOLO(...).predict(): no task= needed for the GFL head, Results/Boxes surface, hipGraph replay == eager."""
    model = E.YOLO("yolo11n-test.yaml")
    model.model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in model.model.state_dict().items()}))
    x = synth.synth_images(2, 128, 128)
    r1 = model.predict(x, conf=0.25, iou=0.7, device="cuda:0", graph=False)
    r2 = model.predict(x, conf=0.25, iou=0.7, device="cuda:0", graph=True)
    r3 = model.predict(x, conf=0.25, iou=0.7, device="cuda:0", graph=True)  # replay
    assert len(r1) == 2
    for a, b, c in zip(r1, r2, r3):
        assert a.boxes.data.shape[1] == 6 and set(a.speed) == {"preprocess", "inference", "postprocess"}
        np.testing.assert_array_equal(a.boxes.data.cpu().numpy(), b.boxes.data.cpu().numpy())
        np.testing.assert_array_equal(a.boxes.data.cpu().numpy(), c.boxes.data.cpu().numpy())
        assert a.boxes.xyxy.shape[1] == 4 and a.boxes.conf.ndim == 1 and a.boxes.cls.ndim == 1
    with pytest.raises(RuntimeError):
        model.predict(x, device="cpu")


@pytest.mark.parametrize("name,hw", [("yolo11s-test.yaml", 128), ("yolo11s.yaml", 128), ("yolo11m-test.yaml", 64)])
def test_fp32_other_scales_vs_oracle(E, cfg_dir, name, hw):
    """Wider models: other channel counts exercise other tile variants (NT, chunked weights, 4-head attention)."""
    m, sd = _build(E, name, torch.float32)
    x = synth.synth_images(1, hw, hw)
    want, _ = om.OracleModel(os.path.join(cfg_dir, name), sd)(x)
    y, _ = m(x.cuda())
    np.testing.assert_allclose(y.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-3)


def test_1280_config5_shapes(E, cfg_dir):
    """BASELINE config 5 geometry (1280x1280: A=33600 anchors, N=1600 attention tokens): fp32 parity vs the oracle on one image
    and NMS rows bit-exact on that output."""
    name = "yolo11n-test.yaml"
    m, sd = _build(E, name, torch.float32)
    x = synth.synth_images(1, 1280, 1280)
    want, _ = om.OracleModel(os.path.join(cfg_dir, name), sd)(x)
    y, _ = m(x.cuda())
    assert tuple(y.shape) == (1, 84, 33600)
    np.testing.assert_allclose(y.cpu().numpy(), want.numpy(), rtol=2e-4, atol=4e-3)
    from edge_yolo_amd.utils import ops
    det = ops.non_max_suppression(y, 0.25, 0.7)[0].cpu().numpy()
    ref = onms.non_max_suppression(y.cpu().numpy(), 0.25, 0.7)[0]
    np.testing.assert_array_equal(det, ref)


def test_pipelined_runner_matches_single_graph(E):
    """The two-stream pipeline (forward(i+1) || NMS(i)) must return, per batch, exactly what the single-graph runner returns."""
    from edge_yolo_amd.engine.predictor import GraphRunner, PipelinedRunner
    from edge_yolo_amd.utils import ops
    m, _ = _build(E, "yolo11n-test.yaml", torch.float16)
    xs = [synth.synth_images(2, 128, 128, seed=s).cuda().half() for s in range(5)]

    def fwd(im):
        return m(im)[0]

    def post(pred):
        return ops.nms_device(pred, 0.25, 0.7, max_det=50)[:2]

    single = GraphRunner(lambda im: post(fwd(im)))
    want = []
    for x in xs:
        b, c = single(x)
        want.append((b.clone(), c.clone()))
    pipe = PipelinedRunner(fwd, post, xs[0])
    got = []
    for x in xs:
        j = pipe.submit(x)
        pipe.wait(j)
        torch.cuda.synchronize()
        b, c = pipe.outputs(j)
        got.append((b.clone(), c.clone()))
    # and back-to-back without waiting in between (the overlapped regime): last two results must still be right
    js = [pipe.submit(x) for x in xs[-2:]]
    pipe.wait()
    torch.cuda.synchronize()
    for (wb, wc), (gb, gc) in zip(want, got):
        assert torch.equal(wc, gc) and torch.equal(wb, gb)
    for j, (wb, wc) in zip(js, want[-2:]):
        b, c = pipe.outputs(j)
        assert torch.equal(wc, c) and torch.equal(wb, b)


def test_multi_stage_pipeline_matches_direct_execution():
    """PipelinedRunner with 4 stages (layer cuts as bench.py uses them) on DIFFERENT consecutive batches: every batch's boxes equal
    the ones the plain model + NMS produce (buffer-set rotation and the cross-stream event chain are correct)."""
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd.engine.predictor import PipelinedRunner
    from edge_yolo_amd.nn.tasks import DetectionModel
    from edge_yolo_amd.utils import ops
    import synthdata as synth
    m = DetectionModel("yolo11n-test.yaml")
    m.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    m = m.cuda().fuse().half().eval()
    n = len(m.model)
    cuts = [0, 9, 16, n - 1, n]
    stages = [(lambda st, lo=lo, hi=hi: m.forward_layers(st if lo else (st, []), lo, hi)) for lo, hi in zip(cuts[:-1], cuts[1:])]
    last = stages.pop()
    stages.append(lambda st: ops.nms_device(last(st)[0][0], 0.25, 0.7, max_det=300)[:2])
    torch.manual_seed(5)
    xs = [torch.rand(2, 3, 128, 160, device="cuda").half() for _ in range(7)]
    pipe = PipelinedRunner(*stages, xs[0])
    got = []
    for x in xs:
        j = pipe.submit(x)
        pipe.wait(j)
        torch.cuda.synchronize()
        b, c = pipe.outputs(j)
        got.append((b.clone(), c.clone()))
    # now back to back without waiting in between, results read after the drain of each buffer set
    js = []
    for x in xs[:4]:
        js.append(pipe.submit(x))
    pipe.wait()
    torch.cuda.synchronize()
    for k, x in enumerate(xs):
        wb, wc, _ = ops.nms_device(m(x)[0], 0.25, 0.7, max_det=300)
        assert torch.equal(got[k][1], wc)
        for i in range(2):
            torch.testing.assert_close(got[k][0][i, : int(wc[i])], wb[i, : int(wc[i])], rtol=0, atol=0, equal_nan=True, msg=lambda s_: f"batch {k} image {i}: {s_}")
    for k, j in enumerate(js):  # 4 in-flight batches used the 4 buffer sets
        wb, wc, _ = ops.nms_device(m(xs[k])[0], 0.25, 0.7, max_det=300)
        b, c = pipe.outputs(j)
        assert torch.equal(c, wc)
        for i in range(2):
            torch.testing.assert_close(b[i, : int(wc[i])], wb[i, : int(wc[i])], rtol=0, atol=0, equal_nan=True, msg=lambda s_: f"in-flight batch {k} image {i}: {s_}")


def test_predict_batches_matches_predict():
    """The throughput generator (4-stage batch pipeline) returns, in order, the same boxes as one predict() call per batch."""
    import edge_yolo_amd
    torch.manual_seed(11)
    model = edge_yolo_amd.YOLO("yolo11n-test.yaml")
    # input-sensitive weights and pageable fp32 host batches with half=True: every batch goes through a temporary device tensor that the
    # copy stream reads after the caller dropped it (regression: without record_stream a batch could receive the next batch's pixels --
    # invisible with default-initialised weights, whose boxes barely depend on the input)
    model.model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in model.model.state_dict().items()}))
    xs = [torch.rand(2, 3, 128, 160) for _ in range(9)]
    outs = list(model.predict_batches(xs, conf=0.25, half=True))
    assert len(outs) == len(xs) and all(len(o) == 2 for o in outs)
    for x, res in zip(xs, outs):
        ref = model.predict(x, conf=0.25, half=True)
        for a, b in zip(res, ref):
            assert torch.equal(a.boxes.data.cpu(), b.boxes.data.cpu())


def test_predict_classes_and_agnostic_options(E):
    """predict(classes=..., agnostic_nms=...) (reference cfg/default.yaml:51-65 -> ops.non_max_suppression, detect/predict.py:25-32):
    the options reach the device NMS; rows bit-exact vs the oracle NMS with the same options on the same `pred`."""
    model = E.YOLO("yolo11n-test.yaml")
    model.model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in model.model.state_dict().items()}))
    x = synth.synth_images(2, 128, 128)
    base = model.predict(x, conf=0.25, iou=0.7, device="cuda:0")
    pred = model.model(model.predictor.preprocess(x))[0].cpu().numpy()  # the reference-layout tensor (the predictor itself never writes it)
    for kw, okw in (({"classes": [0, 2, 5, 11]}, {"classes": [0, 2, 5, 11]}), ({"agnostic_nms": True}, {"agnostic": True}),
                    ({"classes": [1, 3], "agnostic_nms": True, "max_det": 7}, {"classes": [1, 3], "agnostic": True, "max_det": 7})):
        res = model.predict(x, conf=0.25, iou=0.7, device="cuda:0", **kw)
        want = onms.non_max_suppression(pred, 0.25, 0.7, **okw)
        for r, w in zip(res, want):
            d = r.boxes.data.cpu().numpy()
            w = w.copy()
            w[:, :4] = np.clip(w[:, :4], 0, 128)  # tensor source: clip_boxes to the input shape (detect/predict.py:36-38 scale_boxes -> clip)
            np.testing.assert_array_equal(d, w)
            if "classes" in kw:
                assert set(d[:, 5].astype(int).tolist()) <= set(kw["classes"])
    assert sum(len(r) for r in base) > 0


def test_save_load_predict_roundtrip(E, tmp_path):
    """predict -> save -> YOLO(path) -> predict gives identical boxes (predict() folds BatchNorm in place; the checkpoint records it)."""
    model = E.YOLO("yolo11n-test.yaml", nc=10)  # GC10-DET class count
    model.model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in model.model.state_dict().items()}))
    x = synth.synth_images(2, 128, 160)
    r1 = model.predict(x, conf=0.25, half=True, device="cuda:0")
    f = str(tmp_path / "after_predict.pt")
    model.save(f)
    again = E.YOLO(f)
    assert again.model.model[-1].nc == 10
    r2 = again.predict(x, conf=0.25, half=True, device="cuda:0")
    for a, b in zip(r1, r2):
        assert torch.equal(a.boxes.data.cpu(), b.boxes.data.cpu()) and a.boxes.data.shape[0] > 0


def test_wrong_current_device_is_an_error_not_a_fault(E):
    """Kernels are launched on the CURRENT device's stream: an operand on another device must raise before any launch
    (with one visible GPU the guard is exercised through the device index check itself)."""
    from edge_yolo_amd import _lib as L
    x = torch.zeros(1, 8, 4, 4, device="cuda:0")
    L.require_device(x, "test")  # current device == operand device: fine
    if torch.cuda.device_count() > 1:
        with torch.cuda.device(1), pytest.raises(L.HipLibraryError):
            L.require_device(x, "test")


@pytest.mark.parametrize("name,nc", [("yolo11n-test.yaml", 80), ("yolo11n.yaml", 80), ("yolo11n-test.yaml", 10)])
def test_fused_candidates_equal_nms_on_pred(E, name, nc):
    """Head decode with the NMS candidate build fused in (ey_head_decode_levels_nms + ey_nms_candidates, what predict() runs) returns
    bit-identical rows / counts / anchor indices to ey_nms on the materialised `pred`, with and without a class filter, and the
    optional `pred` output equals the plain decode's."""
    from edge_yolo_amd.nn.tasks import DetectionModel
    from edge_yolo_amd.utils import ops
    m = DetectionModel(name, nc=nc)
    m.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    m = m.cuda().fuse().half().eval()
    x = synth.synth_images(3, 160, 224).cuda().half()
    pred = m(x)[0]
    for classes in (None, [0, 3, 7]):
        for conf in (0.25, 0.05):
            cand, _ = m(x, head_nms={"conf": conf, "classes": classes, "keep_pred": True})
            assert torch.equal(cand.pred, pred)
            got = ops.nms_device(cand, conf, 0.6, classes=classes, max_det=100)
            want = ops.nms_device(pred, conf, 0.6, classes=classes, max_det=100)
            for g, w in zip(got, want):
                assert torch.equal(g, w)
            lean, _ = m(x, head_nms={"conf": conf, "classes": classes})  # no pred written
            assert lean.pred is None
            for g, w in zip(ops.nms_device(lean, conf, 0.6, classes=classes, max_det=100, agnostic=True), ops.nms_device(pred, conf, 0.6, classes=classes, max_det=100, agnostic=True)):
                assert torch.equal(g, w)
    with pytest.raises(ValueError):
        ops.nms_device(cand, 0.3, 0.6)  # candidates were built for another threshold


@pytest.mark.parametrize("half,tol", [(False, 0.02), (True, 0.06)])
def test_validator_end_to_end_vs_reference(E, golden_dir, half, tol):
    """§8(f-1) end to end on the GPU: HIP model -> validation-mode NMS on the device -> DetectionValidator (label / ratio_pad scaling,
    per-image stats, AP) against the mAP the REFERENCE got for the same images and labels with ITS model, NMS and validator
    (tests/golden/validator_case.npz).  fp32 predictions agree to ~1e-3, so the statistics may differ by a few borderline matches."""
    from edge_yolo_amd.engine.validator import DetectionValidator, KEYS
    g = np.load(os.path.join(golden_dir, "validator_case.npz"))
    m, _ = _build(E, "yolo11n-test.yaml", torch.float16 if half else torch.float32)
    B = len(g["ori_shape"])
    batch = {"img": synth.synth_images(B, 128, 160, seed=9), "cls": g["cls"], "bboxes": g["bboxes"], "batch_idx": g["batch_idx"],
             "ori_shape": [tuple(s) for s in g["ori_shape"]],
             "ratio_pad": [((float(a), float(a)), (int(p[0]), int(p[1]))) for a, p in zip(g["ratio_gain"], g["ratio_padwh"])]}
    v = DetectionValidator(m, half=half)
    res = v([batch])
    ref = dict(zip(list(g["keys"]), g["values"]))
    print(f"\\n[validator half={half}] ours {[round(res[k], 4) for k in KEYS]} reference {[round(float(ref[k]), 4) for k in KEYS]}")
    assert v.seen == int(g["seen"])
    np.testing.assert_array_equal(v.nt_per_class, g["nt_per_class"])
    for k in (KEYS[2], KEYS[3], KEYS[4], KEYS[1]):
        assert abs(res[k] - float(ref[k])) <= tol, (k, res[k], float(ref[k]))
    if not half:  # same number of predictions per image as the reference produced
        assert [len(t) for t in v.stats["tp"]] == [len(g[f"pred{i}"]) for i in range(B)]
        assert abs(int(np.concatenate(v.stats["tp"]).sum()) - int(g["tp"].sum())) <= 6


@pytest.mark.parametrize("name,tag", [("yolo11l-test.yaml", "edgeline_l_64"), ("yolo11x.yaml", "yolo11x_64")])
def test_fp32_large_scales_vs_reference_golden(E, golden_dir, name, tag):
    """l / x scales (c3k=True DSC3k / C3k inner blocks, 2 repeats per stage, up to 512-channel maps, 8-head attention at x)
    on the GPU against the REFERENCE's forward (tests/golden, fp32, north-star bar 1e-3)."""
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    m, _ = _build(E, name, torch.float32, gain=1.0)  # make_golden.py LARGE_GAIN
    y, raw = m(synth.synth_images(1, 64, 64).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=1e-4, atol=1e-3)
    for i, r in enumerate(raw):
        np.testing.assert_allclose(r.float().cpu().numpy(), g[f"raw{i}"], rtol=1e-4, atol=3e-4)
    yh, _ = _build(E, name, torch.float16, gain=1.0)[0](synth.synth_images(1, 64, 64).cuda().half())  # the f16 kernels take these widths too
    assert float((yh[:, 4:].cpu() - torch.tensor(g["y"][:, 4:])).abs().max()) < 2e-2


def test_nc10_model_and_chain_vs_oracle(E, cfg_dir):
    """GC10-DET class count (nc = 10: c3 = 64, head.py:59): fp32 vs the oracle, and the f16 class tower through the 64 -> 64 -> 10
    register chain (ey_conv_pw_chain narrow shape, scalar channel tail) vs the same model with the chain off."""
    from edge_yolo_amd.nn.tasks import DetectionModel
    from edge_yolo_amd import profiling
    name = "yolo11n-test.yaml"
    m = DetectionModel(name, nc=10)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict(sd)
    x = synth.synth_images(2, 160, 160)
    want, _ = om.OracleModel(os.path.join(cfg_dir, name), sd, nc=10)(x)
    m32 = m.cuda().fuse().float().eval()
    y32, _ = m32(x.cuda())
    assert tuple(y32.shape) == (2, 14, 525)
    np.testing.assert_allclose(y32.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-3)
    mh = m32.half()
    with profiling.trace() as t:
        y_chain, _ = mh(x.cuda().half())
    assert sum(r[0] == "conv_pw2_kernel" for r in t.records) == 3, "the narrow class-tower chain did not run"
    mh.model[-1].chain = False
    y_plain, _ = mh(x.cuda().half())
    torch.cuda.synchronize()
    assert float((y_chain[:, 4:] - y_plain[:, 4:]).abs().max()) < 2e-3 and float((y_chain[:, :4] - y_plain[:, :4]).abs().max()) < 0.25
    assert float((y_chain[:, 4:].cpu() - want[:, 4:]).abs().max()) < 2e-2


def _e2e_yaml():
    from edge_yolo_amd.nn.tasks import yaml_model_load
    d = yaml_model_load("yolo11n-test.yaml")
    d["head"][-1][2] = "E2EDetect"
    return d


def test_e2e_detect_vs_reference_golden(E, golden_dir):
    """SURVEY 8f-4, NMS-free head: E2EDetect (reference head.py:273-298,799-824) on the EdgeLine-n graph against the REFERENCE's forward
    (tests/golden/e2e_128x160.npz: its y (B,300,6) top-k rows, the raw maps of both branches, and the end-to-end branch of
    non_max_suppression).  fp32 bar 1e-3 on boxes / scores; the row order may differ only where two scores sit within fp32 noise."""
    from edge_yolo_amd.nn.tasks import DetectionModel
    from edge_yolo_amd.utils import ops as uops
    g = np.load(os.path.join(golden_dir, "e2e_128x160.npz"))
    m = DetectionModel(_e2e_yaml())
    assert m.end2end and len(m.state_dict()) == int(g["nkeys"]) and sum(p.numel() for p in m.parameters()) == int(g["params"])
    for k, v in json.loads(str(g["state_shapes"])).items():
        assert tuple(m.state_dict()[k].shape) == tuple(v), k
    assert [float(s) for s in m.stride] == [float(s) for s in g["stride"]]
    m.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    m = m.cuda().fuse().float().eval()
    m.model[-1].one2many_in_inference = True
    x = synth.synth_images(2, 128, 160, seed=3).cuda()
    y, aux = m(x)
    assert tuple(y.shape) == (2, 300, 6)
    for br in ("one2one", "one2many"):
        for i, r in enumerate(aux[br]):
            np.testing.assert_allclose(r.float().cpu().numpy(), g[f"{br}{i}"], rtol=1e-4, atol=3e-4)
    y = y.cpu().numpy()
    np.testing.assert_allclose(y[..., 4], g["y"][..., 4], rtol=1e-4, atol=1e-5)
    assert (np.diff(y[..., 4], axis=1) <= 0).all()
    same = y[..., 5] == g["y"][..., 5]
    assert same.mean() > 0.99
    np.testing.assert_allclose(y[..., :4][same], g["y"][..., :4][same], rtol=1e-4, atol=1e-3)
    # the predict-time filter (ops.py:224-228) on the reference's own rows: exact
    gy = torch.tensor(g["y"]).cuda()
    for conf, md, cl, key in ((0.25, 300, None, "det"), (0.05, 20, [int(c) for c in g["detc_classes"]], "detc")):
        out = uops.non_max_suppression(gy, conf, 0.7, classes=cl, max_det=md)
        for i in range(2):
            np.testing.assert_array_equal(out[i].cpu().numpy(), g[f"{key}{i}"])
    # f16 throughput mode, and the top-k kernel against the oracle's two-pass selection on the SAME prediction tensor (bit-exact rows)
    mh = m.half()
    yh, _ = mh(x.half())
    assert float((yh[..., 4].float().cpu() - torch.tensor(g["y"][..., 4])).abs().max()) < 2e-2
    from edge_yolo_amd.nn import _ops
    pred = synth.synth_pred(3, 80, 8400, seed=4, dense=True).cuda()
    rows, idx = _ops.e2e_topk(pred, 300, want_index=True)
    want = om.e2e_postprocess(pred.cpu().permute(0, 2, 1), 300, 80)
    np.testing.assert_array_equal(rows.cpu().numpy(), want.numpy())
    np.testing.assert_array_equal(pred.cpu().numpy()[np.arange(3)[:, None], :4, idx.cpu().numpy()], want.numpy()[..., :4])
    rows1 = _ops.e2e_topk(pred[:, :5].contiguous(), 300)  # nc = 1
    np.testing.assert_array_equal(rows1.cpu().numpy(), om.e2e_postprocess(pred[:, :5].cpu().permute(0, 2, 1), 300, 1).numpy())


def test_e2e_predict_surface(E):
    """YOLO(...).predict() on an end2end model: no NMS stage, rows = the head's top-k filtered by conf / classes / max_det; graph replay
    equals eager."""
    model = E.YOLO(_e2e_yaml())
    model.model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in model.model.state_dict().items()}))
    x = synth.synth_images(2, 128, 160, seed=3)
    r1 = model.predict(x, conf=0.3, iou=0.7, device="cuda:0", graph=False, max_det=50)
    r2 = model.predict(x, conf=0.3, iou=0.7, device="cuda:0", graph=True, max_det=50)
    y, _ = model.model.float()(x.cuda())
    for i, (a, b) in enumerate(zip(r1, r2)):
        np.testing.assert_array_equal(a.boxes.data.cpu().numpy(), b.boxes.data.cpu().numpy())
        want = onms.non_max_suppression(y[i:i + 1].cpu().numpy(), 0.3, 0.7, max_det=50)[0]
        assert len(a.boxes.data) == len(want) and 0 < len(want) <= 50
        np.testing.assert_allclose(a.boxes.data.cpu().numpy()[:, 4], want[:, 4], atol=2e-2)


def test_predict_batches_uint8_image_batches(E):
    """predict_batches on decoded image batches (uint8 (B,h,w,3) BGR): raw bytes over PCIe, LetterBox + BGR->RGB + CHW + /255 in ONE
    device launch (ey_letterbox_batch), boxes scaled back to the image frame.  (a) network-shaped images == the float-tensor path on the
    converted batch, bit for bit; (b) odd-sized images == predict() on the list of ndarrays (per-image ey_letterbox path)."""
    model = E.YOLO("yolo11n-test.yaml")
    model.model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in model.model.state_dict().items()}))
    g = torch.Generator().manual_seed(11)
    for h, w in ((128, 160), (120, 150)):
        batches = [torch.randint(0, 256, (3, h, w, 3), generator=g, dtype=torch.uint8) for _ in range(5)]
        got = list(model.predict_batches(iter([b.pin_memory() for b in batches]), conf=0.25, iou=0.7, device="cuda:0", imgsz=160))
        assert len(got) == 5
        for b, res in zip(batches, got):
            if (h, w) == (128, 160):
                x = b.flip(-1).permute(0, 3, 1, 2).float() / 255.0
                want = model.predict(x, conf=0.25, iou=0.7, device="cuda:0", graph=False)
            else:
                want = model.predict([im.numpy() for im in b], conf=0.25, iou=0.7, device="cuda:0", graph=False, imgsz=160)
            assert len(res) == len(want) == 3
            for r, wv in zip(res, want):
                np.testing.assert_array_equal(r.boxes.data.cpu().numpy(), wv.boxes.data.cpu().numpy())
                assert tuple(r.orig_shape[:2]) == (h, w)


def test_predict_option_changes_rebuild_cleanly(E):
    """Changing predict() options replaces the predictor (and its captured graph); alternating graph / eager / other thresholds, with a
    pipeline generator in between, must neither crash (a graph finalised during another capture) nor change results."""
    model = E.YOLO("yolo11n-test.yaml")
    model.model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in model.model.state_dict().items()}))
    x = synth.synth_images(2, 128, 160, seed=5)
    ref = [r.boxes.data.cpu() for r in model.predict(x, conf=0.25, half=True, graph=False)]
    for k in range(3):
        a = model.predict(x, conf=0.25, half=True, graph=True)
        b = model.predict(x, conf=0.25, half=True, graph=False)
        c = list(model.predict_batches([x, x, x], conf=0.25, half=True))[-1]
        d = model.predict(x, conf=0.3 + 0.01 * k, half=True, graph=True)
        for i in range(2):
            assert torch.equal(a[i].boxes.data.cpu(), ref[i]) and torch.equal(b[i].boxes.data.cpu(), ref[i]) and torch.equal(c[i].boxes.data.cpu(), ref[i])
            assert len(d[i]) <= len(ref[i])
