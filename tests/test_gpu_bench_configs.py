"""-m gpu: the EXACT configurations bench.py measures (BASELINE.json configs C2, C3 and the per-GPU share of C5: f16 storage, batch
32 at 640x640 / batch 8 at 1280x1280, weights and model construction through bench.build_model) against the CPU oracle.

Kernel dispatch depends on M = B*Ho*Wo, dtype and map size, so these shapes reach instantiations and persistent-grid sizes the small
parity tests never launch.  Per configuration:
  * `pred` of several images of the batch (first and last included) vs the fp32 oracle on the same (f16-rounded) input;
  * NMS rows + anchor indices of the HIP path BIT-exact vs the oracle NMS run on the HIP `pred`;
  * a detection-level agreement measure between the f16 HIP path and the fp32 oracle end to end (post-NMS boxes matched at
    IoU >= 0.9, same class, |score difference| <= 0.02) -- the only mAP proxy available without trained weights / a dataset;
  * the pipelined execution bench.py times returns the same boxes as direct execution.
Tolerances are measured bounds (printed by the test) with head-room, not guesses: see TOL below."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model as om, nms as onms  # noqa: E402

# f16 storage through ~100 layers: measured max |score - oracle| and max box error (pixels, relative to the image size) per config
# on MI355X are printed by the test; the asserted bounds leave ~2x head-room.
TOL = {"score": 1.5e-2, "box_frac": 6e-3, "match_floor": 0.85}


def _iou(a, b):
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:4], b[None, :, 2:4])
    inter = np.clip(rb - lt, 0, None).prod(-1)
    area = lambda z: (z[:, 2] - z[:, 0]) * (z[:, 3] - z[:, 1])  # noqa: E731
    return inter / (area(a)[:, None] + area(b)[None, :] - inter + 1e-9)


def matched_rate(det, ref, iou_thr=0.9, score_thr=0.02):
    """fraction of reference post-NMS rows that have a HIP row of the same class with IoU >= iou_thr and |dscore| <= score_thr."""
    if len(ref) == 0:
        return 1.0 if len(det) == 0 else 0.0
    if len(det) == 0:
        return 0.0
    ok = (_iou(ref, det) >= iou_thr) & (ref[:, None, 5] == det[None, :, 5]) & (np.abs(ref[:, None, 4] - det[None, :, 4]) <= score_thr)
    return float(ok.any(1).mean())


@pytest.mark.parametrize("name,imgsz,B,check", [("yolo11n-test.yaml", 640, 32, (0, 1, 15, 31)), ("yolo11n.yaml", 640, 32, (0, 17, 31)),
                                                ("yolo11n-test.yaml", 1280, 8, (0, 7))])
def test_bench_config_f16_vs_oracle(cfg_dir, name, imgsz, B, check):
    import bench
    from edge_yolo_amd.utils import ops
    dev = torch.device("cuda", 0)
    model, sd = bench.build_model(name, torch.float16, dev)
    g = torch.Generator(device=dev).manual_seed(0)
    images = torch.rand(B, 3, imgsz, imgsz, generator=g, device=dev).to(torch.float16)  # bench.py's rank-0 batch
    pred, _ = model(images)
    boxes, count, index = ops.nms_device(pred, 0.25, 0.7, max_det=300)
    torch.cuda.synchronize()
    pred_h, boxes_h, count_h, index_h = pred.cpu().numpy(), boxes.cpu().numpy(), count.cpu().numpy(), index.cpu().numpy()
    oracle = om.OracleModel(os.path.join(cfg_dir, name), {k: v.float() for k, v in sd.items()})
    idx = list(check)
    want, _ = oracle(images[idx].float().cpu())
    want = want.numpy()
    got = pred_h[idx]
    ds = float(np.abs(got[:, 4:] - want[:, 4:]).max())
    db = float(np.abs(got[:, :4] - want[:, :4]).max())
    rates = []
    for k, i in enumerate(idx):
        # NMS of the HIP path on ITS pred: rows and anchor indices bit-exact vs the oracle NMS on the same tensor
        ref_rows, ref_idx = onms.non_max_suppression(pred_h[i:i + 1], 0.25, 0.7, max_det=300, return_idx=True)
        n = int(count_h[i])
        assert n == ref_rows[0].shape[0]
        np.testing.assert_array_equal(index_h[i, :n], ref_idx[0])
        np.testing.assert_array_equal(boxes_h[i, :n], ref_rows[0])
        # end-to-end agreement with the fp32 oracle (its own pred -> its own NMS)
        o_rows = onms.non_max_suppression(want[k:k + 1], 0.25, 0.7, max_det=300)[0]
        rates.append(matched_rate(boxes_h[i, :n], o_rows))
    print(f"\n[{name} {imgsz} B{B}] max|dscore|={ds:.2e} max|dbox|={db:.3f}px ({db / imgsz:.2e} of the image) matched-box rate per image={['%.3f' % r for r in rates]}")
    assert ds < TOL["score"], f"scores differ from the fp32 oracle by {ds}"
    assert db < TOL["box_frac"] * imgsz, f"boxes differ from the fp32 oracle by {db} px"
    assert min(rates) >= TOL["match_floor"], f"matched-box rate {rates} below the floor"


def test_bench_pipeline_matches_direct_at_c3():
    """The 4-stage batch pipeline of bench.py (cuts from bench.pipeline_cuts) at the C3 shape returns exactly the boxes of direct
    execution, for consecutive different batches."""
    import bench
    from edge_yolo_amd.engine.predictor import PipelinedRunner
    from edge_yolo_amd.utils import ops
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model("yolo11n-test.yaml", torch.float16, dev)
    head = model.model[-1]
    n = len(model.model)
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.rand(32, 3, 640, 640, generator=g, device=dev).half() for _ in range(5)]
    want = [ops.nms_device(model(x)[0], 0.25, 0.7, max_det=300) for x in xs]
    torch.cuda.synchronize()
    head.head_streams = False
    bounds = [0] + bench.pipeline_cuts(n, 4) + [n]
    stages = [(lambda st, lo=lo, hi=hi: model.forward_layers(st if lo else (st, []), lo, hi)) for lo, hi in zip(bounds[:-1], bounds[1:])]
    last = stages.pop()
    stages.append(lambda st: ops.nms_device(last(st)[0][0], 0.25, 0.7, max_det=300)[:2])
    pipe = PipelinedRunner(*stages, xs[0])
    js = [pipe.submit(x) for x in xs[:4]]  # four batches in flight
    pipe.wait()
    torch.cuda.synchronize()
    for k, j in enumerate(js):
        b, c = pipe.outputs(j)
        assert torch.equal(c, want[k][1]) and torch.equal(b, want[k][0]), f"in-flight batch {k}"
    j = pipe.submit(xs[4])
    pipe.wait(j)
    torch.cuda.synchronize()
    b, c = pipe.outputs(j)
    assert torch.equal(c, want[4][1]) and torch.equal(b, want[4][0])


def test_bench_config_sparse_regime_f16_vs_oracle(cfg_dir):
    """C3 shape (EdgeLine-n f16, 640x640, batch 32) in the SPARSE NMS regime of SURVEY.md §8(d) (`bench.py --regime sparse`: the synthetic
    weights + Detect.bias_init head biases, ultralytics/nn/modules/head.py:150-161) -- the score surface of a detector whose head was
    initialised the way a trained one starts, not the flat one of default biases.  With random backbone weights nothing reaches
    conf 0.25 there (bench reports 0 detections per image), so the detection-level comparison runs at the validation threshold
    conf = 0.001 (cfg/default.yaml val conf; ~200 candidates per image, NMS keeps a subset): every oracle box clearly above the
    threshold must have an f16 HIP box of the same class with a score within 20 % (measured worst relative score error over all
    anchors x classes: 13 %) that the NMS itself would call the same object (IoU >= iou_thres = 0.7) -- floor 0.98 over all checked rows (measured 0.985 on 333), 0.93 per image -- and at IoU >= 0.9 --
    floor 0.95 over all rows (measured 0.970), 0.85 per image: the rows that fail the strict match are suppression-order flips (two candidates of IoU 0.7-0.9 whose scores differ by
    less than the f16 noise: the kept one changes, measured IoU to the oracle's 0.70-0.88), not displaced boxes.  The measured
    |dscore| / |dbox| are printed."""
    import bench
    from edge_yolo_amd.utils import ops
    dev = torch.device("cuda", 0)
    name, imgsz, B, conf = "yolo11n-test.yaml", 640, 32, 0.001
    model, sd = bench.build_model(name, torch.float16, dev, regime="sparse")
    g = torch.Generator(device=dev).manual_seed(0)
    images = torch.rand(B, 3, imgsz, imgsz, generator=g, device=dev).to(torch.float16)
    pred, _ = model(images)
    boxes, count, index = ops.nms_device(pred, conf, 0.7, max_det=300)
    b25, c25, _ = ops.nms_device(pred, 0.25, 0.7, max_det=300)
    torch.cuda.synchronize()
    pred_h, boxes_h, count_h, index_h = pred.cpu().numpy(), boxes.cpu().numpy(), count.cpu().numpy(), index.cpu().numpy()
    idx = [0, 4, 9, 13, 20, 24, 27, 31]
    oracle = om.OracleModel(os.path.join(cfg_dir, name), {k: v.float() for k, v in sd.items()})
    want, _ = oracle(images[idx].float().cpu())
    want = want.numpy()
    got = pred_h[idx]
    ds = float(np.abs(got[:, 4:] - want[:, 4:]).max())
    rel = float((np.abs(got[:, 4:] - want[:, 4:]) / np.maximum(want[:, 4:], 1e-4)).max())
    db = float(np.abs(got[:, :4] - want[:, :4]).max())
    rates, rates_obj, nrows, hits = [], [], [], [0, 0, 0]
    for k, i in enumerate(idx):
        ref_rows, ref_idx = onms.non_max_suppression(pred_h[i:i + 1], conf, 0.7, max_det=300, return_idx=True)
        n = int(count_h[i])
        assert n == ref_rows[0].shape[0]
        np.testing.assert_array_equal(index_h[i, :n], ref_idx[0])  # NMS bit-exact on the HIP tensor in this regime too
        np.testing.assert_array_equal(boxes_h[i, :n], ref_rows[0])
        o_rows = onms.non_max_suppression(want[k:k + 1], conf, 0.7, max_det=300)[0]
        o_rows = o_rows[o_rows[:, 4] >= 1.25 * conf]  # rows within f16 noise of the threshold may legitimately fall on either side
        det = boxes_h[i, :n]
        iou_m, same = _iou(o_rows, det), o_rows[:, None, 5] == det[None, :, 5]
        close = same & (np.abs(o_rows[:, None, 4] - det[None, :, 4]) <= 0.2 * o_rows[:, None, 4])
        ok = (iou_m >= 0.9) & close
        rates.append(float(ok.any(1).mean()) if len(o_rows) else 1.0)
        rates_obj.append(float(((iou_m >= 0.7) & close).any(1).mean()) if len(o_rows) else 1.0)
        hits[0] += int(ok.any(1).sum()); hits[1] += int(((iou_m >= 0.7) & close).any(1).sum()); hits[2] += len(o_rows)
        for r in np.nonzero(~ok.any(1))[0]:  # what an unmatched oracle row looks like on the HIP side
            j = int(np.argmax(np.where(same[r], iou_m[r], -1.0)))
            print(f"  image {i}: oracle row score {o_rows[r, 4]:.5f} cls {int(o_rows[r, 5])}: best same-class HIP row IoU {iou_m[r, j]:.3f} score {det[j, 4]:.5f}")
        nrows.append((len(o_rows), n))
    print(f"\n[{name} {imgsz} B{B} sparse regime] detections/img at conf 0.25: {float(c25.float().mean()):.2f}; at conf {conf}: (oracle rows >= 1.25 conf, HIP rows) {nrows}; "
          f"max|dscore|={ds:.2e} (relative {rel:.2e}) max|dbox|={db:.3f}px ({db / imgsz:.2e} of the image) matched-box rate per image: IoU>=0.9 {['%.3f' % r for r in rates]}, same object (IoU>=0.7) {['%.3f' % r for r in rates_obj]}")
    assert all(n_o > 20 for n_o, _ in nrows), "the sparse-regime case must have boxes to compare"
    assert ds < 2e-3 and db < TOL["box_frac"] * imgsz
    print(f"  over the {len(idx)} images: {hits[2]} oracle rows, same object {hits[1] / hits[2]:.4f}, IoU>=0.9 {hits[0] / hits[2]:.4f}")
    # an image holds 25-55 rows, so ONE suppression-order flip is 2-4 % of it: the 0.98 floor is on all checked rows, 0.95 / 0.90 per image
    assert hits[1] / hits[2] >= 0.98, f"same-object rate {hits[1] / hits[2]:.4f} over {hits[2]} rows below 0.98"
    assert hits[0] / hits[2] >= 0.95, f"strict (IoU >= 0.9) rate {hits[0] / hits[2]:.4f} over {hits[2]} rows below 0.95"
    assert min(rates_obj) >= 0.93 and min(rates) >= 0.85, f"per-image rates {rates_obj} / {rates}"  # (measured minima 0.952 / 0.881)
