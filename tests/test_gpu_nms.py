"""-m gpu: ey_nms (through utils.ops.non_max_suppression) vs the reference-produced goldens and the CPU oracle.
Kept rows and anchor indices must be BIT-exact."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nms as onms
import synthdata as synth  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd.utils import ops
    return ops


def _preds(g):
    return {
        "sparse": synth.synth_pred(2, 80, 8400, seed=2), "dense": synth.synth_pred(1, 80, 8400, seed=3, dense=True),
        "nc10": synth.synth_pred(1, 10, 336, seed=5, imgsz=128, dense=True),
        "val_multilabel": synth.synth_pred(1, 80, 2100, seed=4, dense=True),
        "none_pass": synth.synth_pred(2, 80, 336, seed=6) * torch.tensor(1e-3), "hand": torch.tensor(g["hand_pred"]),
    }


def test_golden_cases(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "nms_cases.npz"))
    meta = json.load(open(os.path.join(golden_dir, "nms_cases.json")))
    preds = _preds(g)
    for tag, m in meta.items():
        base = "sparse" if tag.startswith("sparse") else "hand" if tag.startswith("hand") else tag
        out = ops.non_max_suppression(preds[base].cuda(), **m["kw"])
        assert [int(o.shape[0]) for o in out] == m["n"], tag
        for i, o in enumerate(out):
            np.testing.assert_array_equal(o.cpu().numpy(), g[f"{tag}_{i}"], err_msg=tag)


@pytest.mark.parametrize("B,nc,A,dense,conf,iou,max_det", [(4, 80, 8400, False, 0.25, 0.7, 300), (2, 80, 8400, True, 0.25, 0.45, 300),
                                                         (3, 3, 1000, True, 0.1, 0.5, 50), (1, 80, 33600, False, 0.25, 0.7, 300),
                                                         (2, 1, 64, True, 0.3, 0.6, 10), (2, 80, 20000, True, 0.5, 0.7, 100)])
def test_vs_oracle_indices(ops, B, nc, A, dense, conf, iou, max_det):
    pred = synth.synth_pred(B, nc, A, seed=11, dense=dense)
    want, widx = onms.non_max_suppression(pred.numpy(), conf, iou, max_det=max_det, return_idx=True)
    boxes, count, index = ops.nms_device(pred.cuda(), conf, iou, max_det=max_det)
    count = count.cpu().numpy()
    for b in range(B):
        n = int(count[b])
        assert n == want[b].shape[0]
        np.testing.assert_array_equal(index[b, :n].cpu().numpy(), widx[b])  # indices bit-exact
        np.testing.assert_array_equal(boxes[b, :n].cpu().numpy(), want[b])  # rows bit-exact
        assert float(boxes[b, n:].abs().sum()) == 0.0


def test_score_ties_and_pow2_edge(ops):
    """All scores equal: order must be ascending anchor index (stable sort); A exactly a power of two."""
    A = 256
    pred = torch.zeros(1, 4 + 2, A)
    pred[0, 0] = torch.arange(A) * 50.0 + 25  # disjoint boxes
    pred[0, 1] = 25.0
    pred[0, 2:4] = 20.0
    pred[0, 4] = 0.5
    out = ops.non_max_suppression(pred.cuda(), 0.25, 0.5, max_det=300)[0].cpu()
    want = onms.non_max_suppression(pred.numpy(), 0.25, 0.5, max_det=300)[0]
    np.testing.assert_array_equal(out.numpy(), want)
    assert out.shape[0] == A


@pytest.mark.parametrize("B,nc,A,conf,max_det", [(2, 80, 8400, 0.001, 300), (2, 3, 500, 0.05, 100), (1, 1, 300, 0.01, 50)])
def test_multi_label_vs_oracle(ops, B, nc, A, conf, max_det):
    """validation-mode NMS (reference val.py:92-102: conf 0.001, multi_label=True): up to A*nc candidates, max_nms cap 30000."""
    pred = synth.synth_pred(B, nc, A, seed=21, dense=True)
    want = onms.non_max_suppression(pred.numpy(), conf, 0.7, multi_label=True, max_det=max_det)
    out = ops.non_max_suppression(pred.cuda(), conf, 0.7, multi_label=True, max_det=max_det)
    for a, b in zip(out, want):
        np.testing.assert_array_equal(a.cpu().numpy(), b)


def test_heavy_bin_exhausted_subtree(ops):
    """More than 4096 candidates share one top 12-bit score digit (scores in [0.5, 0.53125)) and overlap heavily, so the radix descent
    opens that bin on its next digit, exhausts the subtree with far fewer than max_det boxes kept and climbs back to continue with the
    lower bins (the path whose histogram reuse needed a barrier: ADVICE r1, head_nms.hip)."""
    rng = np.random.default_rng(7)
    A, nc, nheavy = 8192, 3, 5000
    pred = np.zeros((2, 4 + nc, A), np.float32)
    for b in range(2):
        centres = rng.uniform(60, 580, (24, 2))
        which = rng.integers(0, 24, nheavy)
        pred[b, 0:2, :nheavy] = (centres[which] + rng.normal(0, 0.6, (nheavy, 2))).T  # 24 tight clusters: IoU > 0.7 inside a cluster
        pred[b, 2:4, :nheavy] = 48.0
        pred[b, 4, :nheavy] = rng.uniform(0.5, 0.53124, nheavy)
        rest = A - nheavy
        pred[b, 0:2, nheavy:] = rng.uniform(0, 640, (2, rest))
        pred[b, 2:4, nheavy:] = rng.lognormal(np.log(20.0), 0.3, (2, rest))
        pred[b, 4 + (b % nc), nheavy:] = rng.uniform(0.26, 0.49, rest)
    for max_det in (300, 40):
        want, widx = onms.non_max_suppression(pred, 0.25, 0.7, max_det=max_det, return_idx=True)
        boxes, count, index = ops.nms_device(torch.tensor(pred).cuda(), 0.25, 0.7, max_det=max_det)
        for b in range(2):
            n = int(count[b])
            assert n == want[b].shape[0] and (max_det == 40 or n > 24)
            np.testing.assert_array_equal(index[b, :n].cpu().numpy(), widx[b])
            np.testing.assert_array_equal(boxes[b, :n].cpu().numpy(), want[b])


@pytest.mark.parametrize("max_wh", [100.0, 700.0, 7680.0])
def test_class_partitioned_greedy_and_its_fallback(ops, max_wh):
    """Per-class NMS runs class-partitioned (wave w resolves the classes c % 16 == w with no workgroup barrier).  That is only valid
    while boxes of different classes cannot overlap after the c * max_wh offset (ops.py:289); with a small max_wh they do overlap in the
    reference, and the kernel must notice and fall back to the cooperative scan.  Heavy same-class overlap (many chunks, persistent
    per-owner lists), 40 classes (several classes per wave), rows and indices bit-exact in all three regimes."""
    rng = np.random.default_rng(11)
    B, nc, A = 3, 40, 6000
    pred = np.zeros((B, 4 + nc, A), np.float32)
    for b in range(B):
        centres = rng.uniform(40, 600, (150, 2))
        which = rng.integers(0, 150, A)
        pred[b, 0:2] = (centres[which] + rng.normal(0, 4.0, (A, 2))).T
        pred[b, 2:4] = rng.lognormal(np.log(50.0), 0.25, (2, A))
        cls = (which * 7 + rng.integers(0, 2, A)) % nc  # a cluster holds two classes
        pred[b, 4 + cls, np.arange(A)] = rng.uniform(0.26, 0.99, A)
    want, widx = onms.non_max_suppression(pred, 0.25, 0.6, max_det=300, max_wh=max_wh, return_idx=True)
    boxes, count, index = ops.nms_device(torch.tensor(pred).cuda(), 0.25, 0.6, max_det=300, max_wh=max_wh)
    for b in range(B):
        n = int(count[b])
        assert n == want[b].shape[0] and n > 50
        np.testing.assert_array_equal(index[b, :n].cpu().numpy(), widx[b])
        np.testing.assert_array_equal(boxes[b, :n].cpu().numpy(), want[b])


def test_argument_errors(ops):
    pred = synth.synth_pred(1, 4, 64, seed=1).cuda()
    with pytest.raises(AssertionError):
        ops.non_max_suppression(pred, conf_thres=1.5)
    with pytest.raises(NotImplementedError):
        ops.non_max_suppression(pred, rotated=True)
