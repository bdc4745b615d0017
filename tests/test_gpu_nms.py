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


# ---------------------------------------------------------------------------------------------------------------------------------
# The predict-mode fast path (csrc/nms_fast.inc.h: select K best -> all-pairs suppression bit matrix -> mask-arithmetic resolve) and
# its fallback to the general kernel.  `nms_fast_k` (ey_tune_set) = K; 0 switches the fast path off.
def _set_fast_k(k):
    from edge_yolo_amd import _lib
    _lib.check(_lib.lib().ey_tune_set(b"nms_fast_k", int(k)), "nms_fast_k")


def _lib_tune(name, v):
    from edge_yolo_amd import _lib
    _lib.check(_lib.lib().ey_tune_set(name.encode(), int(v)), name)


def _fast_meta(ws, B, A):
    """(n_sel, n_total, done) per image from the workspace (layout: nms_fast.inc.h)."""
    P = (A + 255) // 256 * 256
    keys_bytes = (B * P * 12 + 255) // 256 * 256
    img = (2048 * 28 + 528 * 512 + 16 + 255) // 256 * 256
    raw = ws.cpu().numpy()
    out = []
    for b in range(B):
        off = keys_bytes + b * img + 2048 * 28 + 528 * 512
        out.append(tuple(int(v) for v in raw[off:off + 12].view(np.int32)))
    return out


def _clustered(rng, B, nc, A, nclusters, sigma, size, lo=0.26, hi=0.99, one_class=False):
    pred = np.zeros((B, 4 + nc, A), np.float32)
    for b in range(B):
        centres = rng.uniform(40, 600, (nclusters, 2))
        which = rng.integers(0, nclusters, A)
        pred[b, 0:2] = (centres[which] + rng.normal(0, sigma, (A, 2))).T
        pred[b, 2:4] = rng.lognormal(np.log(size), 0.25, (2, A))
        cls = np.zeros(A, np.int64) if one_class else (which * 7 + rng.integers(0, 2, A)) % nc
        pred[b, 4 + cls, np.arange(A)] = rng.uniform(lo, hi, A)
    return pred


@pytest.mark.parametrize("case", ["dense_one_class", "clustered", "sparse", "agnostic", "small_max_det", "a33600", "few", "classes"])
def test_fast_path_equals_general_kernel_and_oracle(ops, case):
    """Rows, counts and anchor indices of the fast path (K = 2048, 1000, 200, 64 -- the small ones force the fallback) == the general
    kernel alone (K = 0) == the CPU oracle, bit for bit; the workspace's done flags show which path finished each image."""
    from edge_yolo_amd.nn import _ops
    rng = np.random.default_rng(31)
    kw = dict(conf=0.25, iou=0.7, max_det=300, agnostic=False, classes=None)
    if case == "dense_one_class":  # the random-init model's regime: one dominant class, ~1 of 4 candidates kept, max_det reached deep in the list
        pred = _clustered(rng, 3, 80, 8400, 400, 6.0, 60.0, one_class=True)
    elif case == "clustered":
        pred = _clustered(rng, 2, 40, 6000, 150, 4.0, 50.0)
        kw["iou"] = 0.6
    elif case == "sparse":
        pred = synth.synth_pred(4, 80, 8400, seed=41).numpy()
    elif case == "agnostic":
        pred = _clustered(rng, 2, 20, 3000, 80, 5.0, 40.0)
        kw["agnostic"] = True
    elif case == "small_max_det":
        pred = _clustered(rng, 2, 10, 5000, 200, 5.0, 40.0)
        kw["max_det"] = 17
    elif case == "a33600":
        pred = synth.synth_pred(1, 80, 33600, seed=42, dense=True).numpy()
    elif case == "few":
        pred = synth.synth_pred(3, 5, 200, seed=43, imgsz=128).numpy()
        kw["conf"] = 0.05
    else:
        pred = synth.synth_pred(2, 80, 8400, seed=44, dense=True).numpy()
        kw["classes"] = [1, 5, 70]
    B, no, A = pred.shape
    want, widx = onms.non_max_suppression(pred, kw["conf"], kw["iou"], max_det=kw["max_det"], agnostic=kw["agnostic"], classes=kw["classes"], return_idx=True)
    x = torch.tensor(pred).cuda()
    mask = None
    if kw["classes"] is not None:
        mask = torch.zeros(no - 4, dtype=torch.uint8)
        mask[kw["classes"]] = 1
        mask = mask.cuda()
    ref = None
    try:
        # (K, candidates covered by the all-pairs bit matrix): beyond the matrix the resolve kernel tests candidates against the kept boxes
        # on the fly -- 512 / 1024 put most of the dense cases' work there, 2048 none
        for K, MK in ((0, 1536), (2048, 1536), (2048, 2048), (2048, 512), (2048, 1024), (1000, 512), (1000, 1536), (200, 1536), (64, 512)):
            _set_fast_k(K)
            _lib_tune("nms_mask_k", MK)
            boxes, count, index, ws = _ops.nms(x, kw["conf"], kw["iou"], kw["max_det"], 30000, 7680.0, kw["agnostic"], mask, False, return_workspace=True)
            torch.cuda.synchronize()
            got = (boxes.cpu().numpy(), count.cpu().numpy(), index.cpu().numpy())
            if ref is None:
                ref = got
                for b in range(B):
                    n = int(got[1][b])
                    assert n == want[b].shape[0]
                    np.testing.assert_array_equal(got[2][b, :n], widx[b])
                    np.testing.assert_array_equal(got[0][b, :n], want[b])
            else:
                for a, r in zip(got, ref):
                    np.testing.assert_array_equal(a, r, err_msg=f"{case}: K={K} mask_k={MK} differs from the general kernel")
                meta = _fast_meta(ws, B, A)
                for b, (nsel, ntot, done) in enumerate(meta):
                    assert nsel == min(ntot, nsel) and nsel <= K and (nsel == ntot or nsel > 0)
                    # done <=> the fast path found max_det boxes among its K candidates, or the K were all there is
                    msg = f"{case}: K={K} image {b}: n_sel={nsel} n_total={ntot} kept={int(got[1][b])} done={done}"
                    assert done in (0, 1) and (done == 1 or nsel < ntot) and (nsel < ntot or done == 1), msg
                    assert done == 0 or nsel == ntot or int(got[1][b]) >= kw["max_det"], msg
                if K == 64 and case in ("dense_one_class", "clustered", "a33600"):
                    assert not all(m[2] for m in meta), f"{case}: K=64 should leave images to the general kernel"
                if K == 2048 and case in ("sparse", "few", "classes", "small_max_det"):
                    assert all(m[2] for m in meta), f"{case}: the fast path should finish every image at K=2048 ({meta})"
    finally:
        _set_fast_k(2048)
        _lib_tune("nms_mask_k", 1536)


def test_fast_path_score_ties_across_the_selection_boundary(ops):
    """3000 candidates share ONE score (a tie group far larger than K): the threshold select cannot split it, takes only the keys above
    it, and the general kernel must finish the image; ascending-anchor order inside the tie group is kept."""
    from edge_yolo_amd.nn import _ops
    rng = np.random.default_rng(5)
    A, nc = 4096, 2
    pred = np.zeros((1, 4 + nc, A), np.float32)
    pred[0, 0] = rng.uniform(0, 640, A)
    pred[0, 1] = rng.uniform(0, 640, A)
    pred[0, 2:4] = 30.0
    pred[0, 4, :100] = rng.uniform(0.6, 0.9, 100)
    pred[0, 4, 100:3100] = 0.5
    pred[0, 5, 3100:] = rng.uniform(0.26, 0.45, A - 3100)
    want, widx = onms.non_max_suppression(pred, 0.25, 0.5, max_det=300, return_idx=True)
    try:
        for K in (2048, 512):
            _set_fast_k(K)
            boxes, count, index, ws = _ops.nms(torch.tensor(pred).cuda(), 0.25, 0.5, 300, 30000, 7680.0, False, None, False, return_workspace=True)
            n = int(count[0])
            assert n == want[0].shape[0]
            np.testing.assert_array_equal(index[0, :n].cpu().numpy(), widx[0])
            np.testing.assert_array_equal(boxes[0, :n].cpu().numpy(), want[0])
            nsel, ntot, done = _fast_meta(ws, 1, A)[0]
            assert ntot == A and nsel == 100  # only the keys above the tie group could be selected
    finally:
        _set_fast_k(2048)


def test_fast_path_iou_decisions_on_the_threshold(ops):
    """Pairs whose IoU sits within a few ulp of iou_thres: the reciprocal pre-test of iou_gt_fast must hand them to the exact division.
    400 pairs with the SAME relative geometry (width 64, shift 21: IoU = 43/85 in exact arithmetic) but different heights, so the fp32
    quotients scatter by rounding over three neighbouring floats; the middle one is the threshold: about half of the partners survive,
    and which ones is decided in the last bit."""
    rng = np.random.default_rng(9)
    n = 400
    pred = np.zeros((1, 4 + 1, 2 * n), np.float32)
    for i in range(n):
        cx, h = 200.0 * i + 100.0, float(np.float32(rng.uniform(20.0, 80.0)))
        pred[0, :4, 2 * i] = (cx, 100.0, 64.0, h)
        pred[0, :4, 2 * i + 1] = (cx + 21.0, 100.0, 64.0, h)
        pred[0, 4, 2 * i] = 0.9 - 1e-4 * i
        pred[0, 4, 2 * i + 1] = 0.8 - 1e-4 * i
    thr = float(np.nextafter(np.float32(43.0 / 85.0), np.float32(0)))  # the fp32 quotients take the values float32(43/85) - {0, 1, 2} ulp: the middle one
    want, widx = onms.non_max_suppression(pred, 0.25, thr, max_det=1024, max_wh=1e6, return_idx=True)
    boxes, count, index = ops.nms_device(torch.tensor(pred).cuda(), 0.25, thr, max_det=1024, max_wh=1e6)
    nkept = int(count[0])
    assert nkept == want[0].shape[0] and n + 100 < nkept < 2 * n - 100, nkept  # a real mix of decisions on both sides of the threshold
    np.testing.assert_array_equal(index[0, :nkept].cpu().numpy(), widx[0])
    np.testing.assert_array_equal(boxes[0, :nkept].cpu().numpy(), want[0])
