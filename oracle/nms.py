"""Oracle NMS: numpy fp32 restatement.  TEST INFRASTRUCTURE (see oracle/__init__.py).

`non_max_suppression` follows /root/reference/ultralytics/utils/ops.py:167-316 (single-label and
multi_label branches, class filter, max_nms cap, class offset, max_det), minus the wall-clock abort
(:238,:312-314), which makes the reference's output timing dependent and is deliberately not reproduced.

`tv_nms` restates torchvision.ops.nms (torchvision==0.17.2, requirements.txt:2; call site ops.py:296).
torchvision is NOT in the reference tree nor in this image: PARITY UNPINNED at this boundary.  Published
CPU algorithm (torchvision/csrc/ops/cpu/nms_kernel.cpp): areas=(x2-x1)*(y2-y1); order = stable sort of
scores, descending; for each unsuppressed i in order: keep i; for each later j: inter = max(0,min(x2)-max(x1))
* max(0,min(y2)-max(y1)); suppress j if inter/(area_i+area_j-inter) > thr (strict).  All fp32, one rounding
per operation, no FMA contraction.
"""
import numpy as np

F32 = np.float32


def xywh2xyxy(x):  # ops.py:416-433
    y = np.empty_like(x)
    wh = x[..., 2:] / F32(2)
    y[..., :2] = x[..., :2] - wh
    y[..., 2:] = x[..., :2] + wh
    return y


def tv_nms(boxes, scores, thr):
    boxes = np.asarray(boxes, F32)
    scores = np.asarray(scores, F32)
    n = boxes.shape[0]
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind="stable")  # descending, ties keep the lower index first
    thr = F32(thr)
    suppressed = np.zeros(n, bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(F32(0), xx2 - xx1)
        h = np.maximum(F32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, np.int64)


def non_max_suppression(pred, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        max_det=300, nc=0, max_nms=30000, max_wh=7680, return_idx=False):
    """pred: (B, 4+nc, A) float32 array -> list of (n_i, 6) float32 arrays [x1,y1,x2,y2,conf,cls].
    With return_idx=True also returns, per image, the anchor index of every kept row (single-label only)."""
    pred = np.asarray(pred, F32)
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1  # ops.py:217-218
    if pred.shape[-1] == 6:  # ops.py:224-228: output of an end-to-end head (B, k, 6) -- confidence / class filter only, no suppression
        out = [p[p[:, 4] > F32(conf_thres)][:max_det] for p in pred]
        if classes is not None:
            out = [p[np.isin(p[:, 5], np.asarray(classes, F32))] for p in out]
        return out
    bs = pred.shape[0]
    nc = nc or pred.shape[1] - 4
    mi = 4 + nc
    conf_thres = F32(conf_thres)
    xc = pred[:, 4:mi].max(1) > conf_thres  # ops.py:234
    multi_label = multi_label and nc > 1
    p = pred.transpose(0, 2, 1).copy()
    p[..., :4] = xywh2xyxy(p[..., :4])  # ops.py:244
    out = [np.zeros((0, 6), F32)] * bs
    idx_out = [np.zeros((0,), np.int64)] * bs
    for xi in range(bs):
        anchors = np.nonzero(xc[xi])[0]
        x = p[xi][anchors]  # ops.py:253
        if not x.shape[0]:
            continue
        box, cls = x[:, :4], x[:, 4:mi]
        if multi_label:  # ops.py:270-272
            i, j = np.nonzero(cls > conf_thres)
            x = np.concatenate((box[i], x[i, 4 + j, None], j[:, None].astype(F32)), 1)
            anchors = anchors[i]
        else:  # ops.py:273-275  (max(1): first maximal index)
            j = cls.argmax(1)
            conf = cls[np.arange(len(j)), j]
            x = np.concatenate((box, conf[:, None], j[:, None].astype(F32)), 1)
            m = conf > conf_thres
            x, anchors = x[m], anchors[m]
        if classes is not None:  # ops.py:278-279
            m = np.isin(x[:, 5], np.asarray(classes, F32))
            x, anchors = x[m], anchors[m]
        n = x.shape[0]
        if not n:
            continue
        if n > max_nms:  # ops.py:285-286 (reference argsort is not stable; ties there are implementation defined)
            o = np.argsort(-x[:, 4], kind="stable")[:max_nms]
            x, anchors = x[o], anchors[o]
        c = x[:, 5:6] * F32(0 if agnostic else max_wh)  # ops.py:289
        keep = tv_nms(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]  # ops.py:295-297
        out[xi] = x[keep]
        idx_out[xi] = anchors[keep]
    return (out, idx_out) if return_idx else out
