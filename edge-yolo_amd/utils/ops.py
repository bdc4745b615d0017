"""Post-processing of the detection path, mirroring the reference's ultralytics/utils/ops.py:
`make_divisible` (:130-143), `non_max_suppression` (:167-316), `xywh2xyxy` (:416-433), `scale_boxes` (:92-127),
`clip_boxes` (:319-338).  NMS runs in the HIP library (ey_nms); there is no torch/torchvision fallback."""
import math

import torch


def make_divisible(x, divisor):
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def xywh2xyxy(x):
    assert x.shape[-1] == 4, f"input shape last dimension expected 4 but input shape is {x.shape}"
    y = torch.empty_like(x)
    xy, wh = x[..., :2], x[..., 2:] / 2
    y[..., :2] = xy - wh
    y[..., 2:] = xy + wh
    return y


def clip_boxes(boxes, shape):
    boxes[..., 0] = boxes[..., 0].clamp(0, shape[1])
    boxes[..., 1] = boxes[..., 1].clamp(0, shape[0])
    boxes[..., 2] = boxes[..., 2].clamp(0, shape[1])
    boxes[..., 3] = boxes[..., 3].clamp(0, shape[0])
    return boxes


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None, padding=True, xywh=False):
    """Rescale xyxy boxes from the network input shape to the original image shape (reference :92-127)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    if padding:
        boxes[..., 0] -= pad[0]
        boxes[..., 1] -= pad[1]
        if not xywh:
            boxes[..., 2] -= pad[0]
            boxes[..., 3] -= pad[1]
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)


_CLASS_TENSORS = {}


def _end2end_filter(prediction, conf_thres, classes, max_det):
    """Output of an end2end head, (B, k, 6) rows [x1,y1,x2,y2,score,class] (reference ops.py:224-228): no suppression, only
    `pred[pred[:, 4] > conf][:max_det]` and then the class filter, here on fixed-size tensors (graph-capturable): kept rows are moved
    to the front in their order, the rest zeroed.  Returns (boxes (B,max_det,6), count (B,) int32, index (B,max_det) int32 = source row)."""
    p = prediction.float()
    B, K, _ = p.shape
    keep = p[..., 4] > conf_thres
    keep &= (keep.cumsum(1) - 1) < max_det
    if classes is not None:
        key = (tuple(classes), p.device)
        if key not in _CLASS_TENSORS:  # uploaded once per filter (never inside a captured graph after the first, eager, call)
            _CLASS_TENSORS[key] = torch.as_tensor(list(classes), dtype=torch.float32, device=p.device)
        keep &= (p[..., 5:6] == _CLASS_TENSORS[key]).any(-1)
    order = torch.sort((~keep).to(torch.uint8), dim=1, stable=True).indices  # kept rows first, original (descending score) order
    n = min(K, max_det)
    boxes = torch.zeros((B, max_det, 6), dtype=torch.float32, device=p.device)
    index = torch.full((B, max_det), -1, dtype=torch.int32, device=p.device)
    sel = order[:, :n]
    ok = torch.gather(keep, 1, sel)
    boxes[:, :n] = torch.gather(p, 1, sel[..., None].expand(-1, -1, 6)) * ok[..., None]
    index[:, :n] = torch.where(ok, sel.to(torch.int32), index[:, :n])
    return boxes, keep.sum(1).to(torch.int32), index


def nms_device(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, max_det=300, nc=0, max_nms=30000, max_wh=7680,
               multi_label=False):
    """Batched NMS on the device, fixed-size outputs (graph-capturable): returns (boxes (B,max_det,6), count (B,),
    index (B,max_det)).  `prediction` is (B,4+nc,A); fp16 input is promoted to fp32 first (the reference promotes inside
    NMS, ops.py:275; this build keeps head outputs in fp32 end to end)."""
    from ..nn import _ops
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if isinstance(prediction, _ops.Candidates):  # built by the fused head decode for (conf, classes): selection + suppression only
        c = prediction
        assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
        if abs(c.conf - float(conf_thres)) > 1e-12 or c.classes != (tuple(classes) if classes is not None else None):
            raise ValueError(f"nms_device: the candidates were built for conf={c.conf} classes={c.classes}, not conf={conf_thres} classes={classes}")
        if multi_label and c.nc > 1:
            raise ValueError("nms_device: fused candidates are single-label (predict mode); validation-mode NMS needs `pred`")
        if c.device.index != torch.cuda.current_device():
            with torch.cuda.device(c.device):
                return _ops.nms_candidates(c, iou_thres, max_det, max_nms, max_wh, agnostic)
        return _ops.nms_candidates(c, iou_thres, max_det, max_nms, max_wh, agnostic)
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if prediction.shape[-1] == 6:
        return _end2end_filter(prediction, conf_thres, classes, max_det)
    nc = nc or (prediction.shape[1] - 4)
    if prediction.shape[1] - nc - 4:
        raise NotImplementedError("mask/keypoint channels (nm>0) are outside the detect path")
    if prediction.is_cuda and prediction.device.index != torch.cuda.current_device():
        with torch.cuda.device(prediction.device):  # launches go to the current device's stream (_lib.stream)
            return nms_device(prediction, conf_thres, iou_thres, classes, agnostic, max_det, nc, max_nms, max_wh, multi_label)
    p = prediction.float().contiguous()
    mask = None
    if classes is not None:
        mask = torch.zeros(nc, dtype=torch.uint8)
        mask[torch.as_tensor(list(classes), dtype=torch.long)] = 1
        mask = mask.to(p.device)
    return _ops.nms(p, conf_thres, iou_thres, max_det, max_nms, max_wh, agnostic, mask, multi_label)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, labels=(),
                        max_det=300, nc=0, max_time_img=0.05, max_nms=30000, max_wh=7680, in_place=True, rotated=False):
    """Reference signature (ops.py:167-183); returns a list of (n_i, 6) tensors [x1,y1,x2,y2,conf,cls].
    multi_label=True is the validation-mode variant (one candidate per (anchor, class) above conf, ops.py:270-272).
    Differences, all deliberate: no wall-clock abort (:238,:312-314 make the reference output timing dependent);
    the input tensor is not rewritten to xyxy in place; labels (autolabel) / rotated are not built and raise."""
    if (labels and len(labels)) or rotated:
        raise NotImplementedError("autolabel / rotated NMS variants are not part of the built path")
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    boxes, count, _ = nms_device(prediction, conf_thres, iou_thres, classes, agnostic, max_det, nc, max_nms, max_wh, multi_label)
    n = count.tolist()  # one D2H sync for the whole batch
    return [boxes[i, : n[i]] for i in range(len(n))]
