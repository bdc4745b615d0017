"""Validation loop for the detect task (reference engine/validator.py:107-218 + models/yolo/detect/val.py:92-228, reduced to
what the mAP number needs): forward on the HIP path, validation-mode NMS on the device (conf 0.001, multi_label, iou 0.7,
max_det 300), TP matching at 10 IoU thresholds and AP on the host (utils/metrics.py)."""
import torch

from ..utils import metrics, ops


class DetectionValidator:
    def __init__(self, model, conf=0.001, iou=0.7, max_det=300, half=False):
        self.model, self.conf, self.iou, self.max_det, self.half = model, conf, iou, max_det, half
        self.metrics = metrics.DetMetrics()

    @torch.no_grad()
    def update(self, images, labels):
        """images: (B,3,H,W) float in [0,1] on the model's device; labels: list of (m_i,5) arrays [cls,x1,y1,x2,y2] in input pixels."""
        im = images.half() if self.half else images.float()
        preds = self.model(im)
        pred = preds[0] if isinstance(preds, (list, tuple)) else preds
        dets = ops.non_max_suppression(pred, self.conf, self.iou, multi_label=True, max_det=self.max_det)
        for d, lab in zip(dets, labels):
            self.metrics.update(d.cpu().numpy(), lab)

    def results(self):
        return self.metrics.results()
