"""Validation loop for the detect task, mirroring the reference's models/yolo/detect/val.py (`DetectionValidator.preprocess` :52-66,
`postprocess` :92-102, `_prepare_batch` :104-115, `_prepare_pred` :117-123, `update_metrics` :125-172, `get_stats` :179-188,
`_process_batch` :209-228) and engine/validator.py (`__call__` loop :107-218, `match_predictions` :222-262).

Device work = the forward on the HIP path + validation-mode NMS (conf 0.001, multi_label, iou 0.7, max_det 300: `ey_nms`); label
scaling, TP matching at the 10 IoU thresholds and AP run on the host in numpy (the reference runs them on the CPU too), through
`utils/metrics.py`.  Same predictions -> same statistics -> same mAP as the reference (pinned by tests/golden/validator_case.npz,
produced by the reference's own update_metrics / get_stats / DetMetrics)."""
import numpy as np
import torch

from ..utils import metrics, ops

KEYS = ["metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP75(B)", "metrics/mAP50-95(B)"]  # reference DetMetrics.keys (metrics.py:866-868)


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


class DetectionValidator:
    def __init__(self, model, conf=0.001, iou=0.7, max_det=300, half=False, single_cls=False, agnostic_nms=False, device=None):
        """model: a DetectionModel on its device (fused / dtype set by the caller or by __call__), or a YOLO facade.
        conf None -> 0.001 like the reference (engine/validator.py:100-101)."""
        self.model = getattr(model, "model", model) if not hasattr(model, "forward_layers") else model
        self.conf = 0.001 if conf is None else conf
        self.iou, self.max_det, self.half, self.single_cls, self.agnostic_nms = iou, max_det, half, single_cls, agnostic_nms
        self.device = torch.device(device) if device is not None else next(self.model.parameters()).device
        self.iouv = metrics.IOUV
        self.niou = len(self.iouv)
        self.nc = len(self.model.names)
        self.init_metrics()

    # ---- reference val.py:68-87
    def init_metrics(self):
        self.seen = 0
        self.stats = dict(tp=[], conf=[], pred_cls=[], target_cls=[], target_img=[])
        self.results_dict = dict(zip(KEYS + ["fitness"], [0.0] * 6))
        self.box = None

    # ---- reference val.py:52-66
    def preprocess(self, batch):
        """batch["img"]: (B,3,H,W) uint8 (0..255) or float in [0,1] tensor; the uint8 form is divided by 255 like the reference does."""
        img = batch["img"].to(self.device, non_blocking=True)
        scale = 255.0 if img.dtype == torch.uint8 else 1.0
        img = img.half() if self.half else img.float()
        batch = dict(batch)
        batch["img"] = (img / scale if scale != 1.0 else img).contiguous()
        return batch

    # ---- reference val.py:92-102
    def postprocess(self, preds):
        return ops.non_max_suppression(preds, self.conf, self.iou, multi_label=True, agnostic=self.single_cls or self.agnostic_nms, max_det=self.max_det)

    # ---- reference val.py:104-115: labels of image si -> native (original-image) pixel space
    def _prepare_batch(self, si, batch):
        idx = _np(batch["batch_idx"]).reshape(-1) == si
        cls = _np(batch["cls"]).reshape(-1)[idx].astype(np.float32)
        bbox = _np(batch["bboxes"]).reshape(-1, 4)[idx].astype(np.float32)
        ori_shape = tuple(int(v) for v in batch["ori_shape"][si])
        imgsz = tuple(int(v) for v in batch["img"].shape[2:])
        ratio_pad = batch["ratio_pad"][si] if batch.get("ratio_pad") is not None else None
        if len(cls):
            b = torch.from_numpy(bbox)
            b = ops.xywh2xyxy(b) * torch.tensor(imgsz, dtype=torch.float32)[[1, 0, 1, 0]]  # normalised xywh -> input pixels xyxy
            bbox = ops.scale_boxes(imgsz, b, ori_shape, ratio_pad=ratio_pad).numpy()
        return {"cls": cls, "bbox": bbox.reshape(-1, 4), "ori_shape": ori_shape, "imgsz": imgsz, "ratio_pad": ratio_pad}

    # ---- reference val.py:117-123
    def _prepare_pred(self, pred, pbatch):
        predn = torch.as_tensor(_np(pred), dtype=torch.float32).clone()
        ops.scale_boxes(pbatch["imgsz"], predn[:, :4], pbatch["ori_shape"], ratio_pad=pbatch["ratio_pad"])
        return predn.numpy()

    # ---- reference val.py:209-228
    def _process_batch(self, detections, gt_bboxes, gt_cls):
        return metrics.match_predictions(detections[:, 5], gt_cls, metrics.box_iou(gt_bboxes, detections[:, :4]), self.iouv)

    # ---- reference val.py:125-172 (plots / json / txt saving are out of scope)
    def update_metrics(self, preds, batch):
        for si, pred in enumerate(preds):
            self.seen += 1
            pred = _np(pred).astype(np.float32).reshape(-1, 6)
            npr = pred.shape[0]
            stat = dict(conf=np.zeros(0, np.float32), pred_cls=np.zeros(0, np.float32), tp=np.zeros((npr, self.niou), bool))
            pbatch = self._prepare_batch(si, batch)
            cls, bbox = pbatch.pop("cls"), pbatch.pop("bbox")
            nl = len(cls)
            stat["target_cls"] = cls
            stat["target_img"] = np.unique(cls)
            if npr == 0:
                if nl:
                    for k in self.stats:
                        self.stats[k].append(stat[k])
                continue
            if self.single_cls:
                pred = pred.copy()
                pred[:, 5] = 0
            predn = self._prepare_pred(pred, pbatch)
            stat["conf"], stat["pred_cls"] = predn[:, 4], predn[:, 5]
            if nl:
                stat["tp"] = self._process_batch(predn, bbox, cls)
            for k in self.stats:
                self.stats[k].append(stat[k])

    # ---- reference val.py:179-188 + DetMetrics.process / results_dict (metrics.py:850-896)
    def get_stats(self):
        stats = {k: (np.concatenate(v, 0) if v else np.zeros((0, self.niou) if k == "tp" else 0)) for k, v in self.stats.items()}
        self.nt_per_class = np.bincount(stats["target_cls"].astype(int), minlength=self.nc)
        self.nt_per_image = np.bincount(stats["target_img"].astype(int), minlength=self.nc)
        stats.pop("target_img", None)
        if len(stats["tp"]) and stats["tp"].any():
            r = metrics.ap_per_class(stats["tp"], stats["conf"], stats["pred_cls"], stats["target_cls"])
            ap = r["ap"]
            mean = [float(r["p"].mean()) if len(r["p"]) else 0.0, float(r["r"].mean()) if len(r["r"]) else 0.0,
                    float(ap[:, 0].mean()) if len(ap) else 0.0, float(ap[:, 5].mean()) if len(ap) else 0.0, float(ap.mean()) if len(ap) else 0.0]
            self.box = r
            # fitness: the fork weights only mAP50-95 (Metric.fitness, metrics.py:758-761: w = [0, 0, 0, 0, 1])
            self.results_dict = dict(zip(KEYS + ["fitness"], mean + [mean[4]]))
        return self.results_dict

    @torch.no_grad()
    def __call__(self, dataloader):
        """dataloader: iterable of batch dicts {"img", "cls" (N,1), "bboxes" (N,4 normalised xywh in the network input frame),
        "batch_idx" (N,), "ori_shape" [B x (h,w)], "ratio_pad" [B x ((gain,gain),(padw,padh))] or absent} (the reference's
        dataset collate format, data/dataset.py).  Returns the results dict (reference DetMetrics.results_dict)."""
        self.init_metrics()
        for batch in dataloader:
            batch = self.preprocess(batch)
            with torch.cuda.device(self.device):
                preds = self.model(batch["img"])
                preds = self.postprocess(preds)
            self.update_metrics(preds, batch)
        return self.get_stats()

    # ---- convenience kept from round 1: labels as (m,5) [cls, x1,y1,x2,y2] in network-input pixels, no letterbox
    @torch.no_grad()
    def update(self, images, labels):
        B, _, H, W = images.shape
        cls, box, bi = [], [], []
        for i, lab in enumerate(labels):
            lab = np.asarray(lab, np.float32).reshape(-1, 5)
            xy = lab[:, 1:]
            cls.append(lab[:, :1])
            box.append(np.stack([(xy[:, 0] + xy[:, 2]) / 2 / W, (xy[:, 1] + xy[:, 3]) / 2 / H, (xy[:, 2] - xy[:, 0]) / W, (xy[:, 3] - xy[:, 1]) / H], 1))
            bi.append(np.full(len(lab), i, np.float32))
        batch = self.preprocess({"img": images, "cls": np.concatenate(cls), "bboxes": np.concatenate(box), "batch_idx": np.concatenate(bi),
                                 "ori_shape": [(H, W)] * B, "ratio_pad": None})
        with torch.cuda.device(self.device):
            preds = self.postprocess(self.model(batch["img"]))
        self.update_metrics(preds, batch)

    def results(self):
        r = self.get_stats()
        return dict(mp=r[KEYS[0]], mr=r[KEYS[1]], map50=r[KEYS[2]], map75=r[KEYS[3]], map=r[KEYS[4]])
