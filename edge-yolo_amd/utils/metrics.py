"""Detection validation metrics on the host (numpy), mirroring the reference's definitions so that the same predictions
give the same mAP: `box_iou` (utils/metrics.py:52-72), `match_predictions` (engine/validator.py:222-262, the default
non-scipy branch), `compute_ap` / `ap_per_class` (utils/metrics.py:505-623), `smooth` (:494-502), and the fitness-style
summary of `DetMetrics` (:808).  The reference runs these on the CPU as well; only NMS (validation mode: conf 0.001,
multi_label, models/yolo/detect/val.py:92-102) is device work and goes through `ey_nms`."""
import numpy as np


def box_iou(box1, box2, eps=1e-7):
    """(N,4) x (M,4) xyxy -> (N,M) IoU, float32 arithmetic like the reference."""
    b1 = np.asarray(box1, np.float32)[:, None, :]
    b2 = np.asarray(box2, np.float32)[None, :, :]
    wh = np.clip(np.minimum(b1[..., 2:], b2[..., 2:]) - np.maximum(b1[..., :2], b2[..., :2]), 0, None)
    inter = wh[..., 0] * wh[..., 1]
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    return inter / (a1 + a2 - inter + np.float32(eps))


IOUV = np.linspace(0.5, 0.95, 10)  # mAP@0.5:0.95 thresholds (models/yolo/detect/val.py:38)


def match_predictions(pred_classes, true_classes, iou, iouv=IOUV):
    """iou: (L labels, D detections).  Returns the (D, len(iouv)) boolean "correct" matrix: per threshold, candidate
    (label, detection) pairs of matching class with IoU >= thr are taken in descending IoU order, keeping each detection
    and then each label once."""
    pred_classes, true_classes = np.asarray(pred_classes), np.asarray(true_classes)
    correct = np.zeros((pred_classes.shape[0], len(iouv)), bool)
    iou = np.asarray(iou) * (true_classes[:, None] == pred_classes)
    for i, thr in enumerate(np.asarray(iouv).tolist()):
        m = np.array(np.nonzero(iou >= thr)).T
        if m.shape[0]:
            if m.shape[0] > 1:
                m = m[iou[m[:, 0], m[:, 1]].argsort()[::-1]]
                m = m[np.unique(m[:, 1], return_index=True)[1]]
                m = m[np.unique(m[:, 0], return_index=True)[1]]
            correct[m[:, 1].astype(int), i] = True
    return correct


def smooth(y, f=0.05):
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def compute_ap(recall, precision):
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)  # 101-point COCO interpolation
    trapz = getattr(np, "trapezoid", None) or np.trapz
    return trapz(np.interp(x, mrec, mpre), x), mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """-> dict(tp, fp, p, r, f1, ap (nc,10), classes)."""
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    classes, nt = np.unique(target_cls, return_counts=True)
    nc = classes.shape[0]
    x = np.linspace(0, 1, 1000)
    ap, p_curve, r_curve = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_l, n_p = nt[ci], sel.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_l + eps)
        r_curve[ci] = np.interp(-x, -conf[sel], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p_curve[ci] = np.interp(-x, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    k = smooth(f1_curve.mean(0), 0.1).argmax()
    p, r, f1 = p_curve[:, k], r_curve[:, k], f1_curve[:, k]
    tpn = (r * nt).round()
    fpn = (tpn / (p + eps) - tpn).round()
    return dict(tp=tpn, fp=fpn, p=p, r=r, f1=f1, ap=ap, classes=classes.astype(int))


class DetMetrics:
    """Accumulates per-image matches and reports mp, mr, mAP50, mAP50-95 (reference DetMetrics / Metric, metrics.py:627-808)."""

    def __init__(self, iouv=IOUV):
        self.iouv = iouv
        self.tp, self.conf, self.pred_cls, self.target_cls = [], [], [], []

    def update(self, det, labels):
        """det: (n,6) x1,y1,x2,y2,conf,cls (same pixel frame as labels); labels: (m,5) cls,x1,y1,x2,y2."""
        det, labels = np.asarray(det, np.float32).reshape(-1, 6), np.asarray(labels, np.float32).reshape(-1, 5)
        self.target_cls.append(labels[:, 0])
        if det.shape[0] == 0:
            return
        if labels.shape[0]:
            correct = match_predictions(det[:, 5], labels[:, 0], box_iou(labels[:, 1:], det[:, :4]), self.iouv)
        else:
            correct = np.zeros((det.shape[0], len(self.iouv)), bool)
        self.tp.append(correct)
        self.conf.append(det[:, 4])
        self.pred_cls.append(det[:, 5])

    def results(self):
        if not self.tp:
            return dict(mp=0.0, mr=0.0, map50=0.0, map=0.0)
        r = ap_per_class(np.concatenate(self.tp), np.concatenate(self.conf), np.concatenate(self.pred_cls), np.concatenate(self.target_cls))
        ap = r["ap"]
        return dict(mp=float(r["p"].mean()) if len(r["p"]) else 0.0, mr=float(r["r"].mean()) if len(r["r"]) else 0.0,
                    map50=float(ap[:, 0].mean()) if len(ap) else 0.0, map=float(ap.mean()) if len(ap) else 0.0, per_class=r)
