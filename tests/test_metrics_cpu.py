"""CPU: the host-side validation metrics (utils/metrics.py) against goldens produced by the reference's own functions."""
import os

import numpy as np
import torch

import edge_yolo_amd  # noqa: F401
from edge_yolo_amd.utils import metrics as M


def test_metrics_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "metrics_cases.npz"))
    dm = M.DetMetrics()
    for img in range(12):
        lab, det = g[f"lab{img}"], g[f"det{img}"]
        if len(det) and len(lab):
            iou = M.box_iou(lab[:, 1:], det[:, :4])
            np.testing.assert_allclose(iou, g[f"iou{img}"], rtol=1e-6, atol=1e-7)
            np.testing.assert_array_equal(M.match_predictions(det[:, 5], lab[:, 0], iou), g[f"correct{img}"])
        dm.update(det, lab)
    r = dm.results()["per_class"]
    for k in ("tp", "fp", "p", "r", "f1", "ap", "classes"):
        np.testing.assert_allclose(r[k], g["apc_" + k], rtol=1e-9, atol=1e-12, err_msg=k)
    res = dm.results()
    assert abs(res["map"] - g["apc_ap"].mean()) < 1e-12 and 0 < res["map50"] <= 1


def test_validator_matches_reference_update_metrics_and_map(golden_dir):
    """DetectionValidator (label scaling / ratio_pad handling, per-image stats, AP) on the REFERENCE's own post-NMS predictions must
    reproduce the reference's tp matrix, per-class AP and results_dict (tests/golden/validator_case.npz: reference model outputs ->
    reference val-mode NMS -> reference DetectionValidator.update_metrics / get_stats, models/yolo/detect/val.py:104-188)."""
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd.engine.validator import DetectionValidator, KEYS
    from edge_yolo_amd.nn.tasks import DetectionModel
    g = np.load(os.path.join(golden_dir, "validator_case.npz"))
    v = DetectionValidator(DetectionModel("yolo11n-test.yaml"))
    B = len(g["ori_shape"])
    preds = [g[f"pred{i}"] for i in range(B)]
    batch = {"img": torch.zeros(B, 3, 128, 160), "cls": g["cls"], "bboxes": g["bboxes"], "batch_idx": g["batch_idx"],
             "ori_shape": [tuple(s) for s in g["ori_shape"]],
             "ratio_pad": [((float(a), float(a)), (int(p[0]), int(p[1]))) for a, p in zip(g["ratio_gain"], g["ratio_padwh"])]}
    v.update_metrics(preds, batch)
    res = v.get_stats()
    assert v.seen == int(g["seen"])
    np.testing.assert_array_equal(np.concatenate(v.stats["tp"], 0), g["tp"])
    np.testing.assert_array_equal(v.nt_per_class, g["nt_per_class"])
    np.testing.assert_allclose(v.box["ap"], g["ap"], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(v.box["classes"], g["ap_class_index"])
    assert list(g["keys"]) == KEYS + ["fitness"]
    np.testing.assert_allclose([res[k] for k in g["keys"]], g["values"], rtol=0, atol=1e-9)
    # no labels and no predictions: nothing is recorded (val.py:139-145); predictions without labels count as false positives
    v2 = DetectionValidator(DetectionModel("yolo11n-test.yaml"))
    empty = {"img": torch.zeros(2, 3, 64, 64), "cls": np.zeros((0, 1)), "bboxes": np.zeros((0, 4)), "batch_idx": np.zeros(0), "ori_shape": [(64, 64)] * 2, "ratio_pad": None}
    v2.update_metrics([np.zeros((0, 6), np.float32), preds[0][:5]], empty)
    assert v2.seen == 2 and len(v2.stats["tp"]) == 1 and not v2.stats["tp"][0].any() and v2.get_stats()[KEYS[4]] == 0.0
