"""CPU: the host-side validation metrics (utils/metrics.py) against goldens produced by the reference's own functions."""
import os

import numpy as np

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
