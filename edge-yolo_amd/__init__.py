"""edge-yolo_amd: MI355X (gfx950)-native detection forward path of EdgeLine-YOLO.

Mirrors the reference's Python surface for that path (`YOLO(...).predict()`, the `nn.tasks` YAML module
registry, `utils.ops.non_max_suppression`) on top of hand-written HIP kernels reached through the C ABI in
`include/edgeyolo_hip.h` (`csrc/libedgeyolo_hip.so`).  There is NO CPU fallback: every operator raises if the
library is missing or the tensor is not on a ROCm device.

The directory is called `edge-yolo_amd`; import it as `edge_yolo_amd` (see `edge_yolo_amd.py` at the repo root).
"""
__version__ = "0.1.0"

from .engine.model import YOLO  # noqa: E402,F401
