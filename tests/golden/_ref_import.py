"""Import recipe for the reference package. Used ONLY by tests/golden/make_golden.py,
which runs in the build container where /root/reference exists. Nothing on the GPU box imports this.

Missing third-party modules are replaced by inert stand-ins that the detection forward path never
computes with (cv2, thop), or by the two published Haar filter constants (pywt); torchvision.ops.nms is
NOT available, so NMS goldens use a documented stand-in (see make_golden.py) and that boundary stays
"parity unpinned".
"""
import os, sys, math, types, importlib.metadata as md
from unittest.mock import MagicMock

REF = "/root/reference"


def setup():
    os.environ["YOLO_OFFLINE"] = "true"
    os.environ["YOLO_CONFIG_DIR"] = "/tmp/ey_cfg"
    os.makedirs("/tmp/ey_cfg", exist_ok=True)
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cv2 = MagicMock(name="cv2"); cv2.__version__ = "4.9.0"; sys.modules["cv2"] = cv2
    sys.modules["thop"] = MagicMock(name="thop")
    pywt = types.ModuleType("pywt"); s = 1 / math.sqrt(2)

    class Wavelet:
        def __init__(self, name):
            assert name in ("haar", "db1"), name
            self.dec_lo, self.dec_hi, self.rec_lo, self.rec_hi = [s, s], [-s, s], [s, s], [s, -s]
    pywt.Wavelet = Wavelet; sys.modules["pywt"] = pywt
    tv = types.ModuleType("torchvision"); tv.__version__ = "0.17.2"
    tv.ops = types.ModuleType("torchvision.ops")
    sys.modules["torchvision"] = tv; sys.modules["torchvision.ops"] = tv.ops
    _v = md.version
    md.version = lambda n: "0.17.2" if n == "torchvision" else _v(n)
    return tv
