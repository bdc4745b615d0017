"""CPU-only tests of the host side: C-ABI library loads and exports every declared symbol, YAML registry / graph
builder parity with the reference (structure goldens), BN folding and weight packing, view plumbing, loud failure
without a GPU.  No kernel is launched here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import edge_yolo_amd  # noqa: F401
from edge_yolo_amd import _lib as L
from edge_yolo_amd.nn import modules as M
from edge_yolo_amd.nn.tasks import DetectionModel, guess_model_scale, guess_model_task, yaml_model_load
from edge_yolo_amd.utils import ops as uops
from oracle import model as om
import synthdata as synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "edgeyolo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ey_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/edgeyolo_hip.h but not exported"
    assert declared == set(L.SIGNATURES), "ctypes binding table and header disagree"
    assert L.lib().ey_version() >= 1
    # struct layouts must match the header (sizes as the C compiler lays them out)
    assert L.lib().ey_abi_sizeof(0) == ctypes.sizeof(L.ConvDesc) and L.lib().ey_abi_sizeof(1) == ctypes.sizeof(L.ConvDirectDesc)


@pytest.fixture(scope="module")
def structure(golden_dir):
    return json.load(open(os.path.join(golden_dir, "structure.json")))


@pytest.mark.parametrize("name", ["yolo11n.yaml", "yolo11s.yaml", "yolo11l.yaml", "yolo11n-test.yaml", "yolo11s-test.yaml", "yolo11l-test.yaml",
                                  "yolo11n-tune.yaml", "yolo11n-lineattention.yaml", "yolo11n-DSC3K2_Wavelet.yaml", "yolo11n-GF2Detect.yaml"])
def test_registry_builds_the_reference_graph(structure, name):
    g = structure[name]
    m = DetectionModel(name)
    assert sum(p.numel() for p in m.parameters()) == g["params"]
    assert m.save == g["save"] and [float(s) for s in m.stride] == g["stride"]
    assert [l.type for l in m.model] == [l["type"] for l in g["layers"]]
    assert [int(l.np) for l in m.model] == [l["np"] for l in g["layers"]]
    assert [l.f for l in m.model] == [l["f"] for l in g["layers"]]
    assert len(m.state_dict()) == g["nkeys"]
    if "state_shapes" in g:
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == g["state_shapes"]


def test_nc_override_and_scale_guess(structure):
    m = DetectionModel("yolo11n-test.yaml", nc=10)
    assert sum(p.numel() for p in m.parameters()) == structure["yolo11n-test.yaml@nc10"]["params"]
    sd = m.state_dict()
    for k, s in structure["yolo11n-test.yaml@nc10"]["state_shapes"].items():
        assert list(sd[k].shape) == s
    assert guess_model_scale("yolo11x-test.yaml") == "x" and guess_model_scale("foo.yaml") == ""
    assert yaml_model_load("yolo11s-tune.yaml")["scale"] == "s"
    assert guess_model_task("yolo11n-test.yaml") == "detect"  # the reference raises unless task= is passed
    assert [uops.make_divisible(v, 8) for v in (16.0, 17.0, 256.0, 150.0)] == [16, 24, 256, 152]
    with pytest.raises(FileNotFoundError):
        yaml_model_load("yolo99q.yaml")


def test_fuse_matches_reference_fold(golden_dir):
    g = np.load(os.path.join(golden_dir, "ops_small.npz"))
    c = M.Conv(16, 24, 3, 2)
    c.load_state_dict({k: synth.synth_tensor("conv_s2." + k, tuple(v.shape)) for k, v in c.state_dict().items()})
    c.bn.eps = 1e-3
    w, b = c.folded()
    np.testing.assert_allclose(w.numpy(), g["conv_s2_fused_w"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b.numpy(), g["conv_s2_fused_b"], rtol=1e-5, atol=1e-6)
    c.fuse_bn()
    assert not hasattr(c, "bn") and set(c.state_dict()) == {"conv.weight", "conv.bias"}
    np.testing.assert_allclose(c.conv.weight.detach().numpy(), g["conv_s2_fused_w"], rtol=1e-5, atol=1e-6)
    m = DetectionModel("yolo11n-test.yaml")
    n_bn = sum(isinstance(x, torch.nn.BatchNorm2d) for x in m.modules())
    m.fuse()
    left = sum(isinstance(x, torch.nn.BatchNorm2d) for x in m.modules())
    assert left == 22 and n_bn > left  # DSConv BatchNorms survive fuse(), as in the reference


@pytest.mark.parametrize("cout,cin,k", [(8, 16, 3), (64, 48, 1), (80, 64, 1), (256, 128, 3), (12, 24, 1)])
def test_weight_packing_layout(cout, cin, k):
    """packed[row][ (ky*k+kx)*Cin + c ] with the documented row permutation; pad rows/slack are zero."""
    w = torch.randn(cout, cin, k, k)
    nbytes = L.lib().ey_conv_packed_bytes(L.F32, cout, cin, k)
    buf = torch.zeros(nbytes, dtype=torch.uint8)
    assert L.lib().ey_conv_pack_weight(L.F32, cout, cin, k, w.data_ptr(), buf.data_ptr(), nbytes) == 0
    kp = k * k * cin + 32
    kp += 0 if (kp >> 3) & 1 else 8  # f32 rows: an odd number of 8-element units = 2 (mod 4) 16-byte units (conflict-free LDS pitch, ey_conv_kpad)
    p = buf.view(torch.float32).view(-1, kp)
    nt = L.lib().ey_conv_pack_nt(cout)
    bn = 16 * nt
    seen = set()
    for row in range(p.shape[0]):
        nb, within = divmod(row, bn)
        t, rho = divmod(within, 16)
        gq, j = divmod(rho, 4)
        ch = nb * bn + gq * 4 * nt + 4 * t + j
        if ch < cout:
            seen.add(ch)
            want = w[ch].permute(1, 2, 0).reshape(-1)  # (ky,kx,c)
            assert torch.equal(p[row, : k * k * cin], want)
        else:
            assert float(p[row].abs().sum()) == 0
        assert float(p[row, k * k * cin:].abs().sum()) == 0
    assert seen == set(range(cout))
    assert L.lib().ey_conv_pack_weight(L.F32, cout, cin, 5, w.data_ptr(), buf.data_ptr(), nbytes) != 0
    assert b"k=5" in L.lib().ey_last_error()


def test_nhwc_view_plumbing():
    x = L.empty_nhwc(2, 48, 5, 7, torch.float16, "cpu")
    assert L.is_nhwc_view(x) and L.cstride(x) == 48 and x.is_contiguous(memory_format=torch.channels_last)
    s = x[:, 16:32]
    assert L.is_nhwc_view(s) and L.cstride(s) == 48 and s.data_ptr() == x.data_ptr() + 32
    assert not L.is_nhwc_view(torch.zeros(2, 48, 5, 7))
    assert L.is_nhwc_view(torch.zeros(2, 48, 5, 7).contiguous(memory_format=torch.channels_last))
    assert L.dtype_code(torch.float16) == 0 and L.dtype_code(torch.float32) == 1
    with pytest.raises(TypeError):
        L.dtype_code(torch.bfloat16)


def test_no_cpu_fallback_anywhere():
    """The product path must fail loudly, not fall back, when there is no GPU / the tensor is on the CPU."""
    m = DetectionModel("yolo11n-test.yaml")
    with pytest.raises(L.HipLibraryError):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(L.HipLibraryError):
        uops.non_max_suppression(torch.zeros(1, 84, 100))
    model = edge_yolo_amd.YOLO("yolo11n-test.yaml")
    with pytest.raises(RuntimeError):
        model.predict(torch.zeros(1, 3, 64, 64), device="cpu")
    with pytest.raises(NotImplementedError):
        m.train()
    with pytest.raises(NotImplementedError):
        uops.non_max_suppression(torch.zeros(1, 84, 100), rotated=True)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "edge-yolo_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"


def test_save_load_roundtrip_unfused_and_fused(tmp_path):
    """YOLO.save -> YOLO(path): tensor-only checkpoint, nc and the BN-folded ('fused') layout survive (predict() folds BatchNorm in
    place, so a model saved after predict must come back with the folded conv weights AND biases, not with default BatchNorms)."""
    y = edge_yolo_amd.YOLO("yolo11n-test.yaml", nc=10)
    y.model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in y.model.state_dict().items()}))
    f1 = str(tmp_path / "unfused.pt")
    y.save(f1)
    ck = torch.load(f1, weights_only=True)  # the file is readable by the safe loader
    assert ck["fused"] is False and ck["nc"] == 10
    y1 = edge_yolo_amd.YOLO(f1)
    a, b = y.model.state_dict(), y1.model.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a) and y1.model.model[-1].nc == 10
    y.model.fuse()  # what predict() does
    f2 = str(tmp_path / "fused.pt")
    y.save(f2)
    assert torch.load(f2, weights_only=True)["fused"] is True
    y2 = edge_yolo_amd.YOLO(f2)
    a, b = y.model.state_dict(), y2.model.state_dict()
    assert y2.model.convs_folded() and set(a) == set(b) and all(torch.equal(a[k].float(), b[k].float()) for k in a)
    # a checkpoint whose tensors do not cover the model is an error, not a silently half-initialised model
    ck = torch.load(f2, weights_only=True)
    ck["fused"] = False
    torch.save(ck, f2)
    with pytest.raises(ValueError, match="model tensors"):
        edge_yolo_amd.YOLO(f2)


def test_head_caches_follow_load_state_dict():
    """Detect keeps packed tails / quality-head weights in plain dict caches: a load_state_dict must drop them."""
    m = DetectionModel("yolo11n-test.yaml")
    h = m.model[-1]
    h._tails[123] = object()
    h._qcache[(0, "cpu")] = object()
    h._stride_f = [1.0]
    m.load_state_dict(m.state_dict())
    assert h._tails == {} and h._qcache == {} and h._stride_f is None


def test_tunables_are_api_not_environment():
    lib = L.lib()
    assert lib.ey_tune_get(b"pw_m") == 110000
    assert lib.ey_tune_set(b"pw_m", 5) == 0 and lib.ey_tune_get(b"pw_m") == 5
    assert lib.ey_tune_set(b"pw_m", 110000) == 0
    assert lib.ey_tune_set(b"no_such_knob", 1) != 0 and lib.ey_tune_get(b"no_such_knob") == -1
    for d, _, files in os.walk(os.path.join(ROOT, "edge-yolo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(d, f)).read()
                assert "getenv" not in src and "os.environ" not in src, f"{f} reads the environment"


@pytest.mark.skipif(not os.path.isdir("/root/reference/ultralytics"), reason="needs the reference package (build container only)")
def test_reference_checkpoint_bridge(tmp_path):
    """§8f-3: a checkpoint as the REFERENCE trainer writes it (pickled module, engine/trainer.py:513-546) -> tools/export_reference_weights.py
    (runs where the reference is importable) -> tensor-only file -> YOLO(path): same YAML, nc, every tensor equal."""
    import subprocess
    import sys
    ref_pt, out_pt = str(tmp_path / "ref_best.pt"), str(tmp_path / "bridge.pt")
    code = f"""
import sys
sys.path.insert(0, {os.path.join(ROOT, 'tests', 'golden')!r}); sys.path.insert(0, {ROOT!r})
import _ref_import; _ref_import.setup()
import torch, synthdata as synth
from ultralytics.nn.tasks import DetectionModel
m = DetectionModel('yolo11n-test.yaml', ch=3, nc=10, verbose=False)
m.load_state_dict(synth.synth_state_dict({{k: tuple(v.shape) for k, v in m.state_dict().items()}}))
torch.save({{'epoch': 3, 'model': None, 'ema': m.half(), 'train_args': {{}}}}, {ref_pt!r})   # the trainer's checkpoint layout
sys.path.insert(0, {os.path.join(ROOT, 'tools')!r})
import export_reference_weights as ex
print(ex.export({ref_pt!r}, {out_pt!r}))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ck = torch.load(out_pt, weights_only=True)  # tensor-only: the safe loader reads it
    assert ck["nc"] == 10 and ck["fused"] is False and isinstance(ck["yaml"], dict)
    y = edge_yolo_amd.YOLO(out_pt)
    assert y.model.model[-1].nc == 10
    want = synth.synth_state_dict({k: tuple(v.shape) for k, v in y.model.state_dict().items()})
    got = y.model.state_dict()
    assert set(got) == set(ck["state_dict"])
    for k, v in want.items():
        if v.is_floating_point():  # the checkpoint was the trainer's half-precision EMA: values are the f16-rounded synthetic weights
            torch.testing.assert_close(got[k].float(), v.half().float(), rtol=0, atol=0)
        else:
            assert torch.equal(got[k], v)


def test_e2e_detect_structure_matches_reference(golden_dir):
    """E2EDetect (reference head.py:799-824) builds the reference's state_dict: one2one copies of both towers and the quality head."""
    import json
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd.nn.tasks import DetectionModel, yaml_model_load
    g = np.load(os.path.join(golden_dir, "e2e_128x160.npz"))
    d = yaml_model_load("yolo11n-test.yaml")
    d["head"][-1][2] = "E2EDetect"
    m = DetectionModel(d)
    sd = m.state_dict()
    assert m.end2end and len(sd) == int(g["nkeys"]) and sum(p.numel() for p in m.parameters()) == int(g["params"])
    for k, v in json.loads(str(g["state_shapes"])).items():
        assert tuple(sd[k].shape) == tuple(v), k
    assert [float(s) for s in m.stride] == [float(s) for s in g["stride"]]


def test_custom_architecture_survives_save_reload_save(tmp_path):
    """a model built from a YAML *dict* (custom architecture whose 'yaml_file' names no shipped file) must round-trip through
    save() twice: the checkpoint carries the dict itself, not a path that may not exist where it is loaded."""
    from edge_yolo_amd.nn.tasks import yaml_model_load
    d = dict(yaml_model_load("yolo11n-test.yaml"))
    d["yaml_file"] = "/somewhere/else/my-custom-n.yaml"  # a path on the training machine (tools/export_reference_weights.py writes such dicts)
    d["backbone"] = [list(r) for r in d["backbone"]]
    d["backbone"][2] = [-1, 2, "DSC3K2_Wavelet", [256, False, 0.5]]  # differs from every shipped YAML (e = 0.5 instead of 0.25)
    y = edge_yolo_amd.YOLO(d, nc=7)
    y.model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in y.model.state_dict().items()}))
    f1, f2 = str(tmp_path / "a.pt"), str(tmp_path / "b.pt")
    y.save(f1)
    ck = torch.load(f1, weights_only=True)
    assert isinstance(ck["yaml"], dict) and ck["yaml"]["backbone"][2][3] == [256, False, 0.5]
    y1 = edge_yolo_amd.YOLO(f1)
    y1.save(f2)
    y2 = edge_yolo_amd.YOLO(f2)
    a, b = y.model.state_dict(), y2.model.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert y2.model.model[-1].nc == 7 and [m.type for m in y2.model.model] == [m.type for m in y.model.model]


def test_bias_init_covers_the_one2one_branch():
    """reference head.py:150-161: box biases 1.0, class biases log(5/nc/(640/s)^2), for cv2/cv3 AND (end2end heads) one2one_cv2/one2one_cv3."""
    import math
    from edge_yolo_amd.nn.tasks import yaml_model_load
    d = dict(yaml_model_load("yolo11n-test.yaml"))
    d["head"] = [list(r) for r in d["head"]]
    d["head"][-1][2] = "E2EDetect"
    m = DetectionModel(d)
    h = m.model[-1]
    h.bias_init()
    for towers in ((h.cv2, h.cv3), (h.one2one_cv2, h.one2one_cv3)):
        for a, b, s in zip(*towers, h.stride):
            assert torch.all(a[-1].bias == 1.0)
            assert torch.allclose(b[-1].bias[: h.nc], torch.tensor(math.log(5 / h.nc / (640 / float(s)) ** 2)))


def test_bench_sparse_regime_biases_are_bias_init():
    """bench.py --regime sparse writes exactly Detect.bias_init's values into the synthetic state_dict."""
    import bench
    m = DetectionModel("yolo11n-test.yaml")
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    bench.sparse_biases(sd, len(m.model) - 1, m.model[-1].nc, [float(s) for s in m.stride])
    m.model[-1].bias_init()
    ref = m.state_dict()
    for k in sd:
        assert torch.equal(sd[k], ref[k]), k


def test_host_tensor_range_check_follows_the_reference():
    """LoadTensor._single_check (data/loaders.py:560-566): a HOST tensor whose max exceeds 1 is divided by 255 with a warning (checked on
    the host: no device sync); integer tensors are a TypeError; validate_input=False switches the check off."""
    from edge_yolo_amd.engine.predictor import DetectionPredictor

    class _P(DetectionPredictor):  # preprocess up to the device upload only
        def __init__(self, validate_input=None):
            self.validate_input, self.half, self.device = validate_input, False, torch.device("cpu")
    x = torch.rand(1, 3, 32, 32) * 255
    with pytest.warns(UserWarning, match="Dividing input by 255"):
        y = _P().preprocess(x)
    assert float(y.max()) <= 1.0 and torch.allclose(y, x / 255.0)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert torch.equal(_P().preprocess(x / 255.0), x / 255.0)  # in range: untouched, no warning
        assert torch.equal(_P(validate_input=False).preprocess(x), x)
    with pytest.raises(TypeError):
        _P().preprocess(torch.zeros(1, 3, 32, 32, dtype=torch.uint8))
