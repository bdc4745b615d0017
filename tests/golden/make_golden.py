"""Golden-vector generator: runs the REAL reference (/root/reference, imported with the SURVEY.md
Appendix C recipe) on synthetic weights/inputs from synthdata.py and stores inputs-by-seed +
expected outputs as small fixtures next to this file.  Runs only in the build container; the GPU box
never has /root/reference and never runs this.  Usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402

tv = _ref_import.setup()

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synthdata as synth  # noqa: E402
from oracle.nms import tv_nms  # noqa: E402

# stand-in for the absent third-party torchvision.ops.nms (PARITY UNPINNED there, see oracle/nms.py)
tv.ops.nms = lambda b, s, t: torch.as_tensor(tv_nms(b.numpy(), s.numpy(), t), dtype=torch.long)

from ultralytics.nn.tasks import DetectionModel  # noqa: E402
from ultralytics.nn.modules import block as rb, conv as rc  # noqa: E402
from ultralytics.utils import ops as rops, tal as rtal  # noqa: E402
from ultralytics.utils.torch_utils import fuse_conv_and_bn  # noqa: E402

torch.set_grad_enabled(False)
LARGE_GAIN = 1.0  # l / x scales are twice as deep: the default conv gain (1.9) lets activations grow to ~1e4 there; 1.0 keeps them O(10)
YAMLS = ["yolo11", "yolo11-test", "yolo11-tune", "yolo11-lineattention", "yolo11-DSC3K2_Wavelet", "yolo11-GF2Detect"]


def build(name, nc=80, gain=1.9):
    m = DetectionModel(name, ch=3, nc=nc, verbose=False).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(synth.synth_state_dict(shapes, gain=gain))
    return m, shapes


def structure():
    out = {}
    for y in YAMLS:
        for sc in (["n", "s", "l"] if y in ("yolo11", "yolo11-test") else ["n"]):
            name = y.replace("yolo11", "yolo11" + sc) + ".yaml"
            m = DetectionModel(name, ch=3, nc=80, verbose=False)
            e = dict(params=sum(p.numel() for p in m.parameters()), save=list(m.save),
                     stride=[float(s) for s in m.stride],
                     layers=[dict(i=l.i, f=l.f, type=l.type, np=int(l.np)) for l in m.model],
                     nkeys=len(m.state_dict()), nelem=sum(v.numel() for v in m.state_dict().values()))
            if sc == "n":
                e["state_shapes"] = {k: list(v.shape) for k, v in m.state_dict().items()}
            out[name] = e
            print(name, e["params"])
    # nc=10 (GC10-DET) changes the cls tower width c3 (head.py:59)
    m = DetectionModel("yolo11n-test.yaml", ch=3, nc=10, verbose=False)
    out["yolo11n-test.yaml@nc10"] = dict(params=sum(p.numel() for p in m.parameters()),
                                         state_shapes={k: list(v.shape) for k, v in m.state_dict().items() if k.startswith("model.23.cv3.0")})
    json.dump(out, open(os.path.join(HERE, "structure.json"), "w"), indent=0)


HOOKS = {  # inner modules whose outputs pin the per-op restatements
    "yolo11n-test.yaml": ["model.2.cv1", "model.2.wave.f_ll", "model.2.wave", "model.2.m.0.cv1", "model.2.m.0", "model.6.m.0",
                          "model.10.m.0.attn", "model.10.m.0", "model.23.cv2.0", "model.23.cv3.0.0.0", "model.23.cv3.0", "model.23.reg_conf.0"],
    "yolo11n.yaml": ["model.2.m.0", "model.6.m.0", "model.10.m.0.attn", "model.10.m.0"],
}


def model_small(name, tag, b=2, h=64, w=64, layers=True, gain=1.9):
    m, _ = build(name, gain=gain)
    m.fuse(verbose=False)
    d = {}
    mods = dict(m.named_modules())
    hs = []
    for i, l in enumerate(m.model if layers else []):
        hs.append(l.register_forward_hook(lambda mod, inp, out, i=i: d.__setitem__(f"layer{i}", out.clone()) if torch.is_tensor(out) else None))
    for k in HOOKS.get(name, []) if layers else []:
        hs.append(mods[k].register_forward_hook(lambda mod, inp, out, k=k: d.__setitem__(k, out.clone())))
    x = synth.synth_images(b, h, w)
    y, raw = m(x)
    d["y"] = y
    for i, r in enumerate(raw):
        d[f"raw{i}"] = r
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **{k: v.numpy() for k, v in d.items()})
    print(tag, len(d), y.shape)


def model_640(name, tag, h=640, w=640, step=7):
    m, _ = build(name)
    m.fuse(verbose=False)
    y, raw = m(synth.synth_images(1, h, w))
    d = dict(y_sub=y[:, :, ::step].numpy(), step=np.int64(step), row_sum=y.double().sum(-1).numpy(),
             row_abs=y.double().abs().sum(-1).numpy())
    # the full predict-time post-process on the same output (conf .25, iou .7: cfg/default.yaml:51-65)
    det = rops.non_max_suppression(y.clone(), 0.25, 0.7, max_det=300, max_time_img=1e6)
    d["det0"] = det[0].numpy()
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **d)
    print(tag, y.shape, det[0].shape)


def ops_small():
    d = {}
    # Haar DWT known answer + random (block.py:3582-3642)
    dwt = rb._PywtDWT2D("haar")
    for k, t in zip(("LL", "LH", "HL", "HH"), dwt(torch.tensor([[[[1., 2.], [3., 4.]]]]))):
        d["dwt_quad_" + k] = t
    x = synth.synth_images(2, 6, 10, c=4) * 2 - 1
    for k, t in zip(("LL", "LH", "HL", "HH"), dwt(x)):
        d["dwt_rand_" + k] = t
    xo = synth.synth_images(1, 5, 7, c=2)
    for k, t in zip(("LL", "LH", "HL", "HH"), dwt(xo)):
        d["dwt_odd_" + k] = t

    def filled(mod, prefix):
        mod.eval()
        for mm in mod.modules():
            if isinstance(mm, torch.nn.BatchNorm2d):
                mm.eps = 1e-3  # initialize_weights, torch_utils.py:416
        mod.load_state_dict(synth.synth_state_dict({prefix + "." + k: tuple(v.shape) for k, v in mod.state_dict().items()}) and
                            {k: synth.synth_tensor(prefix + "." + k, tuple(v.shape)) for k, v in mod.state_dict().items()})
        return mod

    # enhancer on even and odd maps (block.py:3645-3710)
    enh = filled(rb._WaveletEnhancer(16), "enh")
    d["enh_even"] = enh(synth.synth_images(2, 10, 14, c=16) - 0.5)
    d["enh_odd"] = enh(synth.synth_images(1, 9, 13, c=16) - 0.5)
    for k in (3, 5, 7):  # DSConv (conv.py:87-104)
        ds = filled(rc.DSConv(16, 24, k), f"ds{k}")
        d[f"dsconv{k}"] = ds(synth.synth_images(2, 9, 11, c=16) - 0.5)
    sp = filled(rb.SPPF(32, 48, 5), "sppf")
    d["sppf"] = sp(synth.synth_images(1, 7, 9, c=32) - 0.5)
    la = filled(rb.PSABlock_LinearAttention(128, num_heads=2), "psa_la")
    d["psa_la"] = la(synth.synth_images(2, 4, 4, c=128) - 0.5)
    at = filled(rb.PSABlock(128, num_heads=2), "psa")
    d["psa"] = at(synth.synth_images(2, 4, 4, c=128) - 0.5)
    cv = filled(rc.Conv(16, 24, 3, 2), "conv_s2")
    d["conv_s2_unfused"] = cv(synth.synth_images(2, 9, 11, c=16) - 0.5)
    fc = fuse_conv_and_bn(cv.conv, cv.bn)
    d["conv_s2_fused_w"], d["conv_s2_fused_b"] = fc.weight, fc.bias
    # anchors / dist2bbox (tal.py:333-357)
    feats = [torch.zeros(1, 8, 4, 6), torch.zeros(1, 8, 2, 3), torch.zeros(1, 8, 1, 2)]
    a, s = rtal.make_anchors(feats, torch.tensor([8., 16., 32.]), 0.5)
    d["anchors"], d["anchor_strides"] = a, s
    dist = synth.synth_images(1, 4, a.shape[0], c=1)[0] * 5
    d["dist2bbox"] = rtal.dist2bbox(dist, a.t().unsqueeze(0), xywh=True, dim=1)
    d["make_divisible"] = torch.tensor([rops.make_divisible(v, 8) for v in (16.0, 17.0, 64 * 0.25, 1024 * 0.25, 100 * 1.5)])
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **{k: v.numpy() for k, v in d.items()})
    print("ops_small", len(d))


def nms_cases():
    d, meta = {}, {}

    def run(tag, pred, **kw):
        # max_time_img: the reference aborts NMS on a wall-clock limit (ops.py:238,312-314); with the slow numpy stand-in for
        # torchvision.ops.nms that limit could truncate a multi-image case on a slow box -> neutralised (not part of the stored kwargs)
        out = rops.non_max_suppression(pred.clone(), max_time_img=1e6, **kw)
        meta[tag] = dict(kw=kw, n=[int(o.shape[0]) for o in out])
        for i, o in enumerate(out):
            d[f"{tag}_{i}"] = o.numpy()

    sparse = synth.synth_pred(2, 80, 8400, seed=2)
    run("sparse", sparse, conf_thres=0.25, iou_thres=0.7)
    run("sparse_iou45_det20", sparse, conf_thres=0.25, iou_thres=0.45, max_det=20)
    run("sparse_agnostic", sparse, conf_thres=0.25, iou_thres=0.7, agnostic=True)
    run("sparse_classes", sparse, conf_thres=0.25, iou_thres=0.7, classes=[0, 3, 17])
    dense = synth.synth_pred(1, 80, 8400, seed=3, dense=True)
    run("dense", dense, conf_thres=0.25, iou_thres=0.7)
    run("val_multilabel", synth.synth_pred(1, 80, 2100, seed=4, dense=True), conf_thres=0.001, iou_thres=0.7, multi_label=True, max_det=300)
    small = synth.synth_pred(1, 10, 336, seed=5, imgsz=128, dense=True)
    run("nc10", small, conf_thres=0.25, iou_thres=0.7)
    run("none_pass", synth.synth_pred(2, 80, 336, seed=6) * torch.tensor(1e-3), conf_thres=0.25, iou_thres=0.7)
    # hand-made: score ties, identical boxes, class-offset separation, touching boxes, zero-area box
    hand = torch.zeros(1, 4 + 3, 12)
    bx = [(50, 50, 40, 40), (52, 50, 40, 40), (50, 52, 40, 40), (50, 50, 40, 40), (150, 50, 40, 40), (190, 50, 40, 40),
          (50, 50, 40, 40), (300, 300, 0, 0), (300, 300, 10, 10), (302, 300, 10, 10), (52, 52, 40, 40), (48, 50, 40, 40)]
    sc = [0.9, 0.9, 0.9, 0.8, 0.7, 0.7, 0.6, 0.5, 0.5, 0.5, 0.26, 0.2]
    cl = [0, 0, 1, 0, 2, 2, 1, 0, 0, 0, 0, 0]
    for i, (b, s, c) in enumerate(zip(bx, sc, cl)):
        hand[0, :4, i] = torch.tensor(b, dtype=torch.float32)
        hand[0, 4 + c, i] = s
    d["hand_pred"] = hand.numpy()
    run("hand", hand, conf_thres=0.25, iou_thres=0.45)
    run("hand_agn", hand, conf_thres=0.25, iou_thres=0.45, agnostic=True)
    np.savez_compressed(os.path.join(HERE, "nms_cases.npz"), **d)
    json.dump(meta, open(os.path.join(HERE, "nms_cases.json"), "w"), indent=0)
    print("nms", meta)


def metrics_cases():
    """reference box_iou / match_predictions / ap_per_class on seeded synthetic detections and labels."""
    from ultralytics.utils import metrics as rm
    from ultralytics.engine.validator import BaseValidator
    r = np.random.default_rng(7)
    d = {}
    tps, confs, pcls, tcls = [], [], [], []
    iouv = torch.linspace(0.5, 0.95, 10)
    holder = type("V", (), {"iouv": iouv})()
    for img in range(12):
        m = int(r.integers(0, 6))
        lab = np.zeros((m, 5), np.float32)
        lab[:, 0] = r.integers(0, 4, m)
        xy = r.uniform(0, 500, (m, 2)); wh = r.uniform(20, 120, (m, 2))
        lab[:, 1:3] = xy; lab[:, 3:5] = xy + wh
        n = int(r.integers(0, 15))
        det = np.zeros((n, 6), np.float32)
        for k in range(n):
            if m and r.random() < 0.6:  # jittered copy of a label
                j = int(r.integers(0, m))
                det[k, :4] = lab[j, 1:] + r.normal(0, 6, 4)
                det[k, 5] = lab[j, 0] if r.random() < 0.85 else r.integers(0, 4)
            else:
                p0 = r.uniform(0, 500, 2); det[k, :2] = p0; det[k, 2:4] = p0 + r.uniform(20, 120, 2); det[k, 5] = r.integers(0, 4)
            det[k, 4] = r.uniform(0.01, 1.0)
        d[f"lab{img}"], d[f"det{img}"] = lab, det
        if n:
            if m:
                iou = rm.box_iou(torch.tensor(lab[:, 1:]), torch.tensor(det[:, :4]))
                d[f"iou{img}"] = iou.numpy()
                c = BaseValidator.match_predictions(holder, torch.tensor(det[:, 5]), torch.tensor(lab[:, 0]), iou).numpy()
            else:
                c = np.zeros((n, 10), bool)
            d[f"correct{img}"] = c
            tps.append(c); confs.append(det[:, 4]); pcls.append(det[:, 5])
        tcls.append(lab[:, 0])
    out = rm.ap_per_class(np.concatenate(tps), np.concatenate(confs), np.concatenate(pcls), np.concatenate(tcls))
    for k, v in zip(("tp", "fp", "p", "r", "f1", "ap", "classes"), out[:7]):
        d["apc_" + k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "metrics_cases.npz"), **d)
    print("metrics", out[5].mean())


def validator_case():
    """End-to-end validation fixture from REAL model outputs: reference EdgeLine-n forward on 6 letterboxed synthetic images ->
    reference validation-mode NMS -> labels = jittered copies of confident detections (in the dataset's collate format, with
    ori_shape / ratio_pad) -> the reference's own DetectionValidator.update_metrics / get_stats / DetMetrics."""
    import types
    from ultralytics.models.yolo.detect.val import DetectionValidator as RV
    from ultralytics.utils import metrics as rm
    m, _ = build("yolo11n-test.yaml")
    m.fuse(verbose=False)
    B, H, W = 6, 128, 160
    x = synth.synth_images(B, H, W, seed=9)
    y, _ = m(x)
    preds = rops.non_max_suppression(y.clone(), 0.001, 0.7, multi_label=True, max_det=300, max_time_img=1e6)
    r = np.random.default_rng(5)
    ori = [(100, 160), (128, 128), (256, 320), (128, 160), (90, 120), (64, 80)]
    ratio_pad = []
    for h0, w0 in ori:  # LetterBox geometry (data/augment.py:1556-1591) as the dataset records it
        g = min(H / h0, W / w0)
        nw, nh = round(w0 * g), round(h0 * g)
        dw, dh = (W - nw) / 2, (H - nh) / 2
        ratio_pad.append(((g, g), (int(round(dw - 0.1)), int(round(dh - 0.1)))))
    cls, box, bidx = [], [], []
    for i, p in enumerate(preds):
        k = [3, 0, 5, 2, 4, 1][i]
        for j in r.permutation(min(len(p), 40))[:k]:
            b = p[j, :4].numpy() + r.normal(0, 2.0, 4)
            c = float(p[j, 5]) if r.random() < 0.8 else float(r.integers(0, 80))
            cls.append([c]); bidx.append(i)
            box.append([(b[0] + b[2]) / 2 / W, (b[1] + b[3]) / 2 / H, abs(b[2] - b[0]) / W, abs(b[3] - b[1]) / H])
    batch = dict(img=x, cls=torch.tensor(cls, dtype=torch.float32), bboxes=torch.tensor(np.array(box), dtype=torch.float32),
                 batch_idx=torch.tensor(bidx, dtype=torch.float32), ori_shape=ori, ratio_pad=ratio_pad, im_file=[f"im{i}.jpg" for i in range(B)])

    class V(RV):
        def __init__(self):  # the validator's own set-up needs a dataset / cfg; only the metric plumbing is exercised
            pass
    v = V()
    v.device = torch.device("cpu")
    v.args = types.SimpleNamespace(single_cls=False, plots=False, save_json=False, save_txt=False, save_conf=False)
    v.iouv = torch.linspace(0.5, 0.95, 10)
    v.niou, v.nc, v.seen = 10, 80, 0
    v.names = {i: str(i) for i in range(80)}
    v.metrics = rm.DetMetrics(names=v.names)
    v.stats = dict(tp=[], conf=[], pred_cls=[], target_cls=[], target_img=[])
    v.confusion_matrix = None
    v.update_metrics([p.clone() for p in preds], batch)
    tp = torch.cat(v.stats["tp"], 0).numpy()
    res = v.get_stats()
    d = dict(cls=batch["cls"].numpy(), bboxes=batch["bboxes"].numpy(), batch_idx=batch["batch_idx"].numpy(), ori_shape=np.array(ori),
             ratio_gain=np.array([rp[0][0] for rp in ratio_pad]), ratio_padwh=np.array([rp[1] for rp in ratio_pad]), tp=tp,
             keys=np.array(list(res.keys())), values=np.array([float(v_) for v_ in res.values()]), seen=np.int64(v.seen),
             nt_per_class=v.nt_per_class, ap=v.metrics.box.all_ap, ap_class_index=np.asarray(v.metrics.box.ap_class_index))
    for i, p in enumerate(preds):
        d[f"pred{i}"] = p.numpy()
    np.savez_compressed(os.path.join(HERE, "validator_case.npz"), **d)
    print("validator", dict(res), "tp", int(tp.sum()), "labels", len(cls))


def e2e_case():
    """E2EDetect (reference head.py:273-298,799-824; no YAML of the reference names it): the EdgeLine-n graph with its head entry renamed,
    built by the reference's own parse_model.  Inference output = Detect.postprocess top-k rows (B, 300, 6) [x1,y1,x2,y2,score,cls]
    of the one2one branch, plus the raw maps of both branches; the predict-time filter of non_max_suppression (ops.py:224-228)."""
    import yaml
    from ultralytics.nn.tasks import yaml_model_load
    d = yaml_model_load("yolo11n-test.yaml")
    assert d["head"][-1][2] in ("GFLHeadv2_uniH", "GF2Detect", "Detect"), d["head"][-1]
    d["head"][-1][2] = "E2EDetect"
    m = DetectionModel(d, ch=3, nc=80, verbose=False).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(synth.synth_state_dict(shapes, gain=1.9))
    nparams = sum(p.numel() for p in m.parameters())
    m.fuse(verbose=False)
    x = synth.synth_images(2, 128, 160, seed=3)
    y, aux = m(x)
    out = dict(y=y.numpy(), stride=np.array([float(s) for s in m.stride]), nkeys=np.int64(len(shapes)), params=np.int64(nparams))
    for br in ("one2one", "one2many"):
        for i, r in enumerate(aux[br]):
            out[f"{br}{i}"] = r.numpy()
    det = rops.non_max_suppression(y.clone(), 0.25, 0.7, max_det=300, classes=None)
    detc = rops.non_max_suppression(y.clone(), 0.05, 0.7, max_det=20, classes=[int(y[0, 0, 5]), int(y[1, 3, 5])])
    for i in range(2):
        out[f"det{i}"] = det[i].numpy()
        out[f"detc{i}"] = detc[i].numpy()
    out["detc_classes"] = np.array([int(y[0, 0, 5]), int(y[1, 3, 5])])
    out["state_shapes"] = np.array(json.dumps({k: list(v) for k, v in shapes.items() if "one2one" in k or k.startswith("model.23.cv3.0")}))
    np.savez_compressed(os.path.join(HERE, "e2e_128x160.npz"), **out)
    print("e2e", y.shape, [tuple(r.shape) for r in aux["one2one"]], [len(t) for t in det], [len(t) for t in detc], float(y[0, 0, 4]), float(y[0, -1, 4]))


def augment_case():
    """DetectionModel._predict_augment (tasks.py:372-408): scales 1 / 0.83 / 0.67, lr-flip of the middle one, _descale_pred, _clip_augmented,
    concat over anchors -- `m(x, augment=True)` of the reference on EdgeLine-n at 64x64 and 96x160, plus the predict-time NMS on it."""
    m, _ = build("yolo11n-test.yaml")
    m.fuse(verbose=False)
    out = {}
    for tag, (b, h, w) in (("64", (2, 64, 64)), ("96x160", (1, 96, 160))):
        x = synth.synth_images(b, h, w, seed=4)
        y, none = m(x, augment=True)
        assert none is None
        out[f"y_{tag}"] = y.numpy()
        det = rops.non_max_suppression(y.clone(), 0.25, 0.7, max_det=300, max_time_img=1e6)
        for i, t in enumerate(det):
            out[f"det_{tag}_{i}"] = t.numpy()
        print("augment", tag, tuple(y.shape), [len(t) for t in det])
    np.savez_compressed(os.path.join(HERE, "augment_n.npz"), **out)


if __name__ == "__main__":
    if "--augment-only" in sys.argv:
        augment_case()
        sys.exit(0)
    if "--e2e-only" in sys.argv:
        e2e_case()
        sys.exit(0)
    if "--validator-only" in sys.argv:
        validator_case()
        sys.exit(0)
    if "--large-only" in sys.argv:
        model_small("yolo11l-test.yaml", "edgeline_l_64", b=1, layers=False, gain=LARGE_GAIN)
        model_small("yolo11x.yaml", "yolo11x_64", b=1, layers=False, gain=LARGE_GAIN)
        sys.exit(0)
    metrics_cases() if "--metrics-only" in sys.argv else None
    if "--metrics-only" in sys.argv:
        sys.exit(0)
    structure()
    ops_small()
    nms_cases()
    metrics_cases()
    validator_case()
    e2e_case()
    augment_case()
    model_small("yolo11n-test.yaml", "edgeline_n_64")
    model_small("yolo11n.yaml", "yolo11n_64")
    for abl in ("GF2Detect", "lineattention", "DSC3K2_Wavelet", "tune"):
        model_small(f"yolo11n-{abl}.yaml", f"{abl.lower()}_n_64", b=1, layers=False)
    model_small("yolo11n-test.yaml", "edgeline_n_96x160", b=1, h=96, w=160, layers=False)
    # l / x scales force c3k=True inside C3k2 / DSC3K2_Wavelet (tasks.py:1069-1072) and use depth > 1: other module paths than n/s/m
    model_small("yolo11l-test.yaml", "edgeline_l_64", b=1, layers=False, gain=LARGE_GAIN)
    model_small("yolo11x.yaml", "yolo11x_64", b=1, layers=False, gain=LARGE_GAIN)
    model_640("yolo11n-test.yaml", "edgeline_n_640")
    model_640("yolo11n.yaml", "yolo11n_640")
