"""Oracle: YAML graph -> functional torch-CPU fp32 forward over a flat reference-keyed state_dict.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Citations are relative to /root/reference/ultralytics/.
The forward reproduces the *fused* inference path the reference runs under AutoBackend
(nn/autobackend.py:144-155 -> BaseModel.fuse nn/tasks.py:214-242): Conv/DWConv have BN folded with
eps=1e-3 (utils/torch_utils.py:238-265, :410-420); DSConv keeps its BatchNorm un-fused (tasks.py:224
only matches Conv/Conv2/DWConv).
"""
import math
import re
from pathlib import Path

import torch
import torch.nn.functional as F
import yaml

BN_EPS = 1e-3  # utils/torch_utils.py:416


# --------------------------------------------------------------------------- graph (nn/tasks.py:958-1181)
def make_divisible(x, d):  # utils/ops.py:130-143
    return math.ceil(x / d) * d


def guess_scale(path):  # nn/tasks.py:1166-1181
    m = re.search(r"yolo[v]?\d+([nslmx])", Path(path).stem)
    return m.group(1) if m else ""


def load_yaml(path):  # nn/tasks.py:1150-1163  ("yolo11n-test.yaml" -> file "yolo11-test.yaml", scale n)
    path = Path(path)
    unified = re.sub(r"(\d+)([nslmx])(.+)?$", r"\1\3", path.stem) + path.suffix
    f = path.with_name(unified)
    if not f.exists():
        f = path
    d = yaml.safe_load(open(f))
    d["scale"] = guess_scale(path)
    return d


_CSP = {"C3k2", "DSC3K2_Wavelet", "C2PSA", "C2PSA_LinearAttention", "C2f", "C3"}
_CH = _CSP | {"Conv", "SPPF", "DWConv", "DSConv"}
_HEADS = {"Detect", "GF2Detect", "GFLHeadv2_uniH", "E2EDetect"}


def parse_graph(d, ch=3):
    """-> (layers, save, legacy) ; layers[i] = dict(i, f, type, args(list, post-rewrite), c2)."""
    d = dict(d)
    nc = d.get("nc")
    scales = d.get("scales")
    depth, width, max_ch = 1.0, 1.0, float("inf")
    scale = d.get("scale") or ""
    if scales:
        if not scale:
            scale = tuple(scales.keys())[0]  # tasks.py:969-971
        depth, width, max_ch = scales[scale]
    legacy = True  # tasks.py:962
    chs = [ch]
    layers, save = [], []
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        args = list(args)
        for j, a in enumerate(args):  # tasks.py:985-988
            if isinstance(a, str):
                if a == "nc":
                    args[j] = nc
                elif a == "None":
                    args[j] = None
        n = max(round(n * depth), 1) if n > 1 else n  # tasks.py:989
        if m in _CH:
            c1, c2 = chs[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_ch) * width, 8)  # tasks.py:1036
            args = [c1, c2, *args[1:]]
            if m in _CSP:
                args.insert(2, n)  # tasks.py:1067
                n = 1
            if m in {"C3k2", "DSC3K2_Wavelet"}:  # tasks.py:1069-1072
                legacy = False
                if scale in "lx":
                    args[3] = True
        elif m == "Concat":
            c2 = sum(chs[x] for x in f)
        elif m in _HEADS:
            args.append([chs[x] for x in f])
            c2 = None
        else:  # nn.Upsample
            c2 = chs[f]
        assert n == 1, "repeat>1 of non-CSP modules is not used by the target YAMLs"
        layers.append(dict(i=i, f=f, type=m, args=args, c2=c2))
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)  # tasks.py:1142
        if i == 0:
            chs = []
        chs.append(c2)
    return layers, sorted(save), legacy


# --------------------------------------------------------------------------- leaf ops
def _fold(sd, p):
    """fuse_conv_and_bn, utils/torch_utils.py:238-265 (conv has no bias)."""
    w = sd[p + ".conv.weight"]
    g, b, mu, var = (sd[p + ".bn." + k] for k in ("weight", "bias", "running_mean", "running_var"))
    s = g / torch.sqrt(var + BN_EPS)
    return w * s.view(-1, 1, 1, 1), b - mu * s


def conv(sd, p, x, k=1, s=1, g=1, act=True):
    """Conv.forward_fuse, nn/modules/conv.py:57-59; autopad :32-38."""
    w, b = _fold(sd, p)
    y = F.conv2d(x, w, b, stride=s, padding=k // 2, groups=g)
    return F.silu(y) if act else y


def dwconv(sd, p, x, k=3, act=True):
    """DWConv, conv.py:124-129 (groups=gcd(c1,c2); on this path c1==c2)."""
    return conv(sd, p, x, k, 1, g=x.shape[1], act=act)


def dsconv(sd, p, x, k):
    """DSConv.forward, conv.py:101-104: dw kxk (no bias) -> pw 1x1 (no bias) -> BN(eps 1e-3, unfused) -> SiLU."""
    c = x.shape[1]
    y = F.conv2d(x, sd[p + ".dw.weight"], None, 1, (k - 1) // 2, 1, c)
    y = F.conv2d(y, sd[p + ".pw.weight"])
    y = F.batch_norm(y, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                     sd[p + ".bn.bias"], False, 0.0, BN_EPS)
    return F.silu(y)


# --------------------------------------------------------------------------- blocks
def bottleneck(sd, p, x, shortcut=True, k=(3, 3)):
    """Bottleneck, block.py:467-480 (c1==c2, e=1.0 on this path)."""
    y = conv(sd, p + ".cv2", conv(sd, p + ".cv1", x, k[0]), k[1])
    return x + y if shortcut else y


def c3k(sd, p, x, n=2, shortcut=True):
    """C3k(C3), block.py:382-396,868-876: cv3(cat(m(cv1 x), cv2 x)); m = n x Bottleneck(k=(3,3), e=1)."""
    a = conv(sd, p + ".cv1", x)
    for j in range(n):
        a = bottleneck(sd, f"{p}.m.{j}", a, shortcut)
    return conv(sd, p + ".cv3", torch.cat((a, conv(sd, p + ".cv2", x)), 1))


def c3k2(sd, p, x, n, c3k_flag, shortcut=True):
    """C3k2(C2f), block.py:357-379,857-865."""
    y = list(conv(sd, p + ".cv1", x).chunk(2, 1))
    for j in range(n):
        y.append(c3k(sd, f"{p}.m.{j}", y[-1], 2, shortcut) if c3k_flag else bottleneck(sd, f"{p}.m.{j}", y[-1], shortcut))
    return conv(sd, p + ".cv2", torch.cat(y, 1))


def dsbottleneck(sd, p, x, k1, k2):
    """DSBottleneck, block.py:1467-1503 (shortcut and c1==c2 -> add)."""
    return x + dsconv(sd, p + ".cv2", dsconv(sd, p + ".cv1", x, k1), k2)


def dsc3k(sd, p, x, n=2, k1=3, k2=5):
    """DSC3k(C3), block.py:1506-1562."""
    a = conv(sd, p + ".cv1", x)
    for j in range(n):
        a = dsbottleneck(sd, f"{p}.m.{j}", a, k1, k2)
    return conv(sd, p + ".cv3", torch.cat((a, conv(sd, p + ".cv2", x)), 1))


_S = torch.tensor(1.0 / math.sqrt(2.0), dtype=torch.float32)  # pywt haar dec_lo, block.py:3597


def haar_dwt(x):
    """_PywtDWT2D.forward, block.py:3619-3642 for wave='haar' (k=2 -> pad 0, :3617): depthwise stride-2
    conv with taps h0 (x) h0 etc.; h0=[s,s], h1=dec_hi[::-1]=[s,-s] (:3598-3599)."""
    h0 = torch.stack([_S, _S])
    h1 = torch.stack([_S, -_S])
    ks = [torch.einsum("i,j->ij", a, b) for a, b in ((h0, h0), (h0, h1), (h1, h0), (h1, h1))]  # :3603-3606
    w = torch.stack(ks)[:, None]  # (4,1,2,2)
    B, C, H, W = x.shape
    y = F.conv2d(x, w.repeat(C, 1, 1, 1), None, 2, 0, 1, C)
    y = y.view(B, C, 4, y.shape[-2], y.shape[-1])
    return y[:, :, 0], y[:, :, 1], y[:, :, 2], y[:, :, 3]


def wavelet_enhancer(sd, p, b):
    """_WaveletEnhancer.forward, block.py:3685-3710."""
    H, W = b.shape[-2:]
    LL, LH, HL, HH = haar_dwt(b)
    parts = [conv(sd, p + ".f_ll", LL, 1)] + [conv(sd, p + ".f_h", t, 3) for t in (LH, HL, HH)]
    w = F.softplus(sd[p + ".alpha"])
    w = w / (w.sum() + 1e-6)
    ups = [F.interpolate(t, size=(H, W), mode="bilinear", align_corners=False) * w[i] for i, t in enumerate(parts)]
    y = conv(sd, p + ".fuse", torch.cat([b] + ups, 1), 1)
    return b + sd[p + ".gamma"].tanh() * y


def dsc3k2_wavelet(sd, p, x, n, dsc3k_flag, k1=3, k2=7):
    """DSC3K2_Wavelet, block.py:3749-3788."""
    y = list(conv(sd, p + ".cv1", x).chunk(2, 1))
    y[1] = wavelet_enhancer(sd, p + ".wave", y[1])
    for j in range(n):
        y.append(dsc3k(sd, f"{p}.m.{j}", y[-1]) if dsc3k_flag else dsbottleneck(sd, f"{p}.m.{j}", y[-1], k1, k2))
    return conv(sd, p + ".cv2", torch.cat(y, 1))


def sppf(sd, p, x, k=5):
    """SPPF, block.py:204-223."""
    y = [conv(sd, p + ".cv1", x)]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], k, 1, k // 2))
    return conv(sd, p + ".cv2", torch.cat(y, 1))


def linear_attention(sd, p, x, heads):
    """LinearAttention.forward, block.py:3360-3373 (qkv bias, proj no bias via PSABlock_LinearAttention :3412-3449)."""
    B, C, H, W = x.shape
    N, hd = H * W, C // heads
    qkv = F.conv2d(x, sd[p + ".qkv.weight"], sd.get(p + ".qkv.bias")).view(B, 3, heads, hd, N).permute(1, 0, 2, 4, 3)
    q, k, v = qkv[0], qkv[1], qkv[2]
    k = F.softmax(k, dim=-1)
    q = F.softmax(q, dim=-2)
    ctx = k.transpose(-2, -1) @ v
    y = (q @ ctx).transpose(2, 3).reshape(B, C, H, W)
    return F.conv2d(y, sd[p + ".proj.weight"], sd.get(p + ".proj.bias"))


def attention(sd, p, x, heads, attn_ratio=0.5):
    """Attention.forward, block.py:1042-1053."""
    B, C, H, W = x.shape
    N = H * W
    hd = C // heads
    kd = int(hd * attn_ratio)
    qkv = conv(sd, p + ".qkv", x, 1, act=False)
    q, k, v = qkv.view(B, heads, kd * 2 + hd, N).split([kd, kd, hd], dim=2)
    attn = ((q.transpose(-2, -1) @ k) * kd ** -0.5).softmax(dim=-1)
    y = (v @ attn.transpose(-2, -1)).view(B, C, H, W) + conv(sd, p + ".pe", v.reshape(B, C, H, W), 3, g=C, act=False)
    return conv(sd, p + ".proj", y, 1, act=False)


def c2psa(sd, p, x, n, linear):
    """C2PSA block.py:1100-1139 / C2PSA_LinearAttention :3452-3497; PSABlock :3376-3408 / :3412-3449."""
    c = x.shape[1] // 2
    heads = max(1, c // 64)
    a, b = conv(sd, p + ".cv1", x).split((c, c), 1)
    for j in range(n):
        q = f"{p}.m.{j}"
        b = b + (linear_attention(sd, q + ".attn", b, heads) if linear else attention(sd, q + ".attn", b, heads))
        b = b + conv(sd, q + ".ffn.1", conv(sd, q + ".ffn.0", b), act=False)
    return conv(sd, p + ".cv2", torch.cat((a, b), 1))


# --------------------------------------------------------------------------- head
def make_anchors(shapes, strides, off=0.5):
    """utils/tal.py:333-345 (shapes = [(h,w),...])."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sx = torch.arange(w, dtype=torch.float32) + off
        sy = torch.arange(h, dtype=torch.float32) + off
        sy, sx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s)))
    return torch.cat(pts), torch.cat(st)


def dgqp(sd, p, box, reg_max=16, topk=4):
    """GF2Detect._compute_quality_from_logits, head.py:227-243 (stat channel = side*5 + [top1..4, mean])."""
    B, _, H, W = box.shape
    prob = box.view(B, 4, reg_max, H, W).softmax(2)
    tk = torch.topk(prob, topk, dim=2).values
    stat = torch.cat([tk, prob.mean(2, keepdim=True)], 2).view(B, -1, H, W)
    h = F.relu(F.conv2d(stat, sd[p + ".0.weight"], sd[p + ".0.bias"]))
    return torch.sigmoid(F.conv2d(h, sd[p + ".2.weight"], sd[p + ".2.bias"]))


def head_towers(sd, p, xs, quality, legacy=False, reg_max=16, br=""):
    """cv2 / cv3 towers (+ DGQP) of one branch; br = "one2one_" selects the end2end copies (head.py:76-78,220-221)."""
    raw, quals = [], []
    for i, x in enumerate(xs):
        t = conv(sd, f"{p}.{br}cv2.{i}.0", x, 3)
        t = conv(sd, f"{p}.{br}cv2.{i}.1", t, 3)
        box = F.conv2d(t, sd[f"{p}.{br}cv2.{i}.2.weight"], sd[f"{p}.{br}cv2.{i}.2.bias"])
        if legacy:  # head.py:64-65
            u = conv(sd, f"{p}.{br}cv3.{i}.1", conv(sd, f"{p}.{br}cv3.{i}.0", x, 3), 3)
        else:  # head.py:66-75 (E2EDetect builds the same DW structure itself, :812-822)
            u = conv(sd, f"{p}.{br}cv3.{i}.0.1", dwconv(sd, f"{p}.{br}cv3.{i}.0.0", x, 3), 1)
            u = conv(sd, f"{p}.{br}cv3.{i}.1.1", dwconv(sd, f"{p}.{br}cv3.{i}.1.0", u, 3), 1)
        cls = F.conv2d(u, sd[f"{p}.{br}cv3.{i}.2.weight"], sd[f"{p}.{br}cv3.{i}.2.bias"])
        if quality:
            quals.append(dgqp(sd, f"{p}.{br}reg_conf.{i}", box, reg_max))
        raw.append(torch.cat((box, cls), 1))
    return raw, quals


def e2e_postprocess(preds, max_det, nc):
    """Detect.postprocess, head.py:167-189: preds (B, A, 4+nc) -> (B, min(max_det, A), 6) = [box(4), score, class].  Two top-k passes:
    the k anchors with the best class score, then the k best (anchor, class) pairs among them.  (Equal scores: torch.topk's order.)"""
    B, A, _ = preds.shape
    boxes, scores = preds.split([4, nc], dim=-1)
    k = min(max_det, A)
    index = scores.amax(dim=-1).topk(k)[1].unsqueeze(-1)
    boxes = boxes.gather(dim=1, index=index.repeat(1, 1, 4))
    scores = scores.gather(dim=1, index=index.repeat(1, 1, nc))
    scores, index = scores.flatten(1).topk(k)
    i = torch.arange(B)[..., None]
    return torch.cat([boxes[i, index // nc], scores[..., None], (index % nc)[..., None].float()], dim=-1)


def detect_head(sd, p, xs, nc, strides, quality, legacy=False, reg_max=16, e2e=False, max_det=300):
    """Detect.forward/_inference head.py:81-148; GF2Detect._inference_with_quality :301-345;
    GFLHeadv2_uniH.forward :880-908; DFL block.py:87-90; dist2bbox tal.py:348-357.
    Returns (y (B,4+nc,A), raw list of (B,64+nc,H,W)).
    e2e (E2EDetect = GF2Detect with end2end=True, head.py:273-298,799-824): the one2one branch is decoded, boxes stay x1y1x2y2
    (decode_bboxes: xywh and not end2end, :163-165), then Detect.postprocess; returns (y (B,k,6), {"one2many": maps, "one2one": maps})."""
    if e2e:
        raw, quals = head_towers(sd, p, xs, quality, False, reg_max, "one2one_")
        many, _ = head_towers(sd, p, xs, False, False, reg_max, "")
    else:
        raw, quals = head_towers(sd, p, xs, quality, legacy, reg_max)
    B = raw[0].shape[0]
    no = nc + 4 * reg_max
    x_cat = torch.cat([r.view(B, no, -1) for r in raw], 2)
    anchors, st = make_anchors([r.shape[-2:] for r in raw], strides)
    anchors, st = anchors.t(), st.t()
    box, cls = x_cat.split((4 * reg_max, nc), 1)
    A = box.shape[-1]
    d = box.view(B, 4, reg_max, A).transpose(2, 1).softmax(1)
    d = F.conv2d(d, sd[p + ".dfl.conv.weight"]).view(B, 4, A)  # expectation over bins
    lt, rb = d.chunk(2, 1)
    x1y1, x2y2 = anchors.unsqueeze(0) - lt, anchors.unsqueeze(0) + rb
    dbox = (torch.cat((x1y1, x2y2), 1) if e2e else torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1)) * st
    prob = cls.sigmoid()
    if quality:
        q = torch.cat([t.view(B, 1, -1) for t in quals], 2)
        prob = prob * q.clamp(1e-6, 1 - 1e-6)
    y = torch.cat((dbox, prob), 1)
    if e2e:
        return e2e_postprocess(y.permute(0, 2, 1), max_det, nc), {"one2many": many, "one2one": raw}
    return y, raw


# --------------------------------------------------------------------------- whole model
class OracleModel:
    """DetectionModel + _predict_once (nn/tasks.py:152-179,320-370) as a functional interpreter."""

    def __init__(self, yaml_path, sd=None, nc=None, ch=3):
        d = load_yaml(yaml_path) if not isinstance(yaml_path, dict) else dict(yaml_path)
        if nc:
            d["nc"] = nc
        self.nc = d["nc"]
        self.layers, self.save, self.legacy = parse_graph(d, ch)
        self.stride = [8.0, 16.0, 32.0]  # == 256/shape probe result, tasks.py:352-364, for the P3-P5 graphs
        self.sd = sd

    def run_layer(self, L, x):
        sd, p, t, a = self.sd, f"model.{L['i']}", L["type"], L["args"]
        if t == "Conv":
            return conv(sd, p, x, a[2] if len(a) > 2 else 1, a[3] if len(a) > 3 else 1)
        if t == "C3k2":
            return c3k2(sd, p, x, a[2], a[3] if len(a) > 3 else False)
        if t == "DSC3K2_Wavelet":
            return dsc3k2_wavelet(sd, p, x, a[2], a[3] if len(a) > 3 else False)
        if t == "SPPF":
            return sppf(sd, p, x, a[2] if len(a) > 2 else 5)
        if t in ("C2PSA", "C2PSA_LinearAttention"):
            return c2psa(sd, p, x, a[2], t == "C2PSA_LinearAttention")
        if t == "nn.Upsample":
            return F.interpolate(x, scale_factor=a[1], mode=a[2])
        if t == "Concat":
            return torch.cat(x, a[0])
        if t in _HEADS:
            return detect_head(sd, p, x, self.nc, self.stride, t != "Detect", self.legacy, e2e=t == "E2EDetect")
        raise NotImplementedError(t)

    @torch.no_grad()
    def forward(self, x, layer_outputs=None):
        y = []
        for L in self.layers:
            f = L["f"]
            if f != -1:
                x = y[f] if isinstance(f, int) else [x if j == -1 else y[j] for j in f]
            x = self.run_layer(L, x)
            if layer_outputs is not None:
                layer_outputs.append(x)
            y.append(x if L["i"] in self.save else None)
        return x  # (pred, raw)

    @torch.no_grad()
    def forward_augment(self, x):
        """DetectionModel._predict_augment (nn/tasks.py:372-408): scales 1 / .83 / .67 (scale_img, utils/torch_utils.py:423-432:
        bilinear resize to int(h*r) x int(w*r), then 0.447-padding up to ceil(h*r/gs)*gs x ceil(w*r/gs)*gs),
        lr-flip (dim 3) of the middle one, _descale_pred (:388-397), _clip_augmented (:399-408), concat over anchors.  -> (y, None)."""
        import math
        img_h, img_w = x.shape[-2:]
        gs = int(max(self.stride))
        ys = []
        for si, fi in zip((1, 0.83, 0.67), (None, 3, None)):
            xi = x.flip(fi) if fi else x
            if si != 1.0:
                h, w = xi.shape[2:]
                s = (int(h * si), int(w * si))
                xi = F.interpolate(xi, size=s, mode="bilinear", align_corners=False)
                hp, wp = (math.ceil(v * si / gs) * gs for v in (h, w))
                xi = F.pad(xi, [0, wp - s[1], 0, hp - s[0]], value=0.447)
            p = self.forward(xi)[0].clone()
            p[:, :4] /= si
            if fi == 3:
                p[:, 0] = img_w - p[:, 0]
            ys.append(p)
        nl = len(self.stride)
        g = sum(4 ** k for k in range(nl))
        i = (ys[0].shape[-1] // g) * 1
        ys[0] = ys[0][..., :-i]
        i = (ys[-1].shape[-1] // g) * 4 ** (nl - 1)
        ys[-1] = ys[-1][..., i:]
        return torch.cat(ys, -1), None

    __call__ = forward
