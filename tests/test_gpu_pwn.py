"""-m gpu: the N-split pointwise kernel (conv_pwn_kernel, behind ey_conv2d; reference conv.py:41-59 at the 20x20 / 40x40 layers) against
the weight-stationary kernel it replaces (bit-identical: same k-step order) and the CPU oracle, at the benchmarked shapes and ragged ones."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om
import synthdata as synth  # noqa: E402
from gpu_util import check, load_synth, to_dev  # noqa: E402
from test_gpu_dsb import _traced, tuned  # noqa: E402


# (batch, H, W, source channels, Cout, act): layers 4/6/8 cv1, the neck's 2-source cv2s, SPPF cv2, C2PSA qkv (no activation), ragged maps
CASES = [(32, 40, 40, (128,), 128, True), (32, 20, 20, (256,), 256, True), (32, 20, 20, (128,), 384, False), (32, 40, 40, (128, 64), 128, True),
         (32, 20, 20, (256, 128), 256, True), (32, 20, 20, (512,), 256, True), (3, 19, 23, (128, 256), 128, True), (5, 17, 13, (192,), 256, True)]


@pytest.mark.parametrize("b,h,w,cs,cout,act", CASES)
def test_pwn_equals_weight_stationary_kernel_and_oracle(b, h, w, cs, cout, act):
    from edge_yolo_amd import _lib as L
    from edge_yolo_amd.nn import _ops, modules as M
    cin = sum(cs)
    m = M.Conv(cin, cout, 1, 1, act=act)
    sd = load_synth(m, "pwn")
    m = to_dev(m, torch.float16)
    x = synth.synth_images(b, h, w, seed=cin + cout, c=cin) - 0.5
    # sources as channel slices of wider buffers (virtual concat)
    srcs, c0 = [], 0
    for c in cs:
        buf = L.empty_nhwc(b, c + 32, h, w, torch.float16, "cuda")
        buf[:, :c].copy_(x[:, c0:c0 + c].half())
        srcs.append(buf[:, :c])
        c0 += c
    code = L.ACT_SILU if act else L.ACT_NONE
    with tuned(pw_m=0):  # (the lean pointwise kernel would take the small ragged cases first)
        got, ker = _traced(lambda: _ops.conv2d(m, srcs, m.folded, 1, 1, 0, code))
    assert len(ker) == 1 and ker[0].startswith("conv_pwn_kernel"), ker
    with tuned(pwn=0, pw_m=0):
        ref, ker2 = _traced(lambda: _ops.conv2d(m, srcs, m.folded, 1, 1, 0, code))
    assert not ker2[0].startswith("conv_pwn_kernel")
    assert torch.equal(got, ref), f"max |diff| {float((got.float() - ref.float()).abs().max())}"
    for ntw in (1, 2):
        with tuned(pwn_ntw=ntw, pw_m=0):
            alt = _ops.conv2d(m, srcs, m.folded, 1, 1, 0, code)
        assert torch.equal(alt, ref), f"ntw={ntw}"
    if b * h * w <= 20000:
        want = om.conv(sd, "pwn", x.half().float(), 1, 1) if act else torch.nn.functional.conv2d(x.half().float(), *_fold(sd))
        check(got, want, torch.float16, what=f"1x1 {cs}->{cout} {h}x{w}")


def _fold(sd):
    w, bn_w, bn_b, mean, var = sd["pwn.conv.weight"], sd["pwn.bn.weight"], sd["pwn.bn.bias"], sd["pwn.bn.running_mean"], sd["pwn.bn.running_var"]
    scale = bn_w / torch.sqrt(var + 1e-3)
    return w * scale.view(-1, 1, 1, 1), bn_b - mean * scale


def test_pwn_upsampled_source():
    """neck: cat(upsample(x), skip) folded into the 1x1 conv -- the first source is read through a nearest x2 upsample."""
    from edge_yolo_amd import _lib as L
    from edge_yolo_amd.nn import _ops, modules as M
    m = M.Conv(384, 128, 1, 1)
    load_synth(m, "pwnu")
    m = to_dev(m, torch.float16)
    lo = L.empty_nhwc(8, 256, 10, 10, torch.float16, "cuda"); lo.copy_((torch.rand(8, 256, 10, 10) - 0.5).half())
    hi = L.empty_nhwc(8, 128, 20, 20, torch.float16, "cuda"); hi.copy_((torch.rand(8, 128, 20, 20) - 0.5).half())
    with tuned(pw_m=0):
        got, ker = _traced(lambda: _ops.conv2d(m, [lo, hi], m.folded, 1, 1, 0, L.ACT_SILU, up=[1, 0]))
    assert ker[0].startswith("conv_pwn_kernel"), ker
    with tuned(pwn=0, pw_m=0):
        ref = _ops.conv2d(m, [lo, hi], m.folded, 1, 1, 0, L.ACT_SILU, up=[1, 0])
    assert torch.equal(got, ref)
