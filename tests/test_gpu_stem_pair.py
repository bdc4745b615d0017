"""-m gpu: layers 0 + 1 as one kernel (ey_stem_pair; reference conv.py:41-59 twice): bit-identical to the two-launch form (MFMA stem +
register-stationary 3x3) at the benchmarked shape and at ragged ones, within the f16 tolerance of the CPU oracle, and taken by the model."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om
import synthdata as synth  # noqa: E402
from gpu_util import check, load_synth, to_dev  # noqa: E402
from test_gpu_dsb import _traced, tuned  # noqa: E402


def _pair():
    from edge_yolo_amd.nn import modules as M
    m0, m1 = M.Conv(3, 16, 3, 2), M.Conv(16, 32, 3, 2)
    sd = {**load_synth(m0, "s0"), **load_synth(m1, "s1")}
    return to_dev(m0, torch.float16), to_dev(m1, torch.float16), sd


@pytest.mark.parametrize("tile", [1, 3, 4, 5])
@pytest.mark.parametrize("b,h,w", [(32, 640, 640), (2, 64, 64), (3, 96, 160), (1, 34, 56), (2, 130, 72), (1, 8, 8), (2, 67, 24)])
def test_stem_pair_bitwise_and_oracle(b, h, w, tile):
    from edge_yolo_amd.nn import _ops
    if b * h * w > 4_000_000 and tile > 1:
        pytest.skip("one tile shape at the full benchmark size is enough")
    m0, m1, sd = _pair()
    x = synth.synth_images(b, h, w, seed=h + w)
    xd = x.to("cuda", torch.float16)
    with tuned(stem_pair=tile):
        got, ker = _traced(lambda: _ops.stem_pair(m0, m1, xd))
    assert ker == ["stem_pair_kernel"], ker
    two, ker2 = _traced(lambda: m1(m0(xd)))
    assert ker2 == ["stem_kernel", "conv3r_kernel<2,2>"], ker2
    assert torch.equal(got, two), f"max |diff| {float((got.float() - two.float()).abs().max())}"
    if b * h * w <= 1_000_000:
        xr = x.half().float()
        check(got, om.conv(sd, "s1", om.conv(sd, "s0", xr, 3, 2), 3, 2), torch.float16, what=f"stem pair {h}x{w}")


def test_model_takes_the_fused_stem_and_matches_two_launch_form():
    import bench
    model, _ = bench.build_model("yolo11n-test.yaml", torch.float16, torch.device("cuda:0"))
    x = synth.synth_images(2, 128, 160).to("cuda", torch.float16)
    (pred, _), ker = _traced(lambda: model(x))
    assert ker[0] == "stem_pair_kernel" and "stem_kernel" not in ker
    with tuned(stem_pair=0):
        (pred2, _), ker2 = _traced(lambda: model(x))
    assert ker2[:2] == ["stem_kernel", "conv3r_kernel<2,2>"] and len(ker2) == len(ker) + 1
    assert torch.equal(pred, pred2)


def test_shapes_outside_fall_back():
    from edge_yolo_amd.nn import _ops, modules as M
    m0, m1, _ = _pair()
    assert _ops.stem_pair(m0, m1, torch.rand(1, 3, 32, 36, device="cuda").half()) is None  # W % 8
    assert _ops.stem_pair(m0.float(), m1.float(), torch.rand(1, 3, 32, 32, device="cuda")) is None  # fp32 parity mode
    wide = to_dev(M.Conv(16, 64, 3, 2), torch.float16)
    load_synth(wide, "s1w")
    assert _ops.stem_pair(m0.half(), wide, torch.rand(1, 3, 32, 32, device="cuda").half()) is None
