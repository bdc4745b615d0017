"""-m gpu: the enhancer tail conv + the DSC3k's stacked cv1|cv2 conv as one launch (ey_conv_pw_pair / conv_pwc_kernel; reference block.py:3700-3710
+ 382-396): the whole DSC3K2_Wavelet block bit-identical to its unchained form and within the f16 tolerance of the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om
import synthdata as synth  # noqa: E402
from gpu_util import check, load_synth, to_dev  # noqa: E402
from test_gpu_dsb import _traced, tuned  # noqa: E402


@pytest.mark.parametrize("c1,c2,b,h,w", [(128, 128, 32, 40, 40), (256, 256, 32, 20, 20), (128, 128, 3, 13, 18), (256, 256, 2, 6, 6)])
def test_block_with_chained_tail(c1, c2, b, h, w):
    from edge_yolo_amd.nn import modules as M
    blk = M.DSC3K2_Wavelet(c1, c2, 1, True)
    sd = load_synth(blk, "wv")
    blk = to_dev(blk, torch.float16)
    x = synth.synth_images(b, h, w, seed=c1 + h, c=c1) - 0.5
    xd = x.to("cuda", torch.float16)
    y, ker = _traced(lambda: blk(xd))
    assert sum(k.startswith("conv_pwc_kernel") for k in ker) == 1, ker
    with tuned(pwc=0):
        y0, ker0 = _traced(lambda: blk(xd))
    assert not any(k.startswith("conv_pwc_kernel") for k in ker0) and len(ker0) == len(ker) + 1
    assert torch.equal(y, y0), f"max |diff| {float((y.float() - y0.float()).abs().max())}"
    if b * h * w <= 2000:
        check(y, om.dsc3k2_wavelet(sd, "wv", x.half().float(), 1, True), torch.float16, what="DSC3K2_Wavelet (c3k)")


def test_model_is_unchanged_by_the_chain():
    import bench
    model, _ = bench.build_model("yolo11n-test.yaml", torch.float16, torch.device("cuda:0"))
    x = synth.synth_images(2, 128, 160).to("cuda", torch.float16)
    (pred, _), ker = _traced(lambda: model(x))
    n = sum(k.startswith("conv_pwc_kernel") for k in ker)
    assert n >= 3, ker
    with tuned(pwc=0):
        (pred2, _), ker2 = _traced(lambda: model(x))
    assert len(ker2) == len(ker) + n
    assert torch.equal(pred, pred2)
