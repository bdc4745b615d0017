"""-m gpu: the DSBottleneck pair kernel (ey_dsb_pair; reference block.py:1467-1503) at the benchmarked shapes and at ragged ones:
bit-identical to its two-launch form (2 x ey_dsconv, register-strip kernels) and within the f16 tolerance of the CPU oracle."""
import contextlib

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om
import synthdata as synth  # noqa: E402
from gpu_util import check, load_synth, to_dev  # noqa: E402


@contextlib.contextmanager
def tuned(**kv):
    from edge_yolo_amd import _lib as L
    old = {k: L.lib().ey_tune_get(k.encode()) for k in kv}
    try:
        for k, v in kv.items():
            L.check(L.lib().ey_tune_set(k.encode(), int(v)), "tune")
        yield
    finally:
        for k, v in old.items():
            L.check(L.lib().ey_tune_set(k.encode(), int(v)), "tune")


def _traced(fn):
    """(result, kernel labels of the launches fn() made)."""
    from edge_yolo_amd import profiling
    with profiling.trace() as t:
        y = fn()
    torch.cuda.synchronize()
    return y, [r[0] for r in t.records]


def _module(c, k2, add=True, seed=0):
    from edge_yolo_amd.nn import modules as M
    m = M.DSBottleneck(c, c, shortcut=add, e=1.0, k1=3, k2=k2)
    sd = load_synth(m, f"dsb{seed}")
    return to_dev(m, torch.float16), sd


# (channels, k2, batch, H, W): the eight pairs of the EdgeLine-n step at batch 32 (layers 4, 6, 13, 16, 19, 22), then ragged maps
SHAPES = [(32, 5, 32, 40, 40), (64, 5, 32, 20, 20), (64, 7, 32, 40, 40), (32, 7, 32, 80, 80), (32, 7, 3, 13, 9), (64, 5, 2, 7, 5), (64, 7, 5, 22, 25), (32, 5, 1, 3, 33), (64, 5, 4, 1, 1)]


@pytest.mark.parametrize("c,k2,b,h,w", SHAPES)
def test_pair_is_bit_identical_to_two_launches_and_matches_oracle(c, k2, b, h, w):
    m, sd = _module(c, k2)
    x = (synth.synth_images(b, h, w, seed=c + k2, c=c) - 0.5)
    xd = x.to("cuda", torch.float16)
    got, ker = _traced(lambda: m(xd))
    assert ker == [f"dsb_pair_kernel<3,{k2}>"], ker
    with tuned(dsb_pair=0, tz_kmask=0):  # (C 32, k 7 would otherwise take the Toeplitz-MFMA depthwise stage: another summation order)
        two, ker2 = _traced(lambda: m(xd))
    assert ker2 == ["dsconv_strip_kernel<3>", f"dsconv_strip_kernel<{k2}>"], ker2
    assert torch.equal(got, two), f"max |diff| {float((got.float() - two.float()).abs().max())}"
    check(got, om.dsbottleneck(sd, "dsb0", x.half().float(), 3, k2), torch.float16, what=f"DSBottleneck C{c} k3->k{k2} {h}x{w}")


@pytest.mark.parametrize("rb", [1, 2, 3, 7, 64])
@pytest.mark.parametrize("c,k2", [(32, 7), (64, 5)])
def test_every_band_height_gives_the_same_bits(c, k2, rb):
    """band seams: the first stage is recomputed on the halo rows of each band; rows above / below the map are the second conv's padding."""
    m, _ = _module(c, k2, seed=1)
    xd = (torch.rand(3, c, 11, 14) - 0.5).half().cuda()
    with tuned(dsb_pair=0, tz_kmask=0):
        want = m(xd)
    with tuned(dsb_rb=rb):
        got, ker = _traced(lambda: m(xd))
    assert ker == [f"dsb_pair_kernel<3,{k2}>"]
    assert torch.equal(got, want)


def test_views_and_no_shortcut():
    """input from a channel slice, out= into a channel slice of a wider buffer (the C3k2 concat buffer), shortcut=False."""
    from edge_yolo_amd import _lib as L
    for add in (True, False):
        m, _ = _module(32, 5, add=add, seed=2)
        buf_in = L.empty_nhwc(2, 96, 20, 20, torch.float16, "cuda")
        buf_in.copy_((torch.rand(2, 96, 20, 20) - 0.5).half())
        xin = buf_in[:, 32:64]
        out = L.empty_nhwc(2, 128, 20, 20, torch.float16, "cuda")
        out.zero_()
        with tuned(dsb_pair=0):
            want = m(xin)
        got, ker = _traced(lambda: m(xin, out=out[:, 64:96]))
        assert ker == ["dsb_pair_kernel<3,5>"]
        assert torch.equal(out[:, 64:96], want) and got.data_ptr() == out[:, 64:96].data_ptr()
        assert float(out[:, :64].abs().max()) == 0 and float(out[:, 96:].abs().max()) == 0


def test_shapes_outside_the_kernel_run_as_two_launches():
    from edge_yolo_amd.nn import modules as M
    for c, k1, k2, hw in [(16, 3, 7, 20), (64, 3, 3, 20), (128, 3, 5, 10), (32, 3, 5, 200), (32, 3, 7, 400)]:
        m = M.DSBottleneck(c, c, shortcut=True, e=1.0, k1=k1, k2=k2)
        load_synth(m, "dsbo")
        m = to_dev(m, torch.float16)
        xd = (torch.rand(3, c, hw, hw) - 0.5).half().cuda()
        y, ker = _traced(lambda: m(xd))
        assert len(ker) >= 2 and not any("dsb_pair" in k for k in ker), (c, k1, k2, hw, ker)
        assert y.shape == xd.shape
    # the exact-fp32 parity mode never takes the f16 pair kernel
    m = M.DSBottleneck(32, 32, shortcut=True, e=1.0)
    load_synth(m, "dsbf")
    y, ker = _traced(lambda: to_dev(m, torch.float32)((torch.rand(2, 32, 12, 12) - 0.5).cuda()))
    assert len(ker) == 2


def test_in_place_update_never_takes_the_band_kernel():
    """out aliasing x (the in-place bottleneck chain of C3) would let a workgroup read halo rows another one has already overwritten:
    the wrapper refuses (two launches, which are element-wise safe), and DSC3k alternates between two buffers instead."""
    from edge_yolo_amd import _lib as L
    from edge_yolo_amd.nn import modules as M
    m, _ = _module(32, 5, seed=3)
    buf = L.empty_nhwc(3, 64, 20, 20, torch.float16, "cuda")
    buf.copy_((torch.rand(3, 64, 20, 20) - 0.5).half())
    want = m(buf[:, :32].clone())
    got, ker = _traced(lambda: m(buf[:, :32], out=buf[:, :32]))
    assert len(ker) == 2 and not any("dsb_pair" in k for k in ker), ker
    assert torch.equal(got, want)
    # neighbouring channel slots of one buffer (DSC3K2_Wavelet's chain) are disjoint: the band kernel runs
    got2, ker2 = _traced(lambda: m(buf[:, 32:], out=buf[:, :32]))
    assert ker2 == ["dsb_pair_kernel<3,5>"]
    # DSC3k (n = 2): both pairs as band kernels, same bits as the two-launch form, oracle within the f16 tolerance
    blk = M.DSC3k(64, 64, n=2, k1=3, k2=5)
    sd = load_synth(blk, "dsc3k")
    blk = to_dev(blk, torch.float16)
    x = synth.synth_images(8, 20, 20, seed=5, c=64) - 0.5
    xd = x.to("cuda", torch.float16)
    y, ker3 = _traced(lambda: blk(xd))
    assert ker3.count("dsb_pair_kernel<3,5>") == 2, ker3
    with tuned(dsb_pair=0):
        y0 = blk(xd)
    assert torch.equal(y, y0)
    check(y, om.dsc3k(sd, "dsc3k", x.half().float(), 2, 3, 5), torch.float16, what="DSC3k")
