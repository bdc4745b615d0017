"""-m gpu: every HIP operator against the CPU oracle on seeded inputs (through the Python modules, i.e. through the
C ABI).  fp32 = parity gate, fp16 = throughput mode (tolerances: tests/gpu_util.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om
import synthdata as synth  # noqa: E402
from gpu_util import check, load_synth, to_dev  # noqa: E402

DT = [torch.float32, torch.float16]


def _x(b, c, h, w, dtype, seed=0):
    x = synth.synth_images(b, h, w, seed=seed, c=c) - 0.5
    return x, x.to("cuda", dtype)


@pytest.fixture(scope="module")
def M():
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd.nn import modules
    return modules


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c1,c2,k,s,h,w", [
    (16, 8, 1, 1, 12, 10), (16, 8, 3, 1, 12, 10), (32, 32, 1, 1, 16, 16), (48, 64, 1, 1, 20, 12), (16, 32, 3, 2, 32, 32),
    (64, 64, 3, 2, 16, 16), (64, 80, 1, 1, 9, 7), (80, 80, 1, 1, 9, 7), (96, 128, 1, 1, 8, 8), (128, 128, 3, 2, 14, 10),
    (128, 256, 3, 2, 8, 8), (192, 64, 1, 1, 5, 5), (256, 64, 3, 1, 6, 6), (384, 256, 1, 1, 4, 4), (128, 384, 1, 1, 4, 4),
    (512, 256, 1, 1, 3, 3), (64, 32, 3, 1, 13, 9), (24, 40, 3, 1, 7, 7), (64, 12, 1, 1, 6, 5)])
def test_conv(M, dtype, c1, c2, k, s, h, w):
    m = M.Conv(c1, c2, k, s)
    sd = load_synth(m, "cv")
    x, xd = _x(3, c1, h, w, dtype)
    want = om.conv(sd, "cv", x, k, s)
    check(to_dev(m, dtype)(xd), want, dtype, what=f"Conv {c1}->{c2} k{k}s{s}")
    m.fuse_bn()  # fused parameters must give the same result (BaseModel.fuse)
    check(m(xd), want, dtype, what="fused")


@pytest.mark.parametrize("dtype", DT)
def test_conv_views_residual(M, dtype):
    """out= into a channel slice of a wider buffer, input from a channel slice, residual add."""
    from edge_yolo_amd import _lib as L
    m = M.Conv(32, 32, 3, 1)
    sd = load_synth(m, "cvr")
    x, xd = _x(2, 64, 10, 12, dtype)
    m = to_dev(m, dtype)
    buf = L.empty_nhwc(2, 96, 10, 12, dtype, "cuda")
    buf.zero_()
    xin = L.as_nhwc(xd)[:, 32:]
    m(xin, out=buf[:, 32:64], res=xin)
    want = x[:, 32:] + om.conv(sd, "cvr", x[:, 32:], 3, 1)
    check(buf[:, 32:64], want, dtype)
    assert float(buf[:, :32].abs().max()) == 0 and float(buf[:, 64:].abs().max()) == 0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("h,w", [(64, 64), (34, 50)])
def test_stem(M, dtype, h, w):
    m = M.Conv(3, 16, 3, 2)
    sd = load_synth(m, "stem")
    x = synth.synth_images(2, h, w)
    got = to_dev(m, dtype)(x.to("cuda", dtype))
    check(got, om.conv(sd, "stem", x, 3, 2), dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("k", [3, 5, 7])
def test_dsconv(M, dtype, k):
    m = M.DSConv(16, 24, k)
    sd = load_synth(m, "ds")
    x, xd = _x(2, 16, 9, 11, dtype)
    check(to_dev(m, dtype)(xd), om.dsconv(sd, "ds", x, k), dtype)


@pytest.mark.parametrize("dtype", DT)
def test_dwconv(M, dtype):
    m = M.DWConv(80, 80, 3)
    sd = load_synth(m, "dw")
    x, xd = _x(2, 80, 7, 9, dtype)
    check(to_dev(m, dtype)(xd), om.dwconv(sd, "dw", x, 3), dtype)


@pytest.mark.parametrize("dtype", DT)
def test_conv_direct_fallback(M, dtype):
    """channel counts the MFMA kernel does not take go through the generic direct kernel."""
    m = M.Conv(20, 12, 3, 1)
    sd = load_synth(m, "cd")
    x, xd = _x(2, 20, 6, 7, dtype)
    check(to_dev(m, dtype)(xd), om.conv(sd, "cd", x, 3, 1), dtype)


@pytest.mark.parametrize("dtype", DT)
def test_dwt(M, dtype):
    from edge_yolo_amd.nn.modules.block import _PywtDWT2D
    x, xd = _x(2, 16, 10, 14, dtype)
    got = _PywtDWT2D()(xd)
    for g, w in zip(got, om.haar_dwt(x)):
        check(g, w, dtype)
    q = torch.tensor([[[[1., 2.], [3., 4.]]]]).repeat(1, 8, 1, 1).to("cuda", dtype)
    ll, lh, hl, hh = _PywtDWT2D()(q)
    assert [round(float(t[0, 0, 0, 0])) for t in (ll, lh, hl, hh)] == [5, -1, -2, 0]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("h,w", [(10, 14), (9, 13), (3, 5)])
def test_wavelet_enhancer(M, dtype, h, w):
    """even maps (exact x2 upsample) and odd maps (DWT floors, bilinear resize by a non-integer ratio)."""
    from edge_yolo_amd.nn.modules.block import _WaveletEnhancer
    m = _WaveletEnhancer(16)
    sd = load_synth(m, "enh")
    x, xd = _x(2, 16, h, w, dtype)
    check(to_dev(m, dtype)(xd), om.wavelet_enhancer(sd, "enh", x), dtype)


@pytest.mark.parametrize("c,h,w,b", [(16, 160, 160, 2), (32, 80, 80, 3), (64, 40, 40, 2), (128, 20, 20, 3), (64, 21, 37, 2), (16, 6, 4, 1), (32, 50, 34, 2),
                                     (128, 22, 30, 2), (128, 6, 10, 1), (128, 40, 40, 1)])
def test_wavelet_z_kernel_vs_oracle_and_unfused(M, c, h, w, b):
    """ey_wavelet_z (f16: Haar DWT + f_ll / f_h sub-band convs + the Z contraction in ONE kernel) at the network's real (c, map) pairs,
    odd maps and tiles that overhang the map: against the fp32 oracle enhancer (f16 tolerance) and against the three-launch f16 form
    (same rounding points: tight)."""
    from edge_yolo_amd.nn.modules.block import _WaveletEnhancer
    from edge_yolo_amd import profiling
    m = _WaveletEnhancer(c)
    sd = load_synth(m, f"enh{c}")
    x, xd = _x(b, c, h, w, torch.float16)
    mh = to_dev(m, torch.float16)
    with profiling.trace() as t:
        got = mh(xd)
    assert [r[0] for r in t.records][0] == "wavelet_z_kernel" and len(t.records) == 2, [r[0] for r in t.records]
    mh.fused_z = False
    three = mh(xd)
    torch.cuda.synchronize()
    scale = float(three.float().abs().max())
    assert float((got.float() - three.float()).abs().max()) <= 3e-3 * scale
    check(got, om.wavelet_enhancer(sd, f"enh{c}", x), torch.float16, scale=max(1.0, scale))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c1,c2,dsc3k,e", [(32, 64, False, 0.25), (128, 128, True, 0.5), (384, 128, False, 0.5)])
def test_dsc3k2_wavelet(M, dtype, c1, c2, dsc3k, e):
    m = M.DSC3K2_Wavelet(c1, c2, 1, dsc3k, e)
    sd = load_synth(m, "blk")
    x, xd = _x(2, c1, 12, 8, dtype)
    check(to_dev(m, dtype)(xd), om.dsc3k2_wavelet(sd, "blk", x, 1, dsc3k), dtype, scale=2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c3k", [False, True])
def test_c3k2(M, dtype, c3k):
    m = M.C3k2(64, 128, 1, c3k, 0.5)
    sd = load_synth(m, "c3k2")
    x, xd = _x(2, 64, 9, 9, dtype)
    check(to_dev(m, dtype)(xd), om.c3k2(sd, "c3k2", x, 1, c3k), dtype, scale=2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("h,w", [(7, 9), (20, 20)])
def test_sppf(M, dtype, h, w):
    m = M.SPPF(64, 96, 5)
    sd = load_synth(m, "sppf")
    x, xd = _x(2, 64, h, w, dtype)
    check(to_dev(m, dtype)(xd), om.sppf(sd, "sppf", x), dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c1,h,w", [(256, 20, 20), (80, 13, 9), (128, 40, 40), (48, 7, 21)])
def test_sppf_channel_groups(M, dtype, c1, h, w):
    """SPPF pooling at the network's own shape (256 ch, 20x20: whole 128-byte lines per workgroup), at channel counts that only allow
    narrower groups (c_ = 40, 24) and on a 40x40 plane (1280x1280 inputs: the group shrinks until two planes fit LDS)."""
    m = M.SPPF(c1, c1, 5)
    sd = load_synth(m, "sppfg")
    x, xd = _x(3, c1, h, w, dtype, seed=4)
    check(to_dev(m, dtype)(xd), om.sppf(sd, "sppfg", x), dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("linear", [True, False])
@pytest.mark.parametrize("h,w", [(4, 4), (20, 20)])
def test_c2psa(M, dtype, linear, h, w):
    m = (M.C2PSA_LinearAttention if linear else M.C2PSA)(256, 256, 1)
    sd = load_synth(m, "psa")
    x, xd = _x(2, 256, h, w, dtype)
    check(to_dev(m, dtype)(xd), om.c2psa(sd, "psa", x, 1, linear), dtype, scale=2)


@pytest.mark.parametrize("dtype", DT)
def test_upsample_concat(M, dtype):
    x, xd = _x(2, 32, 5, 6, dtype)
    y, yd = _x(2, 16, 10, 12, dtype, seed=1)
    up = M.Upsample(None, 2, "nearest")(xd)
    check(up, torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest"), dtype)
    cat = M.Concat(1)([up, yd])
    check(cat, torch.cat([torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest"), y], 1), dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("head,nc", [("GFLHeadv2_uniH", 80), ("Detect", 80), ("GFLHeadv2_uniH", 10)])
def test_head(M, dtype, head, nc):
    ch = (64, 128, 256)
    m = getattr(M, head)(nc, ch)
    m.stride = torch.tensor([8., 16., 32.])
    sd = load_synth(m, "model.23")
    xs = [synth.synth_images(2, h, w, seed=i, c=c) - 0.5 for i, (c, (h, w)) in enumerate(zip(ch, [(8, 12), (4, 6), (2, 3)]))]
    want, raw = om.detect_head(sd, "model.23", [t.clone() for t in xs], nc, [8., 16., 32.], head != "Detect")
    got, graw = to_dev(m, dtype)([t.to("cuda", dtype) for t in xs])
    assert got.dtype == torch.float32 and tuple(got.shape) == tuple(want.shape)
    check(got[:, :4], want[:, :4], dtype, scale=40 if dtype == torch.float16 else 5, what="boxes")
    check(got[:, 4:], want[:, 4:], dtype, what="scores")
    for a, b in zip(graw, raw):
        check(a, b, dtype, what="raw")


def test_no_cpu_fallback(M):
    from edge_yolo_amd._lib import HipLibraryError
    m = M.Conv(16, 16, 1)
    with pytest.raises(HipLibraryError):
        m.eval()(torch.zeros(1, 16, 4, 4))


@pytest.mark.parametrize("c1,c2,k,s,h,w", [
    (128, 128, 3, 2, 40, 36), (128, 256, 3, 2, 20, 20), (256, 64, 3, 1, 20, 20), (128, 64, 3, 1, 24, 40), (64, 64, 3, 1, 41, 37), (64, 64, 3, 2, 33, 47),
    (16, 8, 3, 1, 30, 30), (32, 16, 3, 1, 17, 33), (64, 32, 3, 1, 10, 10), (16, 32, 3, 2, 64, 48), (32, 32, 1, 1, 176, 160), (48, 64, 1, 1, 176, 160),
    (96, 128, 1, 1, 176, 160), (80, 80, 1, 1, 176, 160), (64, 80, 1, 1, 90, 70), (256, 256, 1, 1, 20, 20), (128, 64, 1, 1, 20, 20), (384, 256, 1, 1, 20, 20)])
def test_conv_f16_kernels_agree_with_exact_f32_kernels(M, c1, c2, k, s, h, w):
    """The f16 throughput mode dispatches to different kernels (3x3 tile, lean / register-stationary pointwise) than the f32
    parity mode.  On the SAME f16-representable weights and inputs both accumulate exact products in fp32, so they may differ
    only by summation order and the final f16 rounding: a tight check of the f16-only kernels' indexing at sizes that reach them."""
    torch.manual_seed(c1 * 31 + c2)
    m = M.Conv(c1, c2, k, s)
    load_synth(m, "cvx")
    m.fuse_bn()
    with torch.no_grad():
        m.conv.weight.copy_(m.conv.weight.half().float())
    x = (torch.rand(4, c1, h, w) - 0.5).half()
    want = to_dev(m, torch.float32)(x.float().cuda()).float().cpu()
    got = to_dev(m, torch.float16)(x.cuda()).float().cpu()
    scale = float(want.abs().max())
    torch.testing.assert_close(got, want, rtol=2e-3, atol=2e-3 * scale)


@pytest.mark.parametrize("c,k,h,w", [(16, 3, 40, 36), (16, 7, 40, 40), (32, 3, 21, 19), (32, 5, 20, 20), (32, 7, 33, 30), (64, 3, 20, 20), (64, 5, 20, 20), (64, 7, 22, 25)])
def test_dsconv_f16_kernel_agrees_with_exact_f32_kernel(M, c, k, h, w):
    """f16 mode runs the register-strip DSConv kernel, f32 mode the LDS-tile kernel; same f16-representable weights and inputs ->
    they may differ only by fp32 summation order, the f16 rounding of the depthwise intermediate and of the output."""
    torch.manual_seed(c * 7 + k)
    m = M.DSConv(c, c, k)
    load_synth(m, "dsx")
    with torch.no_grad():
        for prm in m.parameters():
            prm.copy_(prm.half().float())
    x = (torch.rand(3, c, h, w) - 0.5).half()
    want = to_dev(m, torch.float32)(x.float().cuda()).float().cpu()
    got = to_dev(m, torch.float16)(x.cuda()).float().cpu()
    scale = float(want.abs().max())
    torch.testing.assert_close(got, want, rtol=4e-3, atol=4e-3 * scale)


@pytest.mark.parametrize("n_hw", [(20, 20), (13, 9), (40, 40), (7, 5)])
def test_linear_attention_mfma_agrees_with_f32_kernel(n_hw):
    """f16 mode runs the MFMA linear-attention kernel (head_dim 64), f32 mode the fp32 VALU kernel: same f16-representable qkv ->
    they differ by the f16 rounding of softmax(k), ctx and softmax(q) (the reference's half-precision matmul operands) only."""
    from edge_yolo_amd.nn import _ops
    from edge_yolo_amd import _lib as L
    h, w = n_hw
    torch.manual_seed(h * 100 + w)
    qkv = L.empty_nhwc(3, 384, h, w, torch.float16, "cuda")
    qkv.copy_((torch.randn(3, 384, h, w) * 1.5).half())
    got = _ops.linear_attention(qkv, 2).float().cpu()
    q32 = L.empty_nhwc(3, 384, h, w, torch.float32, "cuda")
    q32.copy_(qkv.float())
    want = _ops.linear_attention(q32, 2).float().cpu()
    torch.testing.assert_close(got, want, rtol=1e-2, atol=2e-3 * float(want.abs().max()))


@pytest.mark.parametrize("h,w,c2", [(64, 64, 16), (96, 128, 16), (70, 200, 32), (33, 40, 48), (640, 640, 16)])
def test_stem_mfma_agrees_with_f32_kernel(M, h, w, c2):
    """f16 mode runs the MFMA stem (image patch in LDS, 2-byte tap gathers, one 16x16x32 MFMA per 16 pixels); f32 mode the direct
    kernel.  Same f16-representable image and weights -> only summation order and the output rounding differ."""
    torch.manual_seed(h + w + c2)
    m = M.Conv(3, c2, 3, 2)
    load_synth(m, "stemx")
    m.fuse_bn()
    with torch.no_grad():
        m.conv.weight.copy_(m.conv.weight.half().float())
    x = torch.rand(2, 3, h, w).half()
    want = to_dev(m, torch.float32)(x.float().cuda()).float().cpu()
    got = to_dev(m, torch.float16)(x.cuda()).float().cpu()
    torch.testing.assert_close(got, want, rtol=2e-3, atol=2e-3 * float(want.abs().max()))


@pytest.mark.parametrize("n_hw", [(20, 20), (13, 9), (16, 16), (7, 5), (20, 19)])
def test_softmax_attention_mfma_agrees_with_f32_kernel(n_hw):
    """f16 mode runs the MFMA softmax-attention kernel (key_dim 32, head_dim 64, N <= 416), f32 mode the VALU kernel; same
    f16-representable qkv -> they differ by the f16 rounding of the probabilities (the reference's half-precision matmul operand)."""
    from edge_yolo_amd.nn import _ops
    from edge_yolo_amd import _lib as L
    h, w = n_hw
    torch.manual_seed(h * 10 + w)
    qkv = L.empty_nhwc(3, 256, h, w, torch.float16, "cuda")  # 2 heads x (32 + 32 + 64)
    qkv.copy_((torch.randn(3, 256, h, w) * 1.2).half())
    got = _ops.softmax_attention(qkv, 2, 32, 64, 32 ** -0.5).float().cpu()
    q32 = L.empty_nhwc(3, 256, h, w, torch.float32, "cuda")
    q32.copy_(qkv.float())
    want = _ops.softmax_attention(q32, 2, 32, 64, 32 ** -0.5).float().cpu()
    torch.testing.assert_close(got, want, rtol=1e-2, atol=3e-3 * float(want.abs().max()))


@pytest.mark.parametrize("cin,cout,h,w", [(80, 80, 40, 40), (96, 80, 23, 17), (72, 68, 20, 20), (80, 80, 80, 80)])
def test_pointwise_chain_kernel_agrees_with_two_f32_convs(M, cin, cout, h, w):
    """ey_conv_pw_chain (f16: two 1x1 convs in one register-only kernel, the second contraction in the first GEMM's register order)
    against the same two convs run one after the other by the exact-f32 kernels on the same f16-representable weights / input."""
    import torch.nn as nn
    from edge_yolo_amd.nn import _ops
    from edge_yolo_amd import _lib as L
    from edge_yolo_amd.nn.modules.conv import fold_bn
    from edge_yolo_amd.nn.modules.head import _Plain
    torch.manual_seed(cin * 3 + cout)
    c1 = M.Conv(cin, 80, 1)
    load_synth(c1, "ch1")
    c1.fuse_bn()
    c2 = nn.Conv2d(80, cout, 1)
    with torch.no_grad():
        c1.conv.weight.copy_(c1.conv.weight.half().float())
        c2.weight.copy_((c2.weight * 3).half().float())
    x = (torch.rand(2, cin, h, w) - 0.5).half()
    c1f, c2f = to_dev(c1, torch.float32), c2.cuda().float()
    mid = c1f(x.float().cuda())
    want = _Plain(c2f).run(mid.half().float(), L.empty_nhwc(2, cout, h, w, torch.float32, "cuda")).float().cpu()  # mid rounded to f16 like the fused kernel
    c1h, c2h = to_dev(c1, torch.float16), c2.cuda().half()
    out = L.empty_nhwc(2, cout, h, w, torch.float16, "cuda")
    got = _ops.conv_pw_chain(_Plain(c2h), x.cuda(), c1h.folded, L.ACT_SILU, lambda: fold_bn(c2h.weight, c2h.bias, None), L.ACT_NONE, out)
    assert got is not None
    torch.testing.assert_close(got.float().cpu(), want, rtol=3e-3, atol=3e-3 * float(want.abs().max()))


@pytest.mark.parametrize("c1,c2,s,hw,B", [(64, 64, 2, 160, 4), (128, 128, 2, 80, 4), (128, 256, 2, 40, 8), (64, 64, 2, 80, 5), (128, 128, 2, 40, 7),
                                          (64, 64, 1, 80, 3), (128, 64, 1, 40, 6), (256, 64, 1, 20, 9), (64, 64, 1, 20, 32), (64, 32, 1, 20, 11), (256, 64, 1, 21, 3)])
def test_conv3_stream_kernel_at_real_shapes(M, c1, c2, s, hw, B):
    """The 3x3 stream kernel (conv3s_kernel: LDS-resident weights, pixel fragments straight from L2, ring prefetch across tiles) at the
    network's own K >= 576 shapes (layers 3/5/7/17/20, Detect box towers; odd map / batch sizes for the M tail): against the fp32 oracle
    and against the kernels it replaces (tunable c3s = 0) -- the two f16 paths differ only by fp32 summation order."""
    from edge_yolo_amd import _lib as L
    m = M.Conv(c1, c2, 3, s)
    sd = load_synth(m, "c3s")
    x = synth.synth_images(B, hw, hw, seed=3, c=c1) - 0.5
    xd = x.to("cuda", torch.float16)
    m = to_dev(m, torch.float16)
    m.fuse_bn()
    lib = L.lib()
    try:
        L.check(lib.ey_tune_set(b"c3s", 2), "tune")  # 2 = every shape the kernel takes (the default rule keeps it to the big stride-2 layers)
        got = m(xd)
        assert lib.ey_conv_last_variant() // 1000 == 8, f"stream kernel not dispatched (variant {lib.ey_conv_last_variant()})"
        L.check(lib.ey_tune_set(b"c3s", 0), "tune")
        old = m(xd)
        assert lib.ey_conv_last_variant() // 1000 != 8
    finally:
        L.check(lib.ey_tune_set(b"c3s", 1), "tune")
    torch.cuda.synchronize()
    assert float((got.float() - old.float()).abs().max()) <= 2e-3 * max(1.0, float(old.float().abs().max()))
    if B * hw * hw <= 60000:
        check(got, om.conv(sd, "c3s", x.half().float(), 3, s), torch.float16, what=f"Conv {c1}->{c2} k3s{s} {hw}x{hw}")


def test_conv3_stream_kernel_views_residual_act(M):
    """stream kernel with out= into a channel slice, a channel-sliced input view, residual add and no activation."""
    from edge_yolo_amd import _lib as L
    m = M.Conv(64, 64, 3, 1, act=False)
    sd = load_synth(m, "c3sv")
    x = synth.synth_images(2, 22, 18, seed=5, c=128) - 0.5
    xd = x.to("cuda", torch.float16)
    m = to_dev(m, torch.float16)
    buf = L.empty_nhwc(2, 192, 22, 18, torch.float16, "cuda")
    buf.zero_()
    xin = L.as_nhwc(xd)[:, 64:]
    L.check(L.lib().ey_tune_set(b"c3s", 2), "tune")
    try:
        m(xin, out=buf[:, 64:128], res=xin)
        assert L.lib().ey_conv_last_variant() // 1000 == 8
    finally:
        L.check(L.lib().ey_tune_set(b"c3s", 1), "tune")
    want = x[:, 64:].half().float() + om.conv(sd, "c3sv", x[:, 64:].half().float(), 3, 1, act=False)
    check(buf[:, 64:128], want, torch.float16)
    assert float(buf[:, :64].abs().max()) == 0 and float(buf[:, 128:].abs().max()) == 0


@pytest.mark.parametrize("c2,hw,B", [(64, 80, 3), (64, 40, 6), (64, 20, 32), (128, 40, 5), (64, 21, 7), (64, 9, 2), (192, 24, 3)])
def test_conv3_persistent_tile_kernel(M, c2, hw, B):
    """conv3p_kernel (Cin = 64, stride 1: weights LDS-resident for the life of a persistent workgroup, all-channel halo, epilogue of tile i
    deferred into the K loop of tile i + 1) at the Detect box-tower shapes, odd maps (partial tiles, out-of-image halo) and several channel
    tiles: against the fp32 oracle and against the tile kernel it replaces (tunable c3p = 0)."""
    from edge_yolo_amd import _lib as L
    m = M.Conv(64, c2, 3, 1)
    sd = load_synth(m, "c3p")
    x = synth.synth_images(B, hw, hw + (3 if hw in (21, 9) else 0), seed=7, c=64) - 0.5
    xd = x.to("cuda", torch.float16)
    m = to_dev(m, torch.float16)
    m.fuse_bn()
    lib = L.lib()
    try:
        L.check(lib.ey_tune_set(b"c3p", 2), "tune")
        got = m(xd)
        assert lib.ey_conv_last_variant() // 1000 == 9, f"persistent tile kernel not dispatched (variant {lib.ey_conv_last_variant()})"
        L.check(lib.ey_tune_set(b"c3p", 0), "tune")
        old = m(xd)
        assert lib.ey_conv_last_variant() // 1000 != 9
    finally:
        L.check(lib.ey_tune_set(b"c3p", 1), "tune")
    torch.cuda.synchronize()
    assert float((got.float() - old.float()).abs().max()) <= 2e-3 * max(1.0, float(old.float().abs().max()))
    check(got, om.conv(sd, "c3p", x.half().float(), 3, 1), torch.float16, what=f"Conv 64->{c2} k3s1 {hw}x{hw}")


def test_conv3_persistent_tile_kernel_views_residual(M):
    """conv3p with out= into a channel slice, a channel-sliced input, residual add, no activation (the deferred epilogue's operands)."""
    from edge_yolo_amd import _lib as L
    m = M.Conv(64, 64, 3, 1, act=False)
    sd = load_synth(m, "c3pv")
    x = synth.synth_images(3, 26, 30, seed=8, c=128) - 0.5
    xd = x.to("cuda", torch.float16)
    m = to_dev(m, torch.float16)
    buf = L.empty_nhwc(3, 192, 26, 30, torch.float16, "cuda")
    buf.zero_()
    xin = L.as_nhwc(xd)[:, 64:]
    L.check(L.lib().ey_tune_set(b"c3p", 2), "tune")
    try:
        m(xin, out=buf[:, 64:128], res=xin)
        assert L.lib().ey_conv_last_variant() // 1000 == 9
    finally:
        L.check(L.lib().ey_tune_set(b"c3p", 1), "tune")
    want = x[:, 64:].half().float() + om.conv(sd, "c3pv", x[:, 64:].half().float(), 3, 1, act=False)
    check(buf[:, 64:128], want, torch.float16)
    assert float(buf[:, :64].abs().max()) == 0 and float(buf[:, 128:].abs().max()) == 0
