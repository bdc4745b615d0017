"""-m gpu: block programs (csrc/block.hip through nn/_block.py; opt-in, `block_fusion = True`): the 20x20 layer runs of the network
(layers 7-10 and 20-22 of the 24-layer YAMLs, the 20x20 Detect towers) executed by ONE launch each.

Every fused run is checked at its REAL shapes (640x640 input -> 20x20 maps): against the CPU oracle's per-layer outputs with the f16
tolerance, and against the per-layer HIP kernels (same f16 rounding points -> tight agreement); stage-level unit tests cover each
operator of the block kernel against the stand-alone kernel on odd sizes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om  # noqa: E402
import synthdata as synth  # noqa: E402


def _build(name, nc=None):
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd.nn.tasks import DetectionModel
    m = DetectionModel(name, nc=nc)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict(sd)
    return m.cuda().fuse().half().eval(), sd


def _layers(m, x, fusion):
    m.block_fusion = fusion
    m.model[-1].block_fusion = fusion
    outs = {}
    state = (x, [])
    for i in range(len(m.model) - 1):
        state = m.forward_layers(state, i, i + 1) if not fusion else state
    if fusion:  # the runs must be entered at their first layer: whole ranges
        state = m.forward_layers((x, []), 0, len(m.model) - 1)
    xx, y = state
    for i, t in enumerate(y):
        if torch.is_tensor(t):
            outs[i] = t.float().cpu()
    outs[len(m.model) - 2] = xx.float().cpu()
    return outs, state


@pytest.mark.parametrize("name", ["yolo11n-test.yaml", "yolo11n.yaml"])
def test_block_runs_at_real_shapes_vs_oracle_and_per_layer_kernels(cfg_dir, name):
    from edge_yolo_amd.nn import _ops
    m, sd = _build(name)
    x = synth.synth_images(4, 640, 640)
    xh = x.cuda().half()
    # how many launches does each mode take?  (trace = one record per launch)
    from edge_yolo_amd import profiling
    with profiling.trace() as t_on:
        on, _ = _layers(m, xh, True)
    with profiling.trace() as t_off:
        off, _ = _layers(m, xh, False)
    torch.cuda.synchronize()
    n_on, n_off = len(t_on.records), len(t_off.records)
    blocks = [r[0] for r in t_on.records if r[0].startswith("block_kernel")]
    print(f"\n[{name}] launches: per-layer {n_off}, with block programs {n_on} ({blocks})")
    assert len(blocks) >= 1 and n_on < n_off
    # oracle per-layer outputs (fp32) of the saved layers
    lo = []
    om.OracleModel(os.path.join(cfg_dir, name), sd)(xh[:2].float().cpu(), layer_outputs=lo)
    for i in sorted(on):
        a, b = on[i], off[i]
        scale = float(b.abs().max()) + 1e-6
        d = float((a - b).abs().max())
        assert d <= 4e-3 * scale, f"layer {i}: block program vs per-layer kernels differ by {d} (scale {scale})"
        w = lo[i]
        dw = float((a[:2] - w).abs().max())
        assert dw <= 3e-2 * (float(w.abs().max()) + 1e-6), f"layer {i}: block program vs oracle differ by {dw}"


def test_head_small_levels_block_vs_per_layer_and_oracle(cfg_dir):
    """The Detect towers of the 20x20 level as one block program: raw maps and pred vs the per-layer kernels and the oracle."""
    name = "yolo11n-test.yaml"
    m, sd = _build(name)
    x = synth.synth_images(4, 640, 640).cuda().half()
    m.block_fusion = m.model[-1].block_fusion = True
    p1, raw1 = m(x)
    m.block_fusion = m.model[-1].block_fusion = False
    p0, raw0 = m(x)
    torch.cuda.synchronize()
    for a, b in zip(raw1, raw0):
        assert float((a.float() - b.float()).abs().max()) <= 4e-3 * float(b.float().abs().max())
    assert float((p1[:, 4:] - p0[:, 4:]).abs().max()) < 2e-3 and float((p1[:, :4] - p0[:, :4]).abs().max()) < 0.5
    want, _ = om.OracleModel(os.path.join(cfg_dir, name), sd)(x[:2].float().cpu())
    assert float((p1[:2, 4:].cpu() - want[:, 4:]).abs().max()) < 1.5e-2
    assert float((p1[:2, :4].cpu() - want[:, :4]).abs().max()) < 4e-3 * 640


def test_block_program_survives_weight_reload_and_graph_capture():
    """load_state_dict drops recorded programs (they hold packed weights); replay inside a captured hipGraph equals eager."""
    from edge_yolo_amd.engine.predictor import GraphRunner
    m, sd = _build("yolo11n-test.yaml")
    m.block_fusion = m.model[-1].block_fusion = True
    x = synth.synth_images(2, 640, 640).cuda().half()
    y0 = m(x)[0].clone()
    sd2 = synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=3)
    m.load_state_dict({k: v.half() if v.is_floating_point() else v for k, v in sd2.items()})
    y1 = m(x)[0].clone()
    m.block_fusion = m.model[-1].block_fusion = False
    y1_ref = m(x)[0].clone()
    m.block_fusion = m.model[-1].block_fusion = True
    assert float((y1 - y0).abs().max()) > 1e-3  # different weights, different result: the programs were re-recorded
    assert float((y1[:, 4:] - y1_ref[:, 4:]).abs().max()) < 2e-3
    g = GraphRunner(lambda im: (m(im)[0],))
    a = g(x)[0].clone()
    b = g(x)[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, y1)


@pytest.mark.parametrize("H,W,B", [(20, 20, 3), (13, 17, 2), (10, 10, 5)])
def test_block_stage_ops_vs_standalone_kernels(H, W, B):
    """Each block-kernel operator against the stand-alone kernel on the same inputs (odd sizes, channel tails, groups, residuals)."""
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd import _lib as L
    from edge_yolo_amd.nn import _block, _ops
    from edge_yolo_amd.nn import modules as M
    torch.manual_seed(0)
    dev = "cuda"

    def mk(c, h=H, w=W):
        return (torch.randn(B, h, w, c, device=dev) * 0.5).half().permute(0, 3, 1, 2)

    def filled(mod, tag):
        mod.load_state_dict({k: synth.synth_tensor(f"{tag}.{k}", tuple(v.shape)) for k, v in mod.state_dict().items()})
        for mm in mod.modules():
            if isinstance(mm, torch.nn.BatchNorm2d):
                mm.eps = 1e-3
        return mod.to(dev).half().eval()

    cases = []
    c1 = filled(M.Conv(48, 80, 1, 1), "b1")          # Cin tail (48 = 32 + 16), Cout 80 (NT 5)
    c3 = filled(M.Conv(64, 128, 3, 2), "b2")         # 3x3 stride 2
    c3s1 = filled(M.Conv(32, 64, 3, 1), "b3")
    ds5 = filled(M.DSConv(64, 64, 5), "b4")
    dw3 = filled(M.DWConv(80, 80, 3), "b5")
    sp = filled(M.SPPF(128, 128, 5), "b6")
    psa = filled(M.PSABlock_LinearAttention(128, num_heads=2), "b7")
    wav = filled(M.DSC3K2_Wavelet(64, 64, 1, True), "b8") if H % 2 == 0 and W % 2 == 0 else None
    x48, x64, x32, x80, x128 = mk(48), mk(64), mk(32), mk(80), mk(128)
    cases = [("conv1x1", lambda a: [c1(a)], [x48]), ("conv3x3s2", lambda a: [c3(a)], [x64]), ("conv3x3s1+res", lambda a: [c3s1(a, res=None)], [x32]),
             ("dsconv5+res", lambda a: [ds5(a, res=a)], [x64]), ("dwconv3", lambda a: [dw3(a)], [x80]), ("sppf", lambda a: [sp(a)], [x128]),
             ("psa_linattn", lambda a: [psa(a)], [x128])]
    if wav is not None:
        cases.append(("dsc3k2_wavelet", lambda a: [wav(a)], [x64]))
    for tag, fn, ins in cases:
        want = [t.clone() for t in fn(*ins)]
        cache = _block.BlockCache(tag)
        got = cache.run(fn, ins)
        assert got is not None, f"{tag}: not block-executable"
        again = cache.run(fn, ins)  # replay of the recorded program into fresh outputs
        torch.cuda.synchronize()
        for g, g2, w in zip(got, again, want):
            scale = float(w.float().abs().max()) + 1e-6
            assert float((g.float() - w.float()).abs().max()) <= 4e-3 * scale, tag
            assert torch.equal(g, g2), tag


def test_c2psa_pointwise_chains_at_real_shape(cfg_dir):
    """C2PSA_LinearAttention at its real shape (256 channels, 20x20): [cv1 -> qkv] and [proj -> ffn -> ffn -> cv2] as ONE launch each
    (tiled block programs: one workgroup per 16-pixel tile): 3 launches instead of 7, same result as one launch per conv (same f16
    rounding points) and the oracle within the f16 tolerance."""
    import edge_yolo_amd  # noqa: F401
    from edge_yolo_amd import profiling
    from edge_yolo_amd.nn import modules as M
    from gpu_util import load_synth, to_dev, check
    m = M.C2PSA_LinearAttention(256, 256, 1)
    sd = load_synth(m, "c2psa")
    x = (synth.synth_images(4, 20, 20, c=256) - 0.5)
    mh = to_dev(m, torch.float16)
    mh.pw_chains = True
    xd = x.cuda().half().contiguous(memory_format=torch.channels_last)
    with profiling.trace() as t:
        got = mh(xd)
    names = [r[0] for r in t.records]
    assert len(names) == 3 and names[0].startswith("block_tile_kernel") and names[2].startswith("block_tile_kernel"), names
    again = mh(xd)  # replay of the recorded programs
    mh.pw_chains = False
    with profiling.trace() as t2:
        plain = mh(xd)
    torch.cuda.synchronize()
    assert len(t2.records) == 7
    assert torch.equal(got, again)
    scale = float(plain.float().abs().max())
    assert float((got.float() - plain.float()).abs().max()) <= 3e-3 * scale
    check(got, om.c2psa(sd, "c2psa", x, 1, True), torch.float16, scale=max(1.0, scale))
    # weights reloaded -> the recorded programs (they hold packed weights) are dropped and re-recorded
    mh.pw_chains = True
    sd2 = {k: synth.synth_tensor("other." + k, tuple(v.shape)) for k, v in m.state_dict().items()}
    mh.load_state_dict({k: v.half() if v.is_floating_point() else v for k, v in sd2.items()})
    got2 = mh(xd)
    mh.pw_chains = False
    plain2 = mh(xd)
    assert float((got2.float() - plain2.float()).abs().max()) <= 3e-3 * float(plain2.float().abs().max()) and not torch.equal(got2, got)
