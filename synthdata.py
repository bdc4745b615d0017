"""Deterministic synthetic weights / inputs keyed by tensor NAME (not by RNG call order), so the real reference (in
tests/golden/make_golden.py), the CPU oracle, the HIP product and bench.py all get bit-identical parameters without shipping a
10 MB state_dict.  A neutral data generator: it imports neither the product (`edge-yolo_amd/`) nor the oracle (`oracle/`), and both
sides of every parity test draw from it."""
import zlib
import numpy as np
import torch


def _rng(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def synth_tensor(name, shape, seed=0, gain=1.9):
    """Value rule per key suffix.  Conv kernels get variance gain/fan_in so activations neither die
    nor blow up through ~100 layers; BN statistics are randomised (defaults would make BN a near
    identity) and wave.gamma is non-zero (tanh(0)=0 would silence the wavelet branch,
    nn/modules/block.py:3679,3710)."""
    r = _rng(name, seed)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if name.endswith("dfl.conv.weight"):  # fixed integral weights, block.py:83-85
        return torch.arange(shape[1], dtype=torch.float32).view(shape)
    if ".bn." in name or name.endswith(("bn.weight", "bn.bias")):
        if leaf in ("weight", "running_var"):
            a = r.uniform(0.5, 1.5, shape)
        else:
            a = r.normal(0.0, 0.1, shape)
        return torch.tensor(a, dtype=torch.float32)
    if leaf == "gamma":
        return torch.tensor(r.uniform(0.5, 1.5, shape), dtype=torch.float32)
    if leaf == "alpha":
        return torch.tensor(np.array([0.5, 0.2, 0.2, 0.1]) + r.normal(0, 0.05, shape), dtype=torch.float32)
    if leaf == "bias":
        return torch.tensor(r.normal(0.0, 0.1, shape), dtype=torch.float32)
    if leaf == "weight" and len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        b = (3.0 * gain / fan_in) ** 0.5
        return torch.tensor(r.uniform(-b, b, shape), dtype=torch.float32)
    return torch.tensor(r.normal(0.0, 0.1, shape), dtype=torch.float32)


def synth_state_dict(shapes, seed=0, gain=1.9):
    """shapes: {key: shape}.  Returns {key: tensor}."""
    return {k: synth_tensor(k, s, seed, gain) for k, s in shapes.items()}


def synth_images(b, h, w, seed=0, c=3):
    r = _rng(f"images{b}x{c}x{h}x{w}", seed)
    return torch.tensor(r.random((b, c, h, w)), dtype=torch.float32)


def synth_pred(b, nc, a, seed=2, imgsz=640, dense=False):
    """NMS micro-benchmark input: pred (B,4+nc,A): cx,cy~U(0,imgsz), w,h~LogNormal(ln 64,.6),
    per-anchor best score ~Beta(.5,6) (dense=True: Beta(2,2)), other classes x U(0,.2)."""
    r = _rng(f"pred{b}x{nc}x{a}", seed)
    p = np.empty((b, 4 + nc, a), np.float32)
    p[:, 0:2] = r.uniform(0, imgsz, (b, 2, a))
    p[:, 2:4] = r.lognormal(np.log(64.0), 0.6, (b, 2, a))
    best = r.beta(2, 2, (b, a)) if dense else r.beta(0.5, 6, (b, a))
    cls = r.uniform(0, 0.2, (b, nc, a)) * best[:, None, :]
    j = r.integers(0, nc, (b, a))
    bi, ai = np.meshgrid(np.arange(b), np.arange(a), indexing="ij")
    cls[bi, j, ai] = best
    p[:, 4:] = cls
    return torch.tensor(p)
