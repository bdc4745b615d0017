"""Helpers shared by the -m gpu parity tests: same name-keyed synthetic weights go into the HIP modules (through
their reference-compatible state_dict keys) and into the CPU oracle."""
import torch

import synthdata as synth

# fp32 is the parity gate (north star: boxes/scores within 1e-3 in fp32).  fp16 is the throughput mode: storage is
# half precision (11-bit mantissa), so activations carry ~1e-3 relative error per layer; tolerance documented here.
TOL = {torch.float32: dict(rtol=2e-4, atol=2e-4), torch.float16: dict(rtol=3e-2, atol=3e-2)}


def load_synth(module, prefix):
    sd = {k: synth.synth_tensor(f"{prefix}.{k}", tuple(v.shape)) for k, v in module.state_dict().items()}
    module.load_state_dict(sd)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps = 1e-3
    return {f"{prefix}.{k}": v for k, v in sd.items()}


def to_dev(module, dtype):
    module = module.to("cuda").eval()
    return module.half() if dtype == torch.float16 else module.float()


def check(got, want, dtype, scale=1.0, what=""):
    got = got.detach().float().cpu()
    t = TOL[dtype]
    torch.testing.assert_close(got, want, rtol=t["rtol"], atol=t["atol"] * scale, msg=lambda m: f"{what}: {m}")
