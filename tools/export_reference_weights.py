#!/usr/bin/env python
"""Weight bridge, REFERENCE side: turn a checkpoint written by the reference trainer (a pickled module object,
engine/trainer.py:513-546; read back by nn/tasks.py:815-955) into the tensor-only format `edge_yolo_amd.YOLO(path)` loads with
`torch.load(weights_only=True)`:  {"yaml": <model yaml dict>, "nc": int, "fused": bool, "state_dict": {reference keys: fp32 tensors}}.

Run it where the reference package (`ultralytics`, this fork) is importable -- unpickling a reference checkpoint executes its module
code, which is why edge-yolo_amd never does it:

    python tools/export_reference_weights.py runs/detect/train/weights/best.pt edgeline_gc10.pt
    >>> from edge_yolo_amd import YOLO; YOLO("edgeline_gc10.pt").predict(batch, half=True)
"""
import sys

import torch


def export(src, dst):
    ck = torch.load(src, map_location="cpu", weights_only=False)  # a reference checkpoint: pickled nn.Module (trusted input of the reference's user)
    model = ck.get("ema") or ck["model"] if isinstance(ck, dict) else ck
    model = model.float().eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = dict(model.yaml)  # the parsed YAML incl. 'scale', 'nc', 'ch', 'yaml_file' (tasks.py:331-337)
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}
    first_conv = next(k for k in sd if k.endswith("conv.weight"))
    fused = first_conv.replace("conv.weight", "bn.weight") not in sd  # BaseModel.fuse() deleted the BatchNorms (tasks.py:214-242)
    torch.save({"yaml": cfg, "nc": int(cfg["nc"]), "fused": bool(fused), "state_dict": sd}, dst)
    return len(sd), fused


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    n, fused = export(sys.argv[1], sys.argv[2])
    print(f"wrote {sys.argv[2]}: {n} tensors, fused={fused}")
