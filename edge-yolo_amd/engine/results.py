"""Minimal `Results` / `Boxes` containers with the attribute surface of the reference's engine/results.py
(:187 Results, :938 Boxes) that predict() callers read.  Plotting/saving are out of scope."""
import torch


class Boxes:
    """(n,6) rows [x1,y1,x2,y2,conf,cls] in original-image pixels."""

    def __init__(self, boxes, orig_shape):
        if boxes.ndim == 1:
            boxes = boxes[None, :]
        assert boxes.shape[-1] == 6, f"expected 6 values but got {boxes.shape[-1]}"
        self.data = boxes
        self.orig_shape = orig_shape

    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, -2]

    @property
    def cls(self):
        return self.data[:, -1]

    @property
    def xywh(self):
        b = self.xyxy
        return torch.cat(((b[:, :2] + b[:, 2:]) / 2, b[:, 2:] - b[:, :2]), 1)

    @property
    def xyxyn(self):
        b = self.xyxy.clone()
        b[:, [0, 2]] /= self.orig_shape[1]
        b[:, [1, 3]] /= self.orig_shape[0]
        return b

    def cpu(self):
        return Boxes(self.data.cpu(), self.orig_shape)

    def numpy(self):
        return Boxes(self.data.cpu().numpy(), self.orig_shape)

    def __len__(self):
        return len(self.data)


class Results:
    def __init__(self, orig_img, path, names, boxes=None, speed=None):
        self.orig_img = orig_img
        self.orig_shape = tuple(orig_img.shape[:2]) if orig_img is not None and hasattr(orig_img, "shape") and orig_img.ndim == 3 else None
        self.boxes = Boxes(boxes, self.orig_shape) if boxes is not None else None
        self.names = names
        self.path = path
        self.speed = speed or {"preprocess": None, "inference": None, "postprocess": None}

    def __len__(self):
        return len(self.boxes) if self.boxes is not None else 0
