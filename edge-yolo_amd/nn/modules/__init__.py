"""Module registry of the detection path (names resolve from YAML `module` strings, like the reference's
`globals()[m]` lookup in nn/tasks.py:984)."""
from .block import (DFL, SPPF, C2f, C3, C3k, C3k2, Bottleneck, Attention, PSABlock, C2PSA, LinearAttention, PSABlock_LinearAttention,
                    C2PSA_LinearAttention, DSBottleneck, DSC3k, DSC3K2_Wavelet)
from .conv import Conv, DWConv, DSConv, Concat, Upsample, autopad
from .head import Detect, E2EDetect, GF2Detect, GFLHeadv2_uniH

__all__ = ("Conv", "DWConv", "DSConv", "Concat", "Upsample", "autopad", "DFL", "SPPF", "C2f", "C3", "C3k", "C3k2", "Bottleneck", "Attention",
           "PSABlock", "C2PSA", "LinearAttention", "PSABlock_LinearAttention", "C2PSA_LinearAttention", "DSBottleneck", "DSC3k",
           "DSC3K2_Wavelet", "Detect", "GF2Detect", "E2EDetect", "GFLHeadv2_uniH")
