"""Dense bottleneck (reference: pasco/models/layers.py:646-726 `SPCDense3Dv2`, wired at
pasco/models/unet3d_sparse_v2.py:182-214).

Eleven dense 3-D convolutions (256 -> 256, kernels (3,3,1) (5,5,3) (7,7,5) and 1x1x1) on the
stride-8 grid (32 x 32 x 4 for a 256 x 256 x 32 scene), each followed by BatchNorm3d + ReLU.  This is
genuinely dense GEMM work; round 1 runs it through torch.nn.Conv3d (MIOpen) - see DESIGN.md.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv(c, k):
    pad = tuple(v // 2 for v in k)
    return nn.Sequential(nn.Conv3d(c, c, k, 1, padding=pad, bias=False), nn.Identity())


class SPCDense3Dv2(nn.Module):
    def __init__(self, init_size=16):
        super().__init__()
        c = init_size
        specs = {"a_conv1": (3, 3, 1), "a_conv2": (3, 3, 1), "a_conv3": (5, 5, 3), "a_conv4": (7, 7, 5),
                 "a_conv5": (3, 3, 1), "a_conv6": (5, 5, 3), "a_conv7": (7, 7, 5)}
        for i, (name, k) in enumerate(specs.items(), start=1):
            setattr(self, name, _conv(c, k))
            setattr(self, f"bn_{i}", nn.BatchNorm3d(c))
        self.ch_conv1 = nn.Sequential(nn.Conv3d(c, c, kernel_size=1, stride=1, bias=False), nn.Identity())
        self.bn_ch_conv1 = nn.BatchNorm3d(c)
        for i, k in enumerate(((3, 3, 1), (5, 5, 3), (7, 7, 5)), start=1):
            setattr(self, f"res_{i}", _conv(c, k))
            setattr(self, f"bn_res_{i}", nn.BatchNorm3d(c))

    def _cbr(self, conv, bn, x):
        return F.relu(bn(conv(x)))

    def forward(self, x):
        x1 = self._cbr(self.a_conv1, self.bn_1, x)
        x2 = self._cbr(self.a_conv2, self.bn_2, x1)
        x3 = self._cbr(self.a_conv3, self.bn_3, x1)
        x4 = self._cbr(self.a_conv4, self.bn_4, x1)
        t1 = x2 + x3 + x4
        x5 = self._cbr(self.a_conv5, self.bn_5, t1)
        x6 = self._cbr(self.a_conv6, self.bn_6, t1)
        x7 = self._cbr(self.a_conv7, self.bn_7, t1)
        s = x1 + x2 + x3 + x4 + x5 + x6 + x7
        y0 = self._cbr(self.ch_conv1, self.bn_ch_conv1, s)
        y1 = self._cbr(self.res_1, self.bn_res_1, x)
        y2 = self._cbr(self.res_2, self.bn_res_2, x)
        y3 = self._cbr(self.res_3, self.bn_res_3, x)
        return x1 + y0 + y1 + y2 + y3
