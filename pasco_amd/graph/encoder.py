"""Sparse encoder of PaSCo's U-Net (reference: pasco/models/encoder_v2.py:89-183).

conv k1 (64*M -> 64) then three down-sampling stages (conv k2 s2 + BN + LeakyReLU, BN, ReLU,
followed by 3 residual blocks in the light variant or a dropout in the heavy one).  Returns the
four feature maps at tensor strides 1, 2, 4, 8.
"""
from __future__ import annotations

import torch.nn as nn

from .. import me as ME
from . import fused
from .blocks import BasicConvolutionBlock, ResidualBlock, SpatialDropout, first_prologue, run_sequential


class Encoder3DSepV2(nn.Module):
    def __init__(self, in_channels, f, heavy_decoder=True, dropouts=(0.0, 0.0, 0.0)):
        super().__init__()
        self.enc_in_feats = ME.MinkowskiConvolution(in_channels, f[0], kernel_size=1, stride=1, dimension=3)

        def stage(cin, cout, p):
            head = [BasicConvolutionBlock(cin, cout, ks=2, stride=2), ME.MinkowskiBatchNorm(cout), ME.MinkowskiReLU()]
            if heavy_decoder:
                return nn.Sequential(*head, SpatialDropout(p=p))
            return nn.Sequential(*head, ResidualBlock(cout, cout), ResidualBlock(cout, cout), ResidualBlock(cout, cout))

        if heavy_decoder:
            self.s1 = nn.Sequential(nn.Identity())
        else:
            self.s1 = nn.Sequential(ResidualBlock(f[0], f[0]), ResidualBlock(f[0], f[0]), ResidualBlock(f[0], f[0]),
                                    nn.Identity())
        self.s1s2 = stage(f[0], f[1], dropouts[-3])
        self.s2s4 = stage(f[1], f[2], dropouts[-2])
        self.s4s8 = stage(f[2], f[3], dropouts[-1])

    def forward(self, x: ME.SparseTensor):
        assert not self.training, "inference only"
        # the three strided coordinate maps depend on the input coordinates only: built together, one host read for their sizes
        x.coordinate_manager.stride_chain(x.coordinate_map_key, 3)
        # each stage's last launch also writes the next stage's first operand (fused.conv emit_next)
        s1 = run_sequential(self.s1, fused.conv(x, self.enc_in_feats, emit_next=first_prologue(self.s1)),
                            emit_last=first_prologue(self.s1s2))
        s2 = run_sequential(self.s1s2, s1, emit_last=first_prologue(self.s2s4))
        s4 = run_sequential(self.s2s4, s2, emit_last=first_prologue(self.s4s8))
        s8 = run_sequential(self.s4s8, s4)
        return [s1, s2, s4, s8]
