"""Sparse building blocks of PaSCo's U-Net, restated on fused HIP launches.

Module trees (and therefore state-dict keys) follow the reference so its checkpoints load:
  ResidualBlock                      pasco/maskpls/mink.py:618-658
  BasicConvolutionBlock              pasco/maskpls/mink.py:505-518
  BasicGenerativeDeconvolutionBlock  pasco/maskpls/mink.py:520-534
  MinkowskiSpatialDropout            pasco/models/dropout.py:43-60 (identity at inference)
Each block's eval forward is a fixed sequence of conv launches with BN/activation folded in.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import me as ME
from . import fused
from .fused import ACT_LEAKY, ACT_NONE, ACT_RELU


class SpatialDropout(ME.MinkowskiModuleBase):
    """Channel dropout on sparse features; identity in eval mode (the only mode served)."""

    def __init__(self, p: float = 0.5, *a, **kw):
        super().__init__()
        self.p = p

    def forward(self, x):
        assert not self.training, "inference only"
        return x


class ResidualBlock(nn.Module):
    """Pre-activation block: BN-ReLU-conv3-BN-ReLU-conv3, out = ReLU(skip + y); no BN on the skip.

    Fused schedule (2 launches): conv_a gathers ReLU(BN0(x)) (prologue) and stores ReLU(BN1(.))
    (epilogue); conv_b adds the skip and applies the final ReLU in its epilogue."""

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1, D=3, drop_path=0.0, use_se=False):
        super().__init__()
        assert drop_path == 0.0 and not use_se, "drop-path / SE are not on the served path"
        self.net = nn.Sequential(
            ME.MinkowskiBatchNorm(inc),
            ME.MinkowskiReLU(inplace=True),
            ME.MinkowskiConvolution(inc, outc, kernel_size=ks, dilation=dilation, stride=stride, dimension=D),
            ME.MinkowskiBatchNorm(outc),
            ME.MinkowskiReLU(inplace=True),
            ME.MinkowskiConvolution(outc, outc, kernel_size=ks, dilation=dilation, stride=1, dimension=D),
        )
        self.downsample = nn.Sequential() if (inc == outc and stride == 1) else nn.Sequential(
            ME.MinkowskiConvolution(inc, outc, kernel_size=1, dilation=1, stride=stride, dimension=D))
        self.relu = ME.MinkowskiReLU(inplace=True)

    def next_prologue(self):
        """What this block's first convolution applies to its input: lets the producer of that input emit the
        pre-split operand in its own epilogue (fused.conv emit_next)."""
        return (self.net[0], ACT_RELU)

    def forward(self, x: ME.SparseTensor, emit_next=None) -> ME.SparseTensor:
        skip = x if len(self.downsample) == 0 else fused.conv(x, self.downsample[0])
        y = fused.conv(x, self.net[2], pro_bn=self.net[0], pro_act=ACT_RELU, epi_bn=self.net[3], epi_act=ACT_RELU,
                       emit_next=(None, ACT_NONE), split_only=True)   # y has one reader: never stored as fp32
        return fused.conv(y, self.net[5], residual=skip.F, res_act=ACT_RELU, emit_next=emit_next)


class BasicConvolutionBlock(nn.Module):
    """conv -> BN -> LeakyReLU(0.01); `post_bn` (optional) is the extra BN+ReLU the encoder appends
    (encoder_v2.py:124-126), folded into the same epilogue."""

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1, D=3):
        super().__init__()
        self.net = nn.Sequential(
            ME.MinkowskiConvolution(inc, outc, kernel_size=ks, dilation=dilation, stride=stride, dimension=D),
            ME.MinkowskiBatchNorm(outc),
            ME.MinkowskiLeakyReLU(inplace=True),
        )

    def forward(self, x, post_bn=None, post_act=ACT_NONE, emit_next=None):
        return fused.conv(x, self.net[0], epi_bn=self.net[1], epi_act=ACT_LEAKY,
                          slope=self.net[2].module.negative_slope, epi2_bn=post_bn, res_act=post_act,
                          emit_next=emit_next)


class BasicGenerativeDeconvolutionBlock(nn.Module):
    """generative transposed conv (k2, s2, expand_coordinates) -> BN -> LeakyReLU."""

    def __init__(self, inc, outc, ks=3, stride=1, D=3):
        super().__init__()
        self.net = nn.Sequential(
            ME.MinkowskiConvolutionTranspose(inc, outc, kernel_size=ks, stride=stride, dimension=D,
                                             expand_coordinates=True),
            ME.MinkowskiBatchNorm(outc),
            ME.MinkowskiLeakyReLU(inplace=True),
        )

    def forward(self, x, out_key=None, nbr=None, emit_next=None, split_only=False):
        return fused.conv(x, self.net[0], epi_bn=self.net[1], epi_act=ACT_LEAKY,
                          slope=self.net[2].module.negative_slope, out_key=out_key, nbr=nbr, emit_next=emit_next,
                          split_only=split_only, one_pair=True)     # every child has exactly its one parent


def first_prologue(seq: nn.Sequential):
    """(bn, act) the first convolution of `seq` applies to its input, or None when unknown: a ResidualBlock reads
    ReLU(BN0(x)), a BasicConvolutionBlock reads x as it is."""
    for m in seq:
        if isinstance(m, (nn.Identity, SpatialDropout)):
            continue
        if isinstance(m, ResidualBlock):
            return m.next_prologue()
        if isinstance(m, BasicConvolutionBlock):
            return (None, ACT_NONE)
        return None
    return None


def run_sequential(seq: nn.Sequential, x, emit_last=None):
    """Run a reference-shaped nn.Sequential, folding `BasicConvolutionBlock, BN, ReLU` triples and
    skipping identities / dropouts.  `emit_last` = prologue of whatever convolution reads the result
    (fused.conv emit_next), when the caller knows it."""
    mods = [m for m in seq if not isinstance(m, (nn.Identity, SpatialDropout))]

    def emit_for(j):
        """prologue of the module that will read the output of the step ending before index j"""
        if j >= len(mods):
            return emit_last
        return mods[j].next_prologue() if isinstance(mods[j], ResidualBlock) else None

    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, BasicConvolutionBlock) and i + 2 < len(mods) and \
                isinstance(mods[i + 1], ME.MinkowskiBatchNorm) and isinstance(mods[i + 2], ME.MinkowskiReLU):
            x = m(x, post_bn=mods[i + 1], post_act=ACT_RELU, emit_next=emit_for(i + 3))
            i += 3
            continue
        if isinstance(m, (ResidualBlock, BasicConvolutionBlock)):
            x = m(x, emit_next=emit_for(i + 1))
        else:
            x = m(x)
        i += 1
    return x
