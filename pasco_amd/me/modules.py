"""MinkowskiEngine-named nn.Modules served by the HIP library (SURVEY.md 8(b) census).

Parameter names/shapes follow upstream so reference checkpoints load: convolution `kernel`
[kvol, cin, cout] ([cin, cout] when kvol == 1) and `bias` [1, cout]; batch norm under `.bn.*`.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import backend as B
from .core import CoordinateManager, CoordinateMapKey, SparseTensor, TensorField, _triple


class MinkowskiModuleBase(nn.Module):
    pass


def _kvol(kernel_size) -> int:
    k = _triple(kernel_size)
    return k[0] * k[1] * k[2]


_ME_CONV = os.environ.get("PASCO_ME_CONV", "guarded")
_ME_DEFER = os.environ.get("PASCO_ME_DEFER", "1") != "0"      # eval-mode BatchNorm / ReLU recorded, applied by the next convolution
ME_MIN_ROWS_WINDOWS = 16384     # 3x3x3 maps with at least this many rows get window tables (as pasco_amd.graph.fused does)


def set_me_conv(mode: str) -> None:
    """"guarded" (default): the plain convolution modules run on the split-precision kernels with the exact fp32 kernel as a
    device-side fallback; "exact": the exact fp32 kernel only."""
    global _ME_CONV
    assert mode in ("guarded", "exact")
    _ME_CONV = mode


class _ConvBase(MinkowskiModuleBase):
    is_transpose = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__()
        assert dimension == 3, "dimension=3 is the served case"
        assert kernel_generator is None, "custom kernel generators are not served"
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.dilation = _triple(dilation)
        self.expand_coordinates = expand_coordinates
        self.kernel_volume = _kvol(kernel_size)
        self.dimension = dimension
        self.use_mm = self.kernel_volume == 1 and self.stride == (1, 1, 1)
        shape = (in_channels, out_channels) if self.kernel_volume == 1 else (self.kernel_volume, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.empty((1, out_channels), dtype=torch.float32)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            n = (self.out_channels if self.is_transpose else self.in_channels) * self.kernel_volume
            stdv = 1.0 / math.sqrt(n)
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    # out key + neighbour table for an input tensor
    def _maps(self, x: SparseTensor):
        mgr = x.coordinate_manager
        in_key = x.coordinate_map_key
        if self.use_mm:
            return in_key, None
        if self.is_transpose:
            assert self.expand_coordinates, "transposed convolution is served with expand_coordinates=True"
            assert self.kernel_size == (2, 2, 2) and self.stride == (2, 2, 2), \
                "generative transposed convolution is served for kernel 2 / stride 2"
            out_key = mgr.expand(in_key, self.stride)
            nbr = mgr.kernel_map(in_key, out_key, self.kernel_size, self.dilation, transposed=True)
        else:
            if self.stride != (1, 1, 1):
                assert self.kernel_size == self.stride, "strided convolution is served for kernel == stride"
            out_key = mgr.stride(in_key, self.stride)
            nbr = mgr.kernel_map(in_key, out_key, self.kernel_size, self.dilation, transposed=False)
        return out_key, nbr

    def forward(self, x: SparseTensor, coordinates=None) -> SparseTensor:
        assert isinstance(x, SparseTensor)
        assert coordinates is None, "explicit output coordinates are not served"
        out_key, nbr = self._maps(x)
        mgr = x.coordinate_manager
        raw, pending = x.take_prologue()
        out = self.conv_rows(mgr.backend(), raw, nbr, mgr.size(out_key), mgr=mgr, prologue=pending)
        return SparseTensor(out, coordinate_map_key=out_key, coordinate_manager=mgr)

    def conv_rows(self, be, feats: torch.Tensor, nbr, n_out: int, mgr=None, prologue=None) -> torch.Tensor:
        """The module's own launch(es): exact fp32 results, as upstream's.  Where the split-precision kernels apply (GPU,
        cin % 8 == 0) the products run on them GUARDED: the operand split reports an f16 range overflow (|x| > 2047) into a
        word of this call, and the exact fp32 kernel is launched behind the split one with that word as its predicate
        (`ph_conv_desc.exact_if`): it replaces the result exactly when it has to and costs an empty launch otherwise.  No
        host read, nothing for the caller to check - a maintainer who only swaps the import gets the fast kernels (round 5:
        the exact fp32 MFMA runs at 1 / 16 of the f16 rate and was ~85 % of the unfused route's time).
        `PASCO_ME_CONV=exact` (or `set_me_conv("exact")`): the exact kernel only.
        `prologue` = (scale, shift, act, slope) of a deferred BatchNorm / activation in front (SparseTensor.deferred): applied
        to the gathered rows by the launch itself instead of by separate passes over the tensor."""
        feats = feats.contiguous()
        kernel = self.kernel.detach().contiguous()
        bias = self.bias.detach().reshape(-1).contiguous() if self.bias is not None else None
        pro = {}
        if prologue is not None:
            ps, pb, pact, pslope = prologue
            pro = dict(pro_scale=ps, pro_shift=pb, pro_act=pact, slope=pslope)
        if n_out == 0 or _ME_CONV != "guarded" or not be.split_supported(self.in_channels, self.out_channels) or \
                torch.is_grad_enabled() and self.kernel.requires_grad and self.training:
            return be.conv_fwd(feats, kernel, nbr, n_out, bias=bias, **pro)
        w = self.kernel
        ver = (w._version, w.data_ptr(), w.device)
        hit = self.__dict__.get("_ph_me_split")
        if hit is None or hit[0] != ver:
            hit = (ver, be.split_weight_rows(kernel))
            if w.is_cuda and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(w.device).synchronize()     # the cache serves every stream (fused.publish)
            self.__dict__["_ph_me_split"] = hit
        flag = torch.zeros(1, dtype=torch.int32, device=feats.device)
        xs = be.split_rows(feats, status=flag, **pro)
        win = None
        if mgr is not None and nbr is not None and nbr.shape[0] == 27 and 33 <= self.out_channels <= 64 and \
                n_out >= ME_MIN_ROWS_WINDOWS and be.device_type == "cuda":
            win = mgr.kernel_windows(nbr)       # LDS-window tables of the map (cached by the manager): k_conv_wop2
        out = be.conv_fwd(feats, kernel, nbr, n_out, bias=bias, split=hit[1], in_split=xs, status=flag, win=win)
        be.conv_fwd(feats, kernel, nbr, n_out, bias=bias, out=out, exact_if=flag, **pro)
        return out

    def extra_repr(self):
        return (f"in={self.in_channels}, out={self.out_channels}, kernel_size={list(self.kernel_size)}, "
                f"stride={list(self.stride)}, dilation={list(self.dilation)}")


class MinkowskiConvolution(_ConvBase):
    """reference: mink.py:509-511,625-638; encoder_v2.py:109-111; decoder_v3.py:103-105,133-135,267-282"""
    is_transpose = False


class MinkowskiConvolutionTranspose(_ConvBase):
    """reference: mink.py:524-527 (kernel 2, stride 2, expand_coordinates=True)"""
    is_transpose = True

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator,
                         expand_coordinates, convolution_mode, dimension)


class MinkowskiGenerativeConvolutionTranspose(MinkowskiConvolutionTranspose):
    def __init__(self, *a, **kw):
        kw["expand_coordinates"] = True
        super().__init__(*a, **kw)


# ---- normalisation / activations: wrappers over torch.nn on F, same key -------------
def _same_map(x: SparseTensor, feats: torch.Tensor) -> SparseTensor:
    return SparseTensor(feats, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)


_GRAD_MODE_WARNED = False


def _warn_grad_mode_once() -> None:
    """An eval-mode module called with autograd ENABLED takes the module-by-module route (separate BatchNorm / activation passes,
    exact fp32 products where a parameter requires grad): the results are the same, the speed is not ('a caller who
    forgets torch.no_grad() silently gets the slow route').  Said once per process."""
    global _GRAD_MODE_WARNED
    if not _GRAD_MODE_WARNED:
        _GRAD_MODE_WARNED = True
        import warnings
        warnings.warn("pasco_amd.me: eval-mode modules called with autograd enabled - the fused inference route (deferred BatchNorm / "
                      "activations, split-precision convolutions) needs torch.no_grad() / torch.inference_mode(); running the "
                      "module-by-module route", RuntimeWarning, stacklevel=3)


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def folded(self):
        """(scale, shift) of the eval-mode normalisation, y = x * scale + shift; cached per parameter version."""
        m = self.bn
        bufs, pars = m._buffers, m._parameters
        rm, rv, w, b = bufs["running_mean"], bufs["running_var"], pars.get("weight"), pars.get("bias")
        ver = (rm._version, rv._version, w._version if w is not None else -1, b._version if b is not None else -1, rm.device,
               rm.data_ptr())
        hit = self.__dict__.get("_ph_me_folded")
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                scale = torch.rsqrt(rv.float() + m.eps) * (w.float() if w is not None else 1.0)
                shift = (b.float() if b is not None else 0.0) - rm.float() * scale
                scale, shift = scale.contiguous(), shift.contiguous()
            if rm.is_cuda and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(rm.device).synchronize()       # the cache serves every stream
            hit = (ver, scale, shift)
            self.__dict__["_ph_me_folded"] = hit
        return hit[1], hit[2]

    def forward(self, x: SparseTensor) -> SparseTensor:
        m = self.bn
        if _ME_DEFER and not m.training and torch.is_grad_enabled():
            _warn_grad_mode_once()
        if _ME_DEFER and not m.training and m.track_running_stats and m._buffers.get("running_mean") is not None \
                and not torch.is_grad_enabled() and x._F.dtype == torch.float32:
            if x._pending is not None:
                x.F                        # an activation is pending: it has to be applied before this affine
            scale, shift = self.folded()
            return SparseTensor.deferred(x, scale, shift, B.ACT_NONE, 0.01)
        return _same_map(x, self.bn(x.F))


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    """Inference engine: statistics are frozen, so the synchronised variant is plain BatchNorm1d with
    the same state-dict keys (reference: unet3d_sparse_v2.py:172-175)."""

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        out = module
        if isinstance(module, MinkowskiBatchNorm) and not isinstance(module, MinkowskiSyncBatchNorm):
            out = cls(module.bn.num_features, module.bn.eps, module.bn.momentum, module.bn.affine,
                      module.bn.track_running_stats)
            out.bn.load_state_dict(module.bn.state_dict())
            out.train(module.training)
        for name, child in module.named_children():
            out.add_module(name, cls.convert_sync_batchnorm(child, process_group))
        return out


class _Elementwise(MinkowskiModuleBase):
    MODULE = None

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.module = self.MODULE(*args, **kwargs)

    def forward(self, x: SparseTensor) -> SparseTensor:
        return _same_map(x, self.module(x.F))


class _DeferredAct(_Elementwise):
    """ReLU / LeakyReLU: in inference the activation is recorded on the returned tensor (SparseTensor.deferred) and applied
    by the next convolution's operand prologue, or the first time anything else reads the values."""
    ACT = B.ACT_NONE

    def forward(self, x: SparseTensor) -> SparseTensor:
        if _ME_DEFER and not torch.is_grad_enabled() and x._F.dtype == torch.float32:
            slope = float(getattr(self.module, "negative_slope", 0.01))
            if x._pending is not None and x._pending[2] == B.ACT_NONE:      # behind a deferred BatchNorm: one prologue
                raw, (scale, shift, _, _) = x.take_prologue()
                src = SparseTensor(raw, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
                return SparseTensor.deferred(src, scale, shift, self.ACT, slope)
            x.F                                                             # (materialises whatever else is pending)
            return SparseTensor.deferred(x, None, None, self.ACT, slope)
        return _same_map(x, self.module(x.F))


class MinkowskiReLU(_DeferredAct):
    MODULE = nn.ReLU
    ACT = B.ACT_RELU


class MinkowskiLeakyReLU(_DeferredAct):
    MODULE = nn.LeakyReLU
    ACT = B.ACT_LEAKY


class MinkowskiSigmoid(_Elementwise):
    MODULE = nn.Sigmoid


class MinkowskiSoftmax(_Elementwise):
    MODULE = nn.Softmax


class MinkowskiELU(_Elementwise):
    MODULE = nn.ELU


class MinkowskiDropout(_Elementwise):
    MODULE = nn.Dropout


class MinkowskiLinear(MinkowskiModuleBase):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, x: SparseTensor) -> SparseTensor:
        return _same_map(x, self.linear(x.F))


# ---- pruning / pooling -------------------------------------------------------------------------
class MinkowskiPruning(MinkowskiModuleBase):
    """reference: decoder_v3.py:127,159,285,421-432,496-497; misc.py:17,26"""

    def forward(self, x: SparseTensor, mask: torch.Tensor) -> SparseTensor:
        assert isinstance(x, SparseTensor)
        assert mask.dim() == 1 and mask.shape[0] == x.F.shape[0], "mask must be [N]"
        if mask.dtype != torch.bool:
            mask = mask != 0
        mgr = x.coordinate_manager
        out_key, keep = mgr.prune(x.coordinate_map_key, mask.to(x.device))
        feats = mgr.backend().gather_rows(x.F.contiguous(), keep)
        return SparseTensor(feats, coordinate_map_key=out_key, coordinate_manager=mgr)


class MinkowskiMaxPooling(MinkowskiModuleBase):
    """reference: transformer_predictor_v2.py:100-102,234-236 (kernel == stride)"""

    def __init__(self, kernel_size, stride=1, dilation=1, kernel_generator=None, dimension=None):
        super().__init__()
        assert dimension == 3 and kernel_generator is None
        self.kernel_size = _triple(kernel_size)
        self.stride = _triple(stride)
        self.dilation = _triple(dilation)
        assert self.kernel_size[0] ** 3 <= B.MAX_KVOL, "pooling window too large"

    def forward(self, x: SparseTensor) -> SparseTensor:
        mgr = x.coordinate_manager
        out_key = mgr.stride(x.coordinate_map_key, self.stride)
        nbr = mgr.kernel_map(x.coordinate_map_key, out_key, self.kernel_size, self.dilation)
        out = mgr.backend().maxpool_fwd(x.F.contiguous(), nbr)
        return SparseTensor(out, coordinate_map_key=out_key, coordinate_manager=mgr)


# ---- names that only need to exist so that dead reference classes still define -------------------
class _NotServed(MinkowskiModuleBase):
    def __init__(self, *a, **kw):
        super().__init__()

    def forward(self, *a, **kw):
        raise NotImplementedError(f"{type(self).__name__} is outside the served hot path (SURVEY.md 8(b))")


class MinkowskiGlobalPooling(_NotServed):
    pass


class MinkowskiGlobalAvgPooling(_NotServed):
    pass


class MinkowskiBroadcastMultiplication(_NotServed):
    pass


class MinkowskiChannelwiseConvolution(_NotServed):
    pass


class MinkowskiPoolingTranspose(_NotServed):
    pass


class MinkowskiAvgPooling(_NotServed):
    pass


def cat(*tensors):
    """Channel concat of sparse tensors sharing one coordinate map."""
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    k0 = tensors[0].coordinate_map_key
    assert all(t.coordinate_map_key == k0 for t in tensors), "cat needs one shared coordinate map"
    return _same_map(tensors[0], torch.cat([t.F for t in tensors], dim=1))
