"""Fused launch helpers around `ph_conv_fwd`.

* eval-mode BatchNorm folded to per-channel (scale, shift) and handed to the conv kernel as gather prologue /
  store epilogue, so the sparse blocks of PaSCo need no separate elementwise passes (SURVEY.md 8(a) a6,
  section 9 items 6-7);
* the split-precision operand plumbing (include/pasco_hip.h, mma_mode 2): weights split once per module,
  activations split once per tensor (`split_input`) or written in operand form by the launch that produces them
  (`conv(..., emit_next=...)`, `SplitRows` for tensors that exist only in that form);
* tall linear layers (`linear_rows`, `linear_bn_act`, `batched_rows_matmul`) as identity-map convolutions on the
  same kernel."""
from __future__ import annotations

import contextlib
import os
import threading
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ..me import SparseTensor
from ..me.backend import ACT_LEAKY, ACT_NONE, ACT_RELU
from ..me.modules import MinkowskiBatchNorm, _ConvBase

_CONV_PRECISION = "f16x3"
_PRESPLIT = True     # f16x3 operands pre-split once per tensor (mma_mode 2) instead of per gather (mode 1)
_FUSION = True       # False: route (a) of INTEGRATION.md - every block as the reference's own module sequence
MIN_ROWS_LINEAR = 16384    # tall-operand threshold above which linear layers run on the convolution kernel
MIN_ROWS_WINDOWS = 16384   # 3x3x3 maps with at least this many rows get LDS-window tables (conv_win.hip)
_WINDOWS_WIDE = os.environ.get("PASCO_CONV_WIN", "1") == "2"   # tables for 128-wide layers too (experiments)


def _kernel_device(device) -> bool:
    """True where the convolution kernel can take linear layers: a GPU, or (tests) a CPU checker backend that
    has been allowed to take split-operand descriptors."""
    if device.type == "cuda":
        return True
    from ..me import backend
    cb = backend._checker_backend
    return cb is not None and cb.checker_split


def set_conv_precision(mode: str) -> None:
    """"f16x3" (default): convolutions whose shape allows it (cin % 8 == 0, cout % 4 == 0) form their products
    as hi*hi + hi*lo + lo*hi of f16 halves with fp32 accumulation (pasco_amd/csrc/conv_f16x3.hip): measured
    error against fp64 <= that of the fp32 MFMA path (tests/test_hip_f16x3.py), ~5x less matrix-pipe time;
    activations must stay inside the f16 range (checked on the device, `check_status`).
    "f32": every product on the exact fp32 MFMA."""
    global _CONV_PRECISION, _PRESPLIT
    assert mode in ("f32", "f16x3", "f16x3-inline")
    _PRESPLIT = mode != "f16x3-inline"       # "-inline": the kernel converts at gather time (mode 1; for comparison)
    _CONV_PRECISION = "f32" if mode == "f32" else "f16x3"


_TLS = threading.local()


def conv_precision() -> str:
    """The precision in force for the calling thread: a `precision_override` block, else the process-wide setting."""
    return getattr(_TLS, "precision", None) or _CONV_PRECISION


@contextlib.contextmanager
def precision_override(mode: str):
    """Run the enclosed launches of THIS thread with another convolution precision ("f32" | "f16x3") without touching the
    process-wide setting - other scenes in flight on other threads keep theirs.  PascoNet.forward uses it to redo a step
    on the exact fp32 path when the f16 range flag of the split-precision operands was raised."""
    assert mode in ("f32", "f16x3")
    prev = getattr(_TLS, "precision", None)
    _TLS.precision = mode
    try:
        yield
    finally:
        _TLS.precision = prev


def optimistic() -> bool:
    """True where the graph may take a shortcut WITHOUT the host read that would justify it, leaving a device-side flag for
    the end-of-step check instead (`CBackend.optimistic_word`; `PascoNet.forward` redoes the step with the shortcuts off when
    the flag was raised): the attention-mask block lookups, the "kept rows are the leading rows" selection of the mask
    transformer's outputs, the dense bottleneck's "no all-zero site".  Each saves one host synchronisation per use; each is
    exact whenever its flag stays down.  OFF unless the caller switched it on for the current thread
    (`optimistic_override(True)`) - only a caller that ends the step with `CBackend.check_status` and can redo it may do so:
    `PascoNet.forward` does (PASCO_OPTIMISTIC=0 keeps it off); `UNet3DV2` / the modules used on their own take the checked
    paths."""
    return bool(getattr(_TLS, "optimistic", False))


def optimistic_word(device):
    """The stream's optimistic word (`CBackend.optimistic_word`) when shortcuts are on and a backend serves `device`, else
    None (the caller then takes its checked path)."""
    if not optimistic():
        return None
    from ..me.backend import backend_for
    try:
        return backend_for(device).optimistic_word(device)
    except RuntimeError:
        return None


@contextlib.contextmanager
def optimistic_override(on: bool):
    prev = getattr(_TLS, "optimistic", None)
    _TLS.optimistic = bool(on)
    try:
        yield
    finally:
        _TLS.optimistic = prev


def set_fusion(on: bool) -> None:
    """False = the UNFUSED drop-in route (INTEGRATION.md route (a)): `conv` runs what the reference's module trees
    launch on `pasco_amd.me` - MinkowskiBatchNorm, MinkowskiReLU / LeakyReLU, the plain `MinkowskiConvolution`
    forward (exact fp32 products, bias only), a separate residual add - and the tall linear layers run as torch
    modules.  Same numbers as the fused graph to fp32 rounding; it exists so that the drop-in route has a GPU test
    and a benchmark row of its own."""
    global _FUSION
    _FUSION = bool(on)


def fusion() -> bool:
    return _FUSION


def _act_rows(F: torch.Tensor, act: int, slope: float) -> torch.Tensor:
    if act == ACT_RELU:
        return torch.relu(F)
    if act == ACT_LEAKY:
        return torch.nn.functional.leaky_relu(F, slope)
    return F


def _module_act(y: SparseTensor, act: int, slope: float) -> SparseTensor:
    """What a MinkowskiReLU / MinkowskiLeakyReLU module in the reference's trees does with `y` (pasco_amd.me.modules: recorded on
    the returned tensor in inference, applied by the next convolution's prologue or on first read)."""
    if act == ACT_NONE:
        return y
    from ..me.modules import MinkowskiLeakyReLU, MinkowskiReLU
    mod = MinkowskiReLU() if act == ACT_RELU else MinkowskiLeakyReLU(slope)
    return mod(y)


def _conv_unfused(x, mod, pro_bn, pro_act, epi_bn, epi_act, epi2_bn, residual, res_act, slope, out_key, nbr):
    """The module-by-module sequence (route (a)): BN -> act -> conv (+ bias) -> BN -> act -> BN -> (+ residual) -> act, every
    step through the plain pasco_amd.me modules, as the reference's trees call them."""
    mgr = x.coordinate_manager
    same = lambda t, feats: SparseTensor(feats, coordinate_map_key=t.coordinate_map_key, coordinate_manager=mgr)
    bn_of = lambda bn, t: bn(t) if isinstance(bn, MinkowskiBatchNorm) else same(t, bn(t.F))
    y = x
    if pro_bn is not None:
        y = bn_of(pro_bn, y)
    y = _module_act(y, pro_act, slope)
    if out_key is None:
        y = mod(y)                      # MinkowskiConvolution.forward: its own maps, its own (guarded) launches
    else:                               # the caller fixed the output map (pruned generative expansion)
        raw, pending = y.take_prologue()
        y = SparseTensor(mod.conv_rows(mgr.backend(), raw, nbr, mgr.size(out_key), mgr=mgr, prologue=pending),
                         coordinate_map_key=out_key, coordinate_manager=mgr)
    if epi_bn is not None:
        y = bn_of(epi_bn, y)
    y = _module_act(y, epi_act, slope)
    if epi2_bn is not None:
        y = bn_of(epi2_bn, y)
    if residual is not None:
        y = same(y, y.F + residual)
    return _module_act(y, res_act, slope)


def publish(t):
    """Call before a tensor derived from the parameters (split weights, folded BatchNorms, composed projections, tables)
    is stored in a module-wide cache: the cache is read by every stream (scenes in flight each have their own), but the
    tensor was made by launches on THIS one - the producing stream is drained first, so that no other stream can find the
    entry before its data exists.  Once per cache entry and parameter version (warm-up), never per step; inside a graph
    capture nothing may synchronise (the capture's warm-up calls have built the entries already)."""
    dev = t.device if torch.is_tensor(t) else t
    if dev.type == "cuda" and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(dev).synchronize()
    return t


def _split_of(w, be):
    return be.split_weight_rows(w) if _PRESPLIT else be.split_weight_f16(w)


def split_weight(mod, be):
    """Cached f16 split of a conv module's kernel: (w_split, unscale) for mode 2, (hi, lo, unscale) for mode 1."""
    w = mod.kernel
    ver = (w._version, w.device, _PRESPLIT)
    hit = getattr(mod, "_ph_split", None)
    if hit is None or hit[0] != ver:
        hit = (ver, _split_of(w, be))
        publish(w.device)
        object.__setattr__(mod, "_ph_split", hit)
    return hit[1]


def split_input(x: SparseTensor, be, ps, pb, pro_act, slope):
    """Pre-split rows of act(x.F * ps + pb), cached on the SparseTensor (several consumers of one tensor with the
    same prologue - e.g. the per-subnet heads - split it once)."""
    F = x.F if x.F.is_contiguous() else x.F.contiguous()
    key = _split_key(F, ps, pb, pro_act, slope)
    cache = x.__dict__.setdefault("_ph_in_split", {})
    hit = cache.get(key)
    if hit is None:
        hit = be.split_rows(F, pro_scale=ps, pro_shift=pb, pro_act=pro_act, slope=slope)
        cache[key] = hit
    return hit


def gathered_split(x: SparseTensor, rows: torch.Tensor, out_key):
    """The rows `rows` of `x` as the pre-split operand of a following `conv` (a `SplitRows` on the map `out_key`): the
    operand of `x` is made once (cached on `x`, shared by every gather from it) and its ROWS are gathered - splitting is
    row-wise, so split(gather(x)) == gather(split(x)) bit for bit - instead of gathering fp32 rows and splitting each
    gathered copy.  None when the split path does not apply (the caller gathers fp32 rows)."""
    mgr = x.coordinate_manager
    be = mgr.backend()
    c = x.F.shape[1]
    if not (_FUSION and _PRESPLIT and conv_precision() == "f16x3" and be.split_supported(c, c) and x.F.shape[0] > 0):
        return None
    op = split_input(x, be, None, None, ACT_NONE, 0.01)                     # [n, cpad / 32, 2, 32] f16
    n, cpad = op.shape[0], op.shape[1] * 32
    got = be.gather_rows(op.view(torch.float32).view(n, cpad), rows)         # 4 bytes per channel: rows of cpad "floats"
    return SplitRows(got.view(torch.float16).view(got.shape[0], cpad // 32, 2, 32), c, out_key, mgr)


def split_rows_2d(x2d: torch.Tensor):
    """Pre-split operand of a tall [N, cin] matrix for several `linear_rows` calls on it (None when the split
    path does not apply)."""
    if not (_FUSION and _kernel_device(x2d.device) and conv_precision() == "f16x3" and _PRESPLIT and x2d.shape[1] % 8 == 0):
        return None
    from ..me.backend import backend_for
    return backend_for(x2d.device).split_rows(x2d.contiguous())


def linear_rows(x2d: Optional[torch.Tensor], weight: torch.Tensor, bias, cache_owner, cache_key: str,
                min_rows: Optional[int] = None, in_split=None, residual: Optional[torch.Tensor] = None,
                emit: bool = False, want_out: bool = True, axis=None, out: Optional[torch.Tensor] = None):
    """y = x @ weight.T + bias for a tall [N, cin] operand (`weight` is an nn.Linear-style [cout, cin] tensor
    or a row slice of one).  Large N on the GPU goes through the convolution kernel as an identity-map k=1
    convolution - the same split-precision MFMA GEMM with fused bias - instead of an fp32 library GEMM;
    everything else (small N, CPU checker backend, odd shapes) is torch.nn.functional.linear.
    `residual` [N, cout] (optional) is added in the same launch (y + residual); `axis` = (table [3, T, cout], coords
    int32 [N, 4], lo) adds the per-axis table rows table[0][x - lo] + table[1][y - lo] + table[2][z - lo] the same way
    (the sine position encoding without materialising it).
    `x2d` may be None when `in_split` (its pre-split operand) is given.
    `emit`: also return the pre-split operand of y for a following linear_rows / batched_rows_matmul -> (y, y_split);
    y_split is None when the split path did not apply, and with `want_out=False` y is None when it did.
    `out` (optional, contiguous fp32 [N, cout]): the kernel path writes y there (a row slice of a larger tensor)."""
    cout, cin = weight.shape
    n = x2d.shape[0] if x2d is not None else in_split.shape[0]
    dev = x2d.device if x2d is not None else in_split.device
    be = None
    min_rows = MIN_ROWS_LINEAR if min_rows is None else min_rows
    if _FUSION and _kernel_device(dev) and n >= min_rows and conv_precision() == "f16x3":
        from ..me.backend import backend_for
        be = backend_for(dev)
        if not be.split_supported(cin, cout):
            be = None
    if be is None:
        assert x2d is not None, "linear_rows: a pre-split operand needs the split path"
        y = torch.nn.functional.linear(x2d, weight, bias)
        y = y if residual is None else y + residual
        if axis is not None:
            y = y + axis_rows(axis)
        if out is not None:
            out.copy_(y)
            y = out
        return (y, None) if emit else y
    ver = (weight._version, weight.device, weight.data_ptr(), _PRESPLIT)
    hit = cache_owner.__dict__.get("_ph_lin_" + cache_key)
    if hit is None or hit[0] != ver:
        wt = weight.detach().t().contiguous()                     # [cin, cout]
        hit = (ver, wt, _split_of(wt, be), bias.detach().contiguous() if bias is not None else None)
        publish(wt)
        cache_owner.__dict__["_ph_lin_" + cache_key] = hit
    _, wt, split, b = hit
    do_emit = emit and _PRESPLIT and cout % 32 == 0
    out = be.conv_fwd(None if (x2d is None) else x2d.contiguous(), wt, None, n,
                      xshape=(n, cin) if x2d is None else None, bias=b, split=split,
                      in_split=in_split if _PRESPLIT else None,
                      residual=None if residual is None else residual.contiguous(),
                      emit_split=(None, None, ACT_NONE) if do_emit else None, want_out=want_out or not do_emit,
                      axis=axis, out=out)
    if not emit:
        return out
    return out if do_emit else (out, None)


def axis_rows(axis) -> torch.Tensor:
    """The rows the per-axis table residual stands for (torch formulation: reference / fallback paths)."""
    tab, coords, lo = axis
    idx = (coords[:, 1:4].long() - lo).clamp(0, tab.shape[1] - 1)
    return tab[0][idx[:, 0]] + tab[1][idx[:, 1]] + tab[2][idx[:, 2]]


def linear_bn_act(x2d: Optional[torch.Tensor], lin: nn.Linear, *, pro_bn=None, epi_bn=None, epi_act: int = ACT_NONE,
                  min_rows: Optional[int] = None, in_split=None, emit: bool = False, emit_into: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None):
    """act(BN_epi(Linear(BN_pro(x)))) for a tall [N, cin] operand as ONE launch of the convolution kernel
    (identity map, eval BatchNorm folded into the gather prologue / store epilogue) - the point MLP of
    CylinderFeat (unet3d_sparse_v2.py:27-43) without separate normalisation / activation passes.
    Small N, CPU tensors: plain torch modules.
    `emit`: the result is read only by the next layer of the chain -> return (None, operand) with the result stored
    only as that layer's pre-split operand ((y, None) when the split path does not apply); `in_split` = such an
    operand from the previous layer (then `x2d` is None).  `emit_into` = the rows of a larger operand tensor the emitted
    operand is to be written to (several inputs feeding one chain without concatenating them first); `out` = the same for
    the fp32 result (kernel route only)."""
    cin, cout = lin.in_features, lin.out_features
    n = x2d.shape[0] if x2d is not None else in_split.shape[0]
    dev = x2d.device if x2d is not None else in_split.device
    min_rows = MIN_ROWS_LINEAR if min_rows is None else min_rows
    if not (_FUSION and _kernel_device(dev) and n >= min_rows):
        assert x2d is not None
        y = x2d if pro_bn is None else pro_bn(x2d)
        y = lin(y)
        y = y if epi_bn is None else epi_bn(y)
        y = torch.relu(y) if epi_act == ACT_RELU else y
        return (y, None) if emit else y
    from ..me.backend import backend_for
    be = backend_for(dev)
    ps = pb = es = eb = None
    if pro_bn is not None:
        ps, pb = fold_bn(pro_bn)
    if epi_bn is not None:
        es, eb = fold_bn(epi_bn)
    w = lin.weight
    ver = (w._version, w.device, w.data_ptr(), _PRESPLIT, conv_precision())
    hit = lin.__dict__.get("_ph_lin_w")
    if hit is None or hit[0] != ver:
        wt = w.detach().t().contiguous()                          # [cin, cout]
        split = _split_of(wt, be) if (conv_precision() == "f16x3" and be.split_supported(cin, cout)) else None
        hit = (ver, wt, split, lin.bias.detach().contiguous() if lin.bias is not None else None)
        publish(wt)
        lin.__dict__["_ph_lin_w"] = hit
    _, wt, split, b = hit
    do_emit = emit and split is not None and _PRESPLIT and cout % 32 == 0
    assert x2d is not None or (split is not None and _PRESPLIT), "a pre-split operand needs the split path"
    out = be.conv_fwd(None if x2d is None else x2d.contiguous(), wt, None, n, xshape=(n, cin) if x2d is None else None,
                      bias=b, pro_scale=ps, pro_shift=pb, epi_scale=es, epi_shift=eb, epi_act=epi_act, split=split,
                      in_split=in_split if (split is not None and _PRESPLIT) else None,
                      emit_split=(None, None, ACT_NONE) if do_emit else None, want_out=not do_emit,
                      in_split_has_prologue=True, out_split=emit_into if do_emit else None, out=None if do_emit else out)
    if not emit:
        return out
    return out if do_emit else (out, None)


def prepare_batched_weights(w: torch.Tensor):
    """Operand form of per-batch weights w [B, Q, D] that change every call (the mask-embedding of the query
    heads): (w_split [B*Q, D/32, 2, 32], unscale [Q]).  The power-of-two scale is chosen on the device (no host
    read) and undone by the epilogue scale; launch-only, so it can sit inside a captured graph."""
    from ..me.backend import backend_for
    be = backend_for(w.device)
    B, Q, D = w.shape
    w = w.detach()
    e = 13 - torch.frexp(w.abs().amax())[1]                        # device int: largest magnitude just below 2^14
    # exact powers of two built from the exponent bits (torch.ldexp / pow on the GPU are not exact)
    e = e.to(torch.int32).clamp(-100, 100)
    up = ((e + 127) << 23).view(torch.float32)
    down = ((127 - e) << 23).view(torch.float32)
    w_split = be.split_rows((w * up).reshape(B * Q, D).contiguous(), exp2=0)
    unscale = (torch.ones(Q, device=w.device) * down).contiguous()
    return w_split, unscale


def batched_rows_matmul(x: Optional[torch.Tensor], w: Optional[torch.Tensor], x_split: torch.Tensor, shape=None,
                        prepared=None, bias=None, axis=None) -> torch.Tensor:
    """out[b] = x[b] @ w[b].T for tall x [B, P, D] and per-batch w [B, Q, D] (the mask logits of the query heads),
    on the split-precision kernel with `x_split` = split_rows(x) prepared once.  `prepared` =
    prepare_batched_weights(w) when the caller already has it.  `bias` [B, Q] and `axis` = (tables [B, 3, T, Q],
    coords int32 [B, P, 4], lo) are added after the product (the absorbed form of the mask heads: per-batch bias and
    per-axis table rows)."""
    from ..me.backend import backend_for
    dev = x_split.device
    be = backend_for(dev)
    B, P, D = x.shape if x is not None else shape     # x itself is not read (only its operand split)
    w_split, unscale = prepared if prepared is not None else prepare_batched_weights(w)
    Q = unscale.numel()
    out = torch.empty((B, P, Q), dtype=torch.float32, device=dev)
    xs = x_split.reshape(B, P, -1)
    ws = w_split.reshape(B, Q, -1)
    for b in range(B):
        be.conv_fwd(None, None, None, P, xshape=(P, D), wshape=(1, D, Q), split=(ws[b], 1.0), in_split=xs[b],
                    epi_scale=unscale, out=out[b], epi2_shift=None if bias is None else bias[b],
                    axis=None if axis is None else (axis[0][b], axis[1][b], axis[2]))
    return out


def fold_bn(bn) -> Tuple[torch.Tensor, torch.Tensor]:
    """BatchNorm (eval) -> (scale, shift) with y = x * scale + shift. Cached per module version."""
    m = bn.bn if isinstance(bn, MinkowskiBatchNorm) else bn
    assert isinstance(m, nn.modules.batchnorm._BatchNorm)
    assert not m.training, "fused graph serves inference (module.eval()) only"
    # ~90 calls per step: the tensors straight from the module's dictionaries (nn.Module.__getattr__ costs ~1 us a name)
    bufs, pars = m._buffers, m._parameters
    rm, rv, w, b = bufs["running_mean"], bufs["running_var"], pars.get("weight"), pars.get("bias")
    ver = (rm._version, rv._version, w._version if w is not None else -1, b._version if b is not None else -1, rm.device)
    hit = m.__dict__.get("_ph_folded")
    if hit is not None and hit[0] == ver:
        return hit[1], hit[2]
    with torch.no_grad():
        inv = torch.rsqrt(m.running_var.float() + m.eps)
        scale = inv * (m.weight.float() if m.weight is not None else 1.0)
        shift = (m.bias.float() if m.bias is not None else 0.0) - m.running_mean.float() * scale
        scale, shift = scale.contiguous(), shift.contiguous()
    publish(scale)
    m._ph_folded = (ver, scale, shift)
    return scale, shift


class SplitRows:
    """Rows that exist only as the pre-split operand of the next convolution (their fp32 values were never
    written): what `conv(..., split_only=True)` returns inside a block whose intermediate has one reader."""

    def __init__(self, split: torch.Tensor, channels: int, coordinate_map_key, coordinate_manager):
        self.split = split
        self.channels = channels
        self.coordinate_map_key = coordinate_map_key
        self.coordinate_manager = coordinate_manager


def _split_key(F, ps, pb, pro_act, slope):
    return (F.data_ptr(), F._version, None if ps is None else ps.data_ptr(), None if pb is None else pb.data_ptr(),
            pro_act, float(slope) if pro_act == ACT_LEAKY else 0.0)


def conv(x: SparseTensor, mod: _ConvBase, *, pro_bn=None, pro_act: int = ACT_NONE, epi_bn=None,
         epi_act: int = ACT_NONE, epi2_bn=None, residual: Optional[torch.Tensor] = None,
         res_act: int = ACT_NONE, slope: float = 0.01, out_key=None, nbr=None, emit_next=None,
         split_only: bool = False, one_pair: bool = False, out: Optional[torch.Tensor] = None):
    """One fused launch of a Minkowski-style convolution module on `x`.

    out = act_res( act_epi(BN_epi(conv(act_pro(BN_pro(x))) + bias)) -> BN_epi2 -> (+ residual) )

    `emit_next` = (bn | None, act): the caller knows the next convolution reads act(BN(out)); the launch then also
    writes that convolution's pre-split operand (no separate ph_split_rows pass) and leaves it in the returned
    tensor's operand cache, where the next `conv` call finds it.  With `split_only` (the result has exactly that one
    reader) the fp32 result is not written at all and a `SplitRows` is returned - when the split path does not
    apply, a normal SparseTensor comes back.  `x` may itself be a `SplitRows`.
    `one_pair`: the caller guarantees that every output row of the map has exactly one (offset, input row) pair (the
    generative transposed convolutions: each child has its one parent) - the launch then runs as k = 1 products over row
    lists grouped by offset instead of walking all offsets of every row.
    `out`: a contiguous fp32 [n_out, out_channels] buffer the result is written into (a slice of a batch the caller assembles:
    no copy afterwards)."""
    mgr = x.coordinate_manager
    if not _FUSION:
        assert not isinstance(x, SplitRows)
        y = _conv_unfused(x, mod, pro_bn, pro_act, epi_bn, epi_act, epi2_bn, residual, res_act, slope, out_key, nbr)
        if out is not None:
            out.copy_(y.F)
            y = SparseTensor(out, coordinate_map_key=y.coordinate_map_key, coordinate_manager=mgr)
        return y
    if out_key is None:
        out_key, nbr = mod._maps(x)
    n_out = mgr.size(out_key)
    ps = pb = es = eb = e2s = e2b = None
    if pro_bn is not None:
        ps, pb = fold_bn(pro_bn)
    if epi_bn is not None:
        es, eb = fold_bn(epi_bn)
    if epi2_bn is not None:
        e2s, e2b = fold_bn(epi2_bn)
    bias = mod.bias.detach().reshape(-1) if mod.bias is not None else None
    be = mgr.backend()
    split = in_split = None
    x_rows = None
    if isinstance(x, SplitRows):
        assert pro_bn is None and pro_act == ACT_NONE, "a pre-split input already carries its prologue"
        split, in_split = split_weight(mod, be), x.split
        xshape = (in_split.shape[0], x.channels)
    else:
        x_rows = x.F if x.F.is_contiguous() else x.F.contiguous()
        xshape = None
        if conv_precision() == "f16x3" and be.split_supported(mod.in_channels, mod.out_channels):
            split = split_weight(mod, be)
            if _PRESPLIT and n_out > 0:
                in_split = split_input(x, be, ps, pb, pro_act, slope)
    emit = None
    if emit_next is not None and in_split is not None and mod.out_channels % 32 == 0:
        nbn, nact = emit_next
        ns, nb = fold_bn(nbn) if nbn is not None else (None, None)
        emit = (ns, nb, nact)
    only = split_only and emit is not None and n_out > 0
    win = None
    if in_split is not None and nbr is not None and nbr.shape[0] == 27 and n_out >= MIN_ROWS_WINDOWS and \
            be.device_type == "cuda" and (33 <= mod.out_channels <= 64 or _WINDOWS_WIDE):   # 64-wide tiles (measured)
        win = mgr.kernel_windows(nbr)
    rowlist = None
    if one_pair and in_split is not None and nbr is not None and nbr.shape[0] <= 8 and n_out >= MIN_ROWS_LINEAR and \
            be.device_type == "cuda" and os.environ.get("PASCO_CONV_RL", "1") != "0":
        rowlist = mgr.kernel_rowlist(nbr)
    if out is not None:
        assert not only and out.is_contiguous() and tuple(out.shape) == (n_out, mod.out_channels) and out.dtype == torch.float32
    out = be.conv_fwd(
        x_rows, mod.kernel.detach(), nbr, n_out, xshape=xshape, bias=bias,
        pro_scale=ps, pro_shift=pb, pro_act=pro_act, epi_scale=es, epi_shift=eb, epi_act=epi_act,
        epi2_scale=e2s, epi2_shift=e2b, residual=residual, res_act=res_act, slope=slope, split=split,
        in_split=in_split, emit_split=emit, want_out=not only, win=win, in_split_has_prologue=True, rowlist=rowlist,
        out=out if n_out > 0 else None)
    if emit is None:
        return SparseTensor(out, coordinate_map_key=out_key, coordinate_manager=mgr)
    out, out_split = out
    if only:
        assert emit[0] is None and emit[1] is None and emit[2] == ACT_NONE, "split_only serves a plain next operand"
        return SplitRows(out_split, mod.out_channels, out_key, mgr)
    y = SparseTensor(out, coordinate_map_key=out_key, coordinate_manager=mgr)
    # key it the way the next conv's split_input will look it up (a leaky prologue would use this launch's slope)
    y.__dict__.setdefault("_ph_in_split", {})[_split_key(y.F, emit[0], emit[1], emit[2], slope)] = out_split
    return y


__all__ = ["fold_bn", "conv", "set_conv_precision", "conv_precision", "precision_override", "optimistic", "optimistic_override", "set_fusion", "fusion", "linear_rows", "split_rows_2d", "gathered_split", "batched_rows_matmul", "prepare_batched_weights", "linear_bn_act", "ACT_NONE", "ACT_RELU", "ACT_LEAKY"]
