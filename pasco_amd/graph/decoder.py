"""Generative sparse decoder (reference: pasco/models/decoder_v3.py:77-172 `DecoderBlock`,
:175-511 `DecoderGenerativeSepConvV2`).

Per level (tensor stride 4, 2, 1): generative up-sampling, pruning to the global bounds, coordinate
channels + BN + conv k1 ("resize"), union-add with the encoder skip, residual blocks, one
completion head per MIMO subnet; then the occupied voxels are kept (OR over subnets of
argmax != 0, decoder_v3.py:337-339,380-381) and the per-subnet panoptic branch feeds the mask
transformer.

Launch-level differences to the reference, results unchanged:
  * the bounds prune is applied to the generated coordinates BEFORE the up-convolution's features
    are computed (children outside the bounds are never materialised);
  * prune-by-class and prune-by-bounds in predict_panop are one compaction (mask AND);
  * BN / activations ride in conv prologues / epilogues.
`keep_override` (benchmark only, SURVEY.md 8(d) "teacher-forced keep") replaces the argmax-derived
masks by membership tests against given voxel sets.
"""
from __future__ import annotations

import os
from collections import defaultdict
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import me as ME
from . import fused
from .blocks import BasicGenerativeDeconvolutionBlock, ResidualBlock, SpatialDropout, run_sequential
from .fused import ACT_NONE, ACT_RELU


def _corner(v, coords: torch.Tensor) -> torch.Tensor:
    """A box corner as a [1, 3] tensor of the coordinates' dtype on their device; the cast of a device tensor is kept on it
    (`_ph_corner`: the same corner tensor of the batch dict is tested against at every level - one cast kernel, not one per
    call)."""
    if torch.is_tensor(v):
        # keyed by stream as well: the cast made on another scene thread's stream may not have run yet when this one reads it
        stream = torch.cuda.current_stream(coords.device).cuda_stream if coords.device.type == "cuda" else 0
        key = (coords.dtype, coords.device, stream, v._version)
        cache = getattr(v, "_ph_corner", None)
        if cache is not None and key in cache:
            return cache[key]
        out = v.to(device=coords.device, dtype=coords.dtype).reshape(1, 3)
        try:
            if cache is None or len(cache) > 8:
                cache = {}
                v._ph_corner = cache
            cache[key] = out
        except AttributeError:
            pass
        return out
    return torch.as_tensor(v, device=coords.device).to(coords.dtype).reshape(1, 3)


def inside_bounds(coords: torch.Tensor, lo, hi) -> torch.Tensor:
    """Inclusive box test on int32 [N,4] coordinates (decoder_v3.py:151-158, misc.py:16-27)."""
    lo, hi = _corner(lo, coords), _corner(hi, coords)
    xyz = coords[:, 1:]
    return ((xyz >= lo) & (xyz <= hi)).all(dim=1)


def box_mask(mgr, coords: torch.Tensor, lo, hi) -> torch.Tensor:
    """`inside_bounds` in one launch where the library serves it (ph_keep_mask without a source = the box test alone) instead of
    four element-wise torch kernels per level."""
    if fused.fusion() and coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4:
        try:
            be = mgr.backend()
        except Exception:
            be = None
        if be is not None and be.has("keep_mask") and os.environ.get("PASCO_KEEP_FUSED", "1") != "0":
            lo_t, hi_t = _corner(lo, coords).reshape(3).contiguous(), _corner(hi, coords).reshape(3).contiguous()
            return be.keep_mask([], coords.contiguous(), lo_t, hi_t)
    return inside_bounds(coords, lo, hi)


_FIRST_ROWS = {}


def _first_rows(n: int, device, k: int = 1000) -> torch.Tensor:
    """bool [n]: the first k rows (the reference's "nothing kept" fallback selection); one tensor per (n, device, stream)
    instead of an arange + compare per call."""
    # per stream: a tensor made on another scene thread's stream may not have been written yet when this stream reads it
    stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    key = (n, str(device), k, stream)
    hit = _FIRST_ROWS.get(key)
    if hit is None:
        if len(_FIRST_ROWS) > 64:
            _FIRST_ROWS.clear()
        hit = torch.arange(n, device=device) < k
        _FIRST_ROWS[key] = hit
    return hit


def batch_sparse_tensor(tensors: List[ME.SparseTensor], n_max: Optional[int] = None):
    """Zero-pad per-subnet tensors to [M, Nmax, C] / [M, Nmax, 4] (reference: pasco/models/utils.py:659-670).
    `n_max` overrides the pad length (subnet-parallel heads pad to the longest subnet of ALL ranks, because
    the padded rows take part in the attention - SURVEY.md section 9 item 5)."""
    n_max = max([t.F.shape[0] for t in tensors] + ([n_max] if n_max is not None else []))
    f0, c0 = tensors[0].F, tensors[0].C
    bf = f0.new_empty((len(tensors), n_max, f0.shape[1]))
    bc = c0.new_empty((len(tensors), n_max, c0.shape[1]))
    for i, t in enumerate(tensors):          # rows copied once, only the padding is zero-filled
        n = t.F.shape[0]
        bf[i, :n] = t.F
        bc[i, :n] = t.C
        if n < n_max:
            bf[i, n:].zero_()
            bc[i, n:].zero_()
    return bf, bc


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, n_heads, compl_head_dim, heavy_decoder=True, dropout=0.0):
        super().__init__()
        self.upsample = BasicGenerativeDeconvolutionBlock(in_channels, out_channels, ks=2, stride=2)
        self.resize = nn.Sequential(
            ME.MinkowskiBatchNorm(out_channels + 3),
            ME.MinkowskiConvolution(out_channels + 3, out_channels, kernel_size=1, bias=True, dimension=3),
        )
        n_res = 7 if heavy_decoder else 3
        layers = [ResidualBlock(out_channels, out_channels) for _ in range(n_res)]
        if heavy_decoder:
            layers.append(SpatialDropout(p=dropout))
        self.process = nn.Sequential(*layers)
        self.n_heads = n_heads
        self.completion_heads = nn.ModuleDict({
            str(i): nn.Sequential(ME.MinkowskiConvolution(out_channels, compl_head_dim, kernel_size=1, bias=True,
                                                          dimension=3))
            for i in range(n_heads)})

    def forward(self, x: ME.SparseTensor, shortcut: ME.SparseTensor, global_min, global_max):
        mgr = x.coordinate_manager
        up = self.upsample.net[0]
        # children -> bounds prune -> features only for surviving children
        out_key = mgr.expand_pruned(x.coordinate_map_key, up.stride, lambda kids: box_mask(mgr, kids, global_min, global_max))
        nbr = mgr.kernel_map(x.coordinate_map_key, out_key, up.kernel_size, up.dilation, transposed=True)
        # `dec + shortcut` (decoder_v3.py:163) lists the rows of `dec` first: when the absorbed `resize` applies, its launch
        # writes them straight into the union's feature tensor (round 5) - the up-sampled features are never copied (the
        # finest level's copy was 2 x 175 MB per step); same values, same order of the additions
        y_in = None
        if self._resize_applies(x, out_key):
            ukey, _, b2o = mgr.union(out_key, shortcut.coordinate_map_key)
            n_un, n_dec = mgr.size(ukey), mgr.size(out_key)
            buf = torch.empty((n_un, self.resize[1].out_channels), dtype=torch.float32, device=x.F.device)
            dec = self._resize_absorbed(x, out_key, nbr, out=buf[:n_dec])
            if dec is not None and dec.F.data_ptr() == buf.data_ptr():
                if n_un > n_dec:
                    buf[n_dec:].zero_()
                mgr.backend().scatter_add_rows(shortcut.F.contiguous(), b2o.contiguous(), buf)
                y_in = ME.SparseTensor(buf, coordinate_map_key=ukey, coordinate_manager=mgr)
        else:
            dec = None
        if y_in is None and dec is None:
            dec = self.upsample(x, out_key=out_key, nbr=nbr)
            # coordinate channels (absolute coords / tensor stride) + BN + conv k1 with bias
            ts = dec.tensor_stride[0]
            feats = torch.cat([dec.F, dec.C[:, 1:].float() / ts], dim=1)
            dec = ME.SparseTensor(feats, coordinate_map_key=out_key, coordinate_manager=mgr)
            dec = fused.conv(dec, self.resize[1], pro_bn=self.resize[0], pro_act=ACT_NONE)
        if y_in is None:
            y_in = dec + shortcut
        y = run_sequential(self.process, y_in)
        logits = [fused.conv(y, self.completion_heads[str(i)][0]) for i in range(self.n_heads)]
        return y, logits


    # -- `resize` without the concatenation --------------------------------------------------------------------------
    RESIZE_LO, RESIZE_ROWS = -1024, 5120     # coordinate values the tables cover (a coordinate outside raises the status flag)

    def _resize_applies(self, x, out_key) -> bool:
        """The conditions of the absorbed form (`_resize_absorbed`)."""
        mgr = x.coordinate_manager
        conv = self.resize[1]
        c_in, c_out = conv.in_channels - 3, conv.out_channels
        be = mgr.backend()
        return bool(fused.fusion() and fused.conv_precision() == "f16x3" and fused._PRESPLIT and fused._kernel_device(x.F.device)
                    and mgr.size(out_key) >= fused.MIN_ROWS_LINEAR and c_in % 32 == 0 and be.split_supported(c_in, c_out)
                    and os.environ.get("PASCO_RESIZE_ABSORB", "1") != "0")

    def _resize_absorbed(self, x, out_key, nbr, out=None):
        """upsample -> [features | coords / ts] -> BN -> 1x1 conv (decoder_v3.py:103,133) without forming the C + 3 channel
        tensor: BN has no activation behind it, so
            out = F (s_f * W_f) + (bias + b_f W_f) + sum_axis ((c_axis / ts) s_axis + b_axis) W_axis
        = a C -> C product on the up-sampled features (whose launch writes them ONLY as the split operand) plus three rows of
        a per-axis table (ph_conv_desc.axis_table).  No 183 MB concatenation, no odd channel count on the fp32 MFMA path.
        None when the fused split path does not apply (the caller then runs the module sequence)."""
        mgr = x.coordinate_manager
        n_out = mgr.size(out_key)
        conv, bn = self.resize[1], self.resize[0]
        c_out = conv.out_channels
        c_in = conv.in_channels - 3
        be = mgr.backend()
        if not self._resize_applies(x, out_key):
            return None
        dec = self.upsample(x, out_key=out_key, nbr=nbr, emit_next=(None, ACT_NONE), split_only=True)
        if not isinstance(dec, fused.SplitRows):
            return self._resize_plain(dec, out_key)
        ts = out_key.tensor_stride[0]
        w = conv.kernel
        ver = (w._version, w.data_ptr(), conv.bias._version if conv.bias is not None else -1, ts) + tuple(
            (t._version, t.data_ptr()) for t in (bn.bn.running_mean, bn.bn.running_var, bn.bn.weight, bn.bn.bias) if t is not None)
        hit = self.__dict__.get("_ph_resize")
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                s_, b_ = fused.fold_bn(bn)
                s64, b64, w64 = s_.double(), b_.double(), w.detach().reshape(c_in + 3, c_out).double()
                wf = (s64[:c_in, None] * w64[:c_in]).t().contiguous().float()                     # [c_out, c_in]
                bias = b64[:c_in] @ w64[:c_in]
                if conv.bias is not None:
                    bias = bias + conv.bias.detach().reshape(-1).double()
                idx = torch.arange(self.RESIZE_ROWS, device=w.device, dtype=torch.float64) + self.RESIZE_LO
                tab = torch.stack([((idx / ts) * s64[c_in + a] + b64[c_in + a])[:, None] * w64[c_in + a][None, :]
                                   for a in range(3)]).float().contiguous()                       # [3, T, c_out]
            hit = (ver, wf, bias.float().contiguous(), tab)
            fused.publish(tab)
            self.__dict__["_ph_resize"] = hit
        _, wf, bias, tab = hit
        coords = mgr.get_coordinates(out_key)
        out = fused.linear_rows(None, wf, bias, self, "resize", in_split=dec.split,
                                axis=(tab, coords.contiguous(), self.RESIZE_LO), out=out)
        return ME.SparseTensor(out, coordinate_map_key=out_key, coordinate_manager=mgr)

    def _resize_plain(self, dec, out_key):
        mgr = dec.coordinate_manager
        ts = dec.tensor_stride[0]
        feats = torch.cat([dec.F, dec.C[:, 1:].float() / ts], dim=1)
        dec = ME.SparseTensor(feats, coordinate_map_key=out_key, coordinate_manager=mgr)
        return fused.conv(dec, self.resize[1], pro_bn=self.resize[0], pro_act=ACT_NONE)


class DecoderGenerativeSepConvV2(nn.Module):
    def __init__(self, f, n_classes, transformer_predictor, n_infers, heavy_decoder=True, dropouts=(0.0, 0.0, 0.0)):
        super().__init__()
        dec_ch = list(f[::-1])
        self.n_infers = n_infers
        self.n_classes = n_classes
        self.transformer_predictor = transformer_predictor
        self.dec_blocks = nn.ModuleList()
        self.voxel_feats = nn.ModuleDict()
        for i in range(len(dec_ch) - 1):
            scale = 2 ** (len(dec_ch) - 2 - i)
            self.dec_blocks.append(DecoderBlock(dec_ch[i], dec_ch[i + 1], n_heads=n_infers, compl_head_dim=n_classes,
                                                heavy_decoder=heavy_decoder, dropout=dropouts[i]))
            c = dec_ch[i + 1]
            for j in range(n_infers):
                self.voxel_feats[f"scale{scale}_infer{j}"] = nn.Sequential(
                    ME.MinkowskiConvolution(c, c, kernel_size=3, bias=False, dimension=3),
                    ME.MinkowskiBatchNorm(c),
                    ME.MinkowskiReLU(),
                    ME.MinkowskiConvolution(c, c, kernel_size=3, bias=True, dimension=3),
                )
        self.pruning = ME.MinkowskiPruning()

    # -- keep masks ---------------------------------------------------------------------------------
    @staticmethod
    def _occupied(logits: ME.SparseTensor) -> torch.Tensor:
        # argmax(softmax(l)) != 0  <=>  argmax(l) != 0   (decoder_v3.py:337-339, :411-414)
        return logits.F.argmax(dim=-1) != 0

    @staticmethod
    def _keep_sources(keep_override, scale, i, x, logits):
        """What says "kept" for subnet i at this level: the override's lookup rows (int32, >= 0 = kept) or a bool mask."""
        if keep_override is not None:
            if hasattr(keep_override, "member_rows"):
                return keep_override.member_rows(scale, i, x.C)
            return keep_override.member(scale, i, x.C)
        return DecoderGenerativeSepConvV2._occupied(logits)

    @staticmethod
    def _fused_keep(x):
        """The backend whose `keep_mask` (one pass per mask instead of ~10 element-wise torch kernels) serves x, or None."""
        if not fused.fusion() or x.C.dtype != torch.int32 or len(x.C.shape) != 2 or x.C.shape[1] != 4:
            return None
        try:
            be = x.coordinate_manager.backend()
        except Exception:
            return None
        return be if be.has("keep_mask") and os.environ.get("PASCO_KEEP_FUSED", "1") != "0" else None

    def _keep_completion(self, x, scale, sem_logits, keep_override):
        srcs = [self._keep_sources(keep_override, scale, i, x, sem_logits[i] if sem_logits is not None else None)
                for i in range(self.n_infers)]
        be = self._fused_keep(x)
        if be is not None and len(srcs) <= 8 and len({s.dtype for s in srcs}) == 1:
            return be.keep_mask([s.contiguous() for s in srcs])          # OR over the subnets, one pass
        keeps = [s >= 0 if s.dtype == torch.int32 else s for s in srcs]
        keep = keeps[0]
        for k in keeps[1:]:
            keep = keep | k
        return keep

    # -- panoptic branch ----------------------------------------------------------------------------
    def predict_panop(self, xs, sem_logits_at_scales, min_Cs, max_Cs, keep_override=None, subnets=None):
        xs_infers = defaultdict(list)
        sem_logits_pruneds = []

        def keep_mask(i, scale, x):
            logits = sem_logits_at_scales[scale][i]
            src = self._keep_sources(keep_override, scale, i, x, logits)
            be = self._fused_keep(x)
            if be is not None:
                # occupied (or, when nothing is: the first 1000 rows, decoder_v3.py:415-418) AND inside the subnet's box
                # (:151-158), decided on the device in one pass
                lo, hi = _corner(min_Cs[i], x.C).reshape(3), _corner(max_Cs[i], x.C).reshape(3)
                return be.keep_mask([src.contiguous()], x.C.contiguous(), lo.contiguous(), hi.contiguous(), fallback_rows=1000)
            keep = src >= 0 if src.dtype == torch.int32 else src
            # reference fallback (decoder_v3.py:415-418): nothing kept -> keep the first 1000 rows.  Selected on the
            # device (no host read of the count)
            first = _first_rows(keep.shape[0], keep.device)
            keep = torch.where(keep.any(), keep, first)
            return keep & inside_bounds(x.C, min_Cs[i], max_Cs[i])

        pad_to = {s: None for s in xs}
        if subnets is not None:   # pad like the full batch would: longest subnet over all subnets
            for scale, x in xs.items():
                pad_to[scale] = max(int(keep_mask(i, scale, x).sum()) for i in range(self.n_infers))
        # every (subnet, scale) prune of this branch depends only on tensors that exist now: their compactions are launched
        # back to back and the row counts read ONCE (9 map events at M = 3, one synchronisation)
        todo = [(i, scale, x, keep_mask(i, scale, x)) for i in (range(self.n_infers) if subnets is None else subnets)
                for scale, x in xs.items()]
        mgrs = {id(x.coordinate_manager) for _, _, x, _ in todo}
        if len(mgrs) == 1 and todo:
            mgr = todo[0][2].coordinate_manager
            pruned = mgr.prune_batch([(x.coordinate_map_key, keep) for _, _, x, keep in todo])
        else:
            pruned = [x.coordinate_manager.prune(x.coordinate_map_key, keep) for _, _, x, keep in todo]
        # the per-subnet voxel features go straight into the zero-padded [M, Nmax, C] batch the transformer reads
        # (batch_sparse_tensor's layout): the second convolution of each pair writes its slice, only the padding is filled
        infer_ids = list(range(self.n_infers) if subnets is None else subnets)
        n_rows = {(i, scale): x.coordinate_manager.size(out_key) for (i, scale, x, keep), (out_key, rows) in zip(todo, pruned)}
        batch_f, batch_c = {}, {}
        for scale, x in xs.items():
            n_max = max([n_rows[(i, scale)] for i in infer_ids] + ([pad_to[scale]] if pad_to[scale] is not None else []))
            batch_f[scale] = x.F.new_empty((len(infer_ids), n_max, x.F.shape[1]))
            batch_c[scale] = x.C.new_zeros((len(infer_ids), n_max, x.C.shape[1]))      # one fill per scale instead of one per slot
        # the pruned semantic logits are gathered straight into THEIR zero-padded batch (batch_sparse_tensor's layout, utils.py:
        # 659-670); its coordinate batch is the stride-1 feature batch's (same maps, same pad length): no second copy
        sem_F = None
        if 1 in xs:
            l0 = sem_logits_at_scales[1][infer_ids[0]].F
            sem_F = l0.new_empty((len(infer_ids), batch_f[1].shape[1], l0.shape[1]))
        for (i, scale, x, keep), (out_key, rows) in zip(todo, pruned):
            mgr = x.coordinate_manager
            be = mgr.backend()
            if scale == 1:      # logits and features of one map, same mask: one map event (decoder_v3.py:421-427)
                logits = sem_logits_at_scales[scale][i]
                slot1, n1_ = infer_ids.index(i), n_rows[(i, scale)]
                dst = sem_F[slot1, :n1_]
                be.gather_rows(logits.F.contiguous(), rows, out=dst)
                if n1_ < sem_F.shape[1]:
                    sem_F[slot1, n1_:].zero_()
                sem_logits_pruneds.append(ME.SparseTensor(dst, coordinate_map_key=out_key, coordinate_manager=mgr))
            # the subnet's rows of the level, as the first convolution's operand (the level is split once for all subnets)
            xi = fused.gathered_split(x, rows, out_key)
            if xi is None:
                xi = ME.SparseTensor(be.gather_rows(x.F.contiguous(), rows), coordinate_map_key=out_key, coordinate_manager=mgr)
            vf = self.voxel_feats[f"scale{scale}_infer{i}"]
            # the first convolution's only reader is the second: it writes that operand and no fp32 rows
            h = fused.conv(xi, vf[0], epi_bn=vf[1], epi_act=ACT_RELU, emit_next=(None, ACT_NONE), split_only=True)
            slot = infer_ids.index(i)
            n = n_rows[(i, scale)]
            y = fused.conv(h, vf[3], out=batch_f[scale][slot, :n])
            batch_c[scale][slot, :n] = y.C
            if n < batch_f[scale].shape[1]:
                batch_f[scale][slot, n:].zero_()
            xs_infers[scale].append(y)
        batched = {s: (batch_f[s], batch_c[s]) for s in xs_infers}
        # rows of every subnet at every scale, for the transformer's host-side "is there a padded row" decisions
        batched["_meta"] = {"lens": {"fine": [int(t.F.shape[0]) for t in xs_infers[1]],
                                     "level": {s: [int(t.F.shape[0]) for t in v] for s, v in xs_infers.items()}}}
        if sem_F is None:
            sem_F, sem_C = batch_sparse_tensor(sem_logits_pruneds, pad_to[1])
        else:
            sem_C = batch_c[1]
        # a row counts as padding iff its logits AND its coordinates are all zero (transformer_predictor_v2.py:194-205 through
        # decoder_v3.py:443-446): two reductions instead of six element-wise / reduce launches
        keep_pad = (sem_F != 0).any(-1) | (sem_C != 0).any(-1)
        panop = self.transformer_predictor(batched, (sem_F, sem_C), min_Cs, max_Cs, keep_pad, subnets=subnets,
                                           sem_tensors=sem_logits_pruneds)
        return panop, sem_logits_pruneds

    def forward(self, x, features, global_min_coords, global_max_coords, min_Cs, max_Cs,
                is_predict_panop=True, keep_override=None, subnets=None):
        """features = [enc_s1, enc_s2, enc_s4]; x = bottleneck output at tensor stride 8."""
        assert not self.training, "inference only"
        skips = features[::-1]
        sem_logits_at_scales: Dict[int, list] = {}
        xs: Dict[int, ME.SparseTensor] = {}
        for i, block in enumerate(self.dec_blocks):
            scale = 2 ** (len(self.dec_blocks) - 1 - i)
            x, sem_logits = block(x, skips[i], global_min_coords, global_max_coords)
            keep = self._keep_completion(x, scale, sem_logits, keep_override)
            mgr = x.coordinate_manager
            out_key, rows = mgr.prune(x.coordinate_map_key, keep)
            be = mgr.backend()
            x = ME.SparseTensor(be.gather_rows(x.F, rows), coordinate_map_key=out_key, coordinate_manager=mgr)
            sem_logits = [ME.SparseTensor(be.gather_rows(l.F, rows), coordinate_map_key=out_key,
                                          coordinate_manager=mgr) for l in sem_logits]
            xs[scale] = x
            sem_logits_at_scales[scale] = sem_logits
        ret = {"sem_logits_at_scales": sem_logits_at_scales}
        if is_predict_panop:
            panop, pruned = self.predict_panop(xs, sem_logits_at_scales, min_Cs, max_Cs, keep_override, subnets)
            ret["panop_predictions"] = panop
            ret["sem_logits_pruneds"] = pruned
        return ret
