"""PaSCo network graph: point-feature stage, MIMO input merge, sparse U-Net + mask transformer.

Reference: pasco/models/unet3d_sparse_v2.py:15-86 (`CylinderFeat`), :89-256 (`UNet3DV2`);
pasco/models/augmenter.py:13-27 (`Augmenter.merge`); pasco/models/net_panoptic_sparse.py:40-312,
539-576 (`Net.__init__/forward/step_inference`).  Inference only.  Parameter names follow the
reference state dict (`feat.*`, `unet3d.encoder.*`, `unet3d.dense3d.*`,
`unet3d.decoder_generative.*`, `transformer_predictor.*`).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import me as ME
import os

from ..me.backend import F16RangeError, StatusError, backend_for
from . import fused
from .fused import ACT_RELU
from .bottleneck import SPCDense3Dv2
from .decoder import DecoderGenerativeSepConvV2
from .encoder import Encoder3DSepV2
from .ensemble import Ensembler
from .panoptic import panoptic_inference, panoptic_inference_many
from .transformer import TransformerPredictorV2

# SemanticKITTI "thing" class ids (pasco/data/semantic_kitti/params.py, `thing_ids`)
THING_IDS = (1, 2, 3, 4, 5, 6, 7, 8)


def compute_scene_size(min_coords, max_coords, scale=1):
    """ceil((max - min + 1) / scale) * scale (reference: pasco/models/misc.py:30-32)."""
    return (torch.ceil((max_coords - min_coords + 1) / scale) * scale).int()


def unique_rows_sorted(rows: torch.Tensor):
    """`torch.unique(rows, return_inverse=True, dim=0)` for integer coordinate rows [N, 3] (x,y,z) or [N, 4]
    (b,x,y,z): same sorted unique rows and inverse, via ONE 64-bit key per row (the coordinate-map packing:
    10 bits batch, 18 bits per axis biased by 2^17 - order preserving) and a 1-D sort instead of a row-wise
    lexicographic sort."""
    r = rows.to(torch.int64)
    if r.shape[1] == 3:
        b, xyz = None, r
    else:
        b, xyz = r[:, 0], r[:, 1:]
    bias = 1 << 17
    key = ((xyz[:, 0] + bias) << 36) | ((xyz[:, 1] + bias) << 18) | (xyz[:, 2] + bias)
    if b is not None:
        key = key | (b << 54)
    uk, inv = torch.unique(key, return_inverse=True)
    mask = (1 << 18) - 1
    cols = [((uk >> 36) & mask) - bias, ((uk >> 18) & mask) - bias, (uk & mask) - bias]
    if b is not None:
        cols = [uk >> 54] + cols
    return torch.stack(cols, dim=1).to(rows.dtype), inv


class CylinderFeat(nn.Module):
    """Per-point MLP + max over the points of a voxel (reference unet3d_sparse_v2.py:15-86).

    The reference shuffles the points before a *sorted* unique + scatter-max, which is
    result-neutral, so the shuffle is omitted."""

    def __init__(self, fea_dim=3, out_pt_fea_dim=64):
        super().__init__()
        self.PPmodel = nn.Sequential(
            nn.BatchNorm1d(fea_dim), nn.Linear(fea_dim, 64), nn.BatchNorm1d(64), nn.ReLU(),
            nn.Linear(64, 128), nn.BatchNorm1d(128), nn.ReLU(),
            nn.Linear(128, 256), nn.BatchNorm1d(256), nn.ReLU(),
            nn.Linear(256, out_pt_fea_dim))

    def _mlp(self, fea) -> torch.Tensor:
        """PPmodel; in eval mode each Linear runs as one fused launch with its BatchNorms / ReLU folded in.  `fea` may be the
        LIST of the subnets' point features: the first layer then runs once per subnet and writes its rows of the shared
        hidden operand - the [P, 283] concatenation (430 MB at S10) is never formed."""
        if isinstance(fea, (list, tuple)) and (self.training or len(fea) == 1):
            fea = torch.cat(list(fea), dim=0)
        if self.training:
            return self.PPmodel(fea)
        m = self.PPmodel
        # each hidden activation has one reader (the next Linear): it is stored only as that layer's operand
        if isinstance(fea, (list, tuple)):
            h = hs = None
            dev = fea[0].device
            total = sum(int(f.shape[0]) for f in fea)
            c0, c1 = m[1].in_features, m[1].out_features
            if fused.fusion() and fused._kernel_device(dev) and min(int(f.shape[0]) for f in fea) >= fused.MIN_ROWS_LINEAR:
                from ..me.backend import backend_for as _bf
                emit = fused.conv_precision() == "f16x3" and fused._PRESPLIT and c1 % 32 == 0 and _bf(dev).split_supported(c0, c1)
                if emit:      # rows of the shared hidden operand
                    hs = torch.empty((total, c1 // 32, 2, 32), dtype=torch.float16, device=dev)
                else:         # odd input width (283 = 27 + 256): exact-fp32 launches, rows of the shared fp32 result
                    h = torch.empty((total, c1), dtype=torch.float32, device=dev)
                r0 = 0
                for f in fea:
                    r1 = r0 + int(f.shape[0])
                    fused.linear_bn_act(f, m[1], pro_bn=m[0], epi_bn=m[2], epi_act=ACT_RELU, emit=emit,
                                        emit_into=hs[r0:r1] if emit else None, out=None if emit else h[r0:r1])
                    r0 = r1
            else:
                h, hs = fused.linear_bn_act(torch.cat(list(fea), dim=0), m[1], pro_bn=m[0], epi_bn=m[2], epi_act=ACT_RELU, emit=True)
        else:
            h, hs = fused.linear_bn_act(fea, m[1], pro_bn=m[0], epi_bn=m[2], epi_act=ACT_RELU, emit=True)
        h, hs = fused.linear_bn_act(h, m[4], epi_bn=m[5], epi_act=ACT_RELU, in_split=hs, emit=True)
        h, hs = fused.linear_bn_act(h, m[7], epi_bn=m[8], epi_act=ACT_RELU, in_split=hs, emit=True)
        return fused.linear_bn_act(h, m[10], in_split=hs)

    def forward(self, pt_fea: List[torch.Tensor], xy_ind: List[torch.Tensor]):
        ind = torch.cat([F.pad(c, (1, 0), value=i) for i, c in enumerate(xy_ind)], dim=0)
        fea = torch.cat(pt_fea, dim=0)
        unq, inv = unique_rows_sorted(ind)
        h = self._mlp(fea)
        pooled = torch.full((unq.shape[0], h.shape[1]), float("-inf"), dtype=h.dtype, device=h.device)
        pooled.scatter_reduce_(0, inv[:, None].expand_as(h), h, reduce="amax", include_self=True)
        return unq.to(torch.int64), pooled


def merge_subnet_inputs(in_feat: ME.SparseTensor, n_infers: int) -> ME.SparseTensor:
    """MIMO multiplexing: voxel-wise channel concat of the subnets' inputs on the union of their
    coordinates (batch index i -> channel block i), all-zero rows dropped, batch index 0,
    rows in lexicographic (x,y,z) order.  Same result as the reference's dense round trip
    (augmenter.py:13-27: .dense -> cat channels -> ME.to_sparse) without the [M,64,X,Y,Z] tensor."""
    C = in_feat.C
    Fi = in_feat.F
    c = Fi.shape[1]
    xyz = C[:, 1:]
    uniq, inv = unique_rows_sorted(xyz)                            # sorted rows = lexicographic order
    out = Fi.new_zeros((uniq.shape[0], n_infers * c))
    col = C[:, 0].to(torch.int64) * c
    idx = (inv[:, None] * (n_infers * c) + col[:, None] + torch.arange(c, device=Fi.device)[None, :])
    out.view(-1).index_copy_(0, idx.reshape(-1), Fi.reshape(-1))
    nz = (out != 0).any(dim=1)
    if not bool(nz.all()):
        out, uniq = out[nz], uniq[nz]
    coords = torch.cat([torch.zeros((uniq.shape[0], 1), dtype=torch.int32, device=uniq.device),
                        uniq.to(torch.int32)], dim=1)
    return ME.SparseTensor(out.contiguous(), coords)


class BoundsList(list):
    """The per-subnet box corners (device tensors, as the batch dict holds them) plus `.host`: the same numbers as Python ints,
    read once per forward."""
    host = None


def host_bounds(in_feat, global_min_coords, global_max_coords, min_Cs, max_Cs):
    """-> (min_Cs, max_Cs as BoundsList with .host = [[x, y, z], ...], {"gmin", "gmax", "in_max"}) from ONE device read."""
    dev = in_feat.device
    parts = [torch.as_tensor(global_min_coords).reshape(-1), torch.as_tensor(global_max_coords).reshape(-1)]
    parts += [torch.as_tensor(t).reshape(-1) for t in min_Cs] + [torch.as_tensor(t).reshape(-1) for t in max_Cs]
    parts = [t.to(device=dev, dtype=torch.int64) for t in parts]
    c = in_feat.C
    parts.append(c[:, 1:].max(dim=0)[0].to(torch.int64) if c.shape[0] else torch.zeros(3, dtype=torch.int64, device=dev))
    hv = torch.cat(parts).tolist()
    m = len(min_Cs)
    mn, mx = BoundsList(min_Cs), BoundsList(max_Cs)
    mn.host = [hv[6 + 3 * i: 9 + 3 * i] for i in range(m)]
    mx.host = [hv[6 + 3 * m + 3 * i: 9 + 3 * m + 3 * i] for i in range(m)]
    return mn, mx, {"gmin": hv[0:3], "gmax": hv[3:6], "in_max": hv[6 + 6 * m: 9 + 6 * m]}


class UNet3DV2(nn.Module):
    def __init__(self, in_channels, n_classes, transformer_predictor, n_infers, f_maps, heavy_decoder=True,
                 dense3d_dropout=0.0, decoder_dropouts=(0.0, 0.0, 0.0), encoder_dropouts=(0.0, 0.0, 0.0)):
        super().__init__()
        self.n_infers = n_infers
        self.transformer_predictor = transformer_predictor
        self.encoder = Encoder3DSepV2(in_channels, f_maps, heavy_decoder=heavy_decoder, dropouts=encoder_dropouts)
        self.dense3d = nn.Sequential(SPCDense3Dv2(init_size=f_maps[-1]), nn.Dropout3d(dense3d_dropout))
        self.decoder_generative = DecoderGenerativeSepConvV2(
            f_maps, n_classes=n_classes, transformer_predictor=transformer_predictor, n_infers=n_infers,
            heavy_decoder=heavy_decoder, dropouts=decoder_dropouts)
        self.encoder = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(self.encoder)
        self.decoder_generative = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(self.decoder_generative)

    def dense_bottleneck(self, deepest: ME.SparseTensor, bs, global_min_coords, global_max_coords, host=None):
        """stride-8 features -> dense grid -> SPCDense3Dv2 -> back to a sparse tensor that shares the
        encoder's coordinate manager (unet3d_sparse_v2.py:182-214)."""
        scale = deepest.tensor_stride[0]
        dev = deepest.device
        gmin = global_min_coords.to(dev)
        if host is not None:     # read at the top of the forward; the largest stride-`scale` coordinate = floor of the input's
            gmin_h = host["gmin"]
            gmax_h = [max(a, (b // scale) * scale) for a, b in zip(host["gmax"], host["in_max"])]
        else:
            max_c = deepest.C[:, 1:].max(dim=0)[0].to(torch.int64)
            # ONE host read for everything the host needs here (bounds + the largest stride-8 coordinate)
            hv = torch.cat([gmin.reshape(-1).to(torch.int64), global_max_coords.to(dev).reshape(-1).to(torch.int64), max_c]).tolist()
            gmin_h = hv[0:3]
            gmax_h = [max(a, b) for a, b in zip(hv[3:6], hv[6:9])]
        size = [-(-(mx - mn + 1) // scale) for mn, mx in zip(gmin_h, gmax_h)]      # compute_scene_size(...) // scale
        # channels-last rows of the dense grid (sites in lexicographic order = ME.to_sparse order);
        # the reference goes sparse -> dense [1,C,X,Y,Z] -> Conv3d stack -> ME.to_sparse, this is the
        # same computation without leaving the row layout.
        dims = (bs, *[int(v) for v in size])
        nsites = dims[0] * dims[1] * dims[2] * dims[3]
        c = deepest.F.shape[1]
        site = torch.div(deepest.C[:, 1:].to(torch.int64) - gmin.to(torch.int64).reshape(1, 3), scale,
                         rounding_mode="floor")
        inside = (site >= 0).all(dim=1) & (site[:, 0] < dims[1]) & (site[:, 1] < dims[2]) & (site[:, 2] < dims[3])
        lin = ((deepest.C[:, 0].to(torch.int64) * dims[1] + site[:, 0]) * dims[2] + site[:, 1]) * dims[3] + site[:, 2]
        rows = deepest.F.new_zeros((nsites + 1, c))            # one spare row takes the rows outside the grid (no host read)
        rows.index_copy_(0, torch.where(inside, lin, torch.full_like(lin, nsites)), deepest.F)
        rows = rows[:nsites]
        dense3d, dropout = self.dense3d[0], self.dense3d[1]
        assert not dropout.training
        out = dense3d.forward_rows(rows, dims)
        site_coords = dense3d._grid_tables(dims, out.device)[4]       # lexicographic (b, x, y, z): what forward_rows returns
        nz = (out != 0).any(dim=1)                     # ME.to_sparse drops all-zero sites
        opt = fused.optimistic_word(dev)
        if opt is not None:
            # every site of the bottleneck's output is normally non-zero (biases, BatchNorm shifts): keep them all and leave
            # the verification to the end-of-step check instead of reading it here
            opt.bitwise_or_((~nz).any().to(torch.int32))
        elif not bool(nz.all()):
            out, site_coords = out[nz].contiguous(), site_coords[nz]
        coords = site_coords.clone()
        coords[:, 1:] = coords[:, 1:] * scale + gmin.reshape(1, -1).to(coords.dtype)
        mgr = deepest.coordinate_manager
        key = mgr.insert_unique(coords.to(torch.int32).contiguous(), scale)   # grid sites are distinct: no dedup pass, no count read
        return ME.SparseTensor(out, coordinate_map_key=key, coordinate_manager=mgr)

    def forward(self, in_feat, bs, global_min_coords, global_max_coords, min_Cs, max_Cs,
                is_predict_panop=True, keep_override=None, subnets=None):
        assert not self.training, "inference only"
        # ONE host read for every small quantity the host decides with (scene bounds, subnet boxes, the largest input
        # coordinate): the bottleneck's grid extent and the attention mask's "padded row outside its subnet's box" test
        # need no device read of their own later
        min_Cs, max_Cs, hb = host_bounds(in_feat, global_min_coords, global_max_coords, min_Cs, max_Cs)
        feats = self.encoder(in_feat)
        deepest = self.dense_bottleneck(feats[-1], bs, global_min_coords, global_max_coords, host=hb)
        return self.decoder_generative(deepest, feats[:-1], global_min_coords, global_max_coords, min_Cs, max_Cs,
                                       is_predict_panop=is_predict_panop, keep_override=keep_override,
                                       subnets=subnets)


class PascoNet(nn.Module):
    """Inference graph of the reference `Net` (net_panoptic_sparse.py:40-312): `feat` (point MLP),
    MIMO merge, `unet3d` (+ `transformer_predictor`).  Ensembling / panoptic post-processing are the
    next rows of SURVEY.md 8(f)."""

    def __init__(self, n_classes=20, n_infers=1, in_channels=27 + 256, f=64, num_queries=100, heavy_decoder=True,
                 encoder_dropouts=(0.0, 0.0, 0.0), decoder_dropouts=(0.0, 0.0, 0.0), dense3d_dropout=0.0,
                 iou_threshold=0.2, overlap_threshold=0.4, object_mask_threshold=0.7, thing_ids=THING_IDS,
                 hidden_dim=384, dim_feedforward=1024):
        super().__init__()
        self.n_infers = n_infers
        self.n_classes = n_classes
        self.transformer_predictor = TransformerPredictorV2(
            in_channels=[f * 4, f * 2, f], num_classes=n_classes, hidden_dim=hidden_dim, num_queries=num_queries,
            nheads=8, dim_feedforward=dim_feedforward, mask_dim=f, n_infers=n_infers)   # reference: 384 / 1024 fixed
        self.unet3d = UNet3DV2(in_channels=f * n_infers, n_classes=n_classes,
                               transformer_predictor=self.transformer_predictor, n_infers=n_infers,
                               f_maps=[f, f * 2, f * 4, f * 4], heavy_decoder=heavy_decoder,
                               dense3d_dropout=dense3d_dropout, decoder_dropouts=decoder_dropouts,
                               encoder_dropouts=encoder_dropouts)
        self.feat = CylinderFeat(fea_dim=in_channels, out_pt_fea_dim=f)
        self.ensembler = Ensembler()
        self.iou_threshold = iou_threshold
        self.overlap_threshold = overlap_threshold
        self.object_mask_threshold = object_mask_threshold
        self.thing_ids = tuple(thing_ids)

    def prepare_input(self, in_feats: List[torch.Tensor], in_coords: List[torch.Tensor], fused_stage: bool = True) -> ME.SparseTensor:
        """`self.feat` + `ME.SparseTensor` + `Augmenter.merge` (net_panoptic_sparse.py:548-550).  On a device with a backend
        the voxel max and the merge are one sort-free pass over the points (`CBackend.pooled_merge`: no per-subnet voxel
        tensor, no unique / sort, no dense detour); `fused_stage=False` or PASCO_INPUT_FUSED=0 runs the reference's
        sequence of steps (CylinderFeat.forward, SparseTensor, merge)."""
        x = self._prepare_input_fused(in_feats, in_coords) if fused_stage else None
        if x is None:
            coords, feats = self.feat(in_feats, in_coords)
            x = ME.SparseTensor(feats, coords.int())
            x = merge_subnet_inputs(x, self.n_infers)
        # the point MLP runs on the split-precision kernel too: should its range flag turn up in `forward`, the input stage is
        # redone on the exact path together with the rest (the raw inputs are only referenced, not copied)
        x.__dict__["_ph_source"] = (in_feats, in_coords)
        return x

    def _prepare_input_fused(self, in_feats, in_coords):
        dev = in_feats[0].device
        if self.training or os.environ.get("PASCO_INPUT_FUSED", "1") == "0" or len(in_feats) != self.n_infers or \
                not 1 <= self.n_infers <= 8:
            return None
        try:
            be = backend_for(dev)             # the GPU library, or (tests) a registered CPU checker
        except RuntimeError:
            return None
        if not be.has("cells_max"):
            return None
        starts = [0]
        for c in in_coords:
            starts.append(starts[-1] + int(c.shape[0]))
        if starts[-1] == 0:
            return None
        xyz = torch.cat([c if c.dtype == torch.int64 else c.to(torch.int64) for c in in_coords], dim=0).contiguous()
        h = self.feat._mlp(list(in_feats)).contiguous()
        if h.shape[1] % 4 != 0:
            return None
        got = be.pooled_merge(h, xyz, starts)
        if got is None:
            return None
        coords, feats = got
        mgr = ME.CoordinateManager(D=3, device=dev)
        key = mgr.insert_unique(coords, 1)              # occupied sites are distinct by construction: no dedup pass
        return ME.SparseTensor(feats, coordinate_map_key=key, coordinate_manager=mgr)

    def forward(self, in_feat: ME.SparseTensor, global_min_coords, global_max_coords, min_Cs, max_Cs,
                is_predict_panop=True, keep_override=None, subnets=None):
        """The reference's timed window: `self.unet3d(...)` (net_panoptic_sparse.py:228-250)."""
        be = backend_for(in_feat.device)
        run = lambda: self.unet3d(in_feat, 1, global_min_coords, global_max_coords, min_Cs, max_Cs,
                                  is_predict_panop=is_predict_panop, keep_override=keep_override, subnets=subnets)
        try:
            # this function ends the step with the status check and can redo it: the graph may take its optimistic shortcuts
            with fused.optimistic_override(os.environ.get("PASCO_OPTIMISTIC", "1") != "0"):
                ret = run()
            be.check_status(in_feat.device)     # flags of this stream's launches (f16 range, coordinate range, table clamp)
        except StatusError as err:
            if err.bits & ~(1 | 8 | 16):
                raise                           # a coordinate no map / table can hold: nothing to redo
            # bit 0: an activation left the f16 range of the split-precision operands (|x| > 2047 with the 2^5 operand scale)
            # -> every product on the exact fp32 MFMA; bit 3: the fused input stage met an all-zero merged row -> the
            # reference's sequence of steps for the input stage.  The step is redone instead of failing - slower, same graph -
            # and counted so that a serving loop can see it happen.
            if err.bits & 1:
                self.range_fallbacks = getattr(self, "range_fallbacks", 0) + 1
            if err.bits & 8:
                self.input_fallbacks = getattr(self, "input_fallbacks", 0) + 1
            if err.bits & 16:                   # an optimistic shortcut did not hold: the checked paths serve this scene
                self.optimistic_fallbacks = getattr(self, "optimistic_fallbacks", 0) + 1
            import contextlib
            with (fused.precision_override("f32") if err.bits & 1 else contextlib.nullcontext()), \
                    (fused.optimistic_override(False) if err.bits & 16 else contextlib.nullcontext()):
                src = in_feat.__dict__.get("_ph_source")
                if src is not None:             # made by prepare_input: its point MLP / merge may be what raised the flag
                    in_feat = self.prepare_input(*src, fused_stage=not (err.bits & 8))
                elif err.bits & 8:
                    raise
                ret = run()
            be.check_status(in_feat.device)
        return ret

    def ensemble(self, ret, Ts):
        """`Net.forward(return_ensemble=True)` after the U-Net (net_panoptic_sparse.py:252-310):
        -> (ssc_confidences, sem_prob_denses, panop_prob_predictions); confidence = max class prob."""
        cache = {}
        sem_prob_denses = self.ensembler.ensemble_sem_compl(ret["sem_logits_at_scales"], Ts, cache=cache)
        panop = self.ensembler.ensemble_panop(ret["panop_predictions"], sem_prob_denses, Ts,
                                              iou_threshold=self.iou_threshold, cache=cache)
        # max over classes on the channels-last rows the dense views are made of (contiguous reduction)
        X, Y, Z = self.ensembler.scene_size
        if cache.get("sem_conf") is not None:      # written by the same pass that made the probabilities
            ssc_confidences = [r.reshape(X, Y, Z) for r in cache["sem_conf"]]
        else:
            ssc_confidences = [r.max(dim=1)[0].reshape(X, Y, Z) for r in cache["sem_rows"]]
        dev = ssc_confidences[0].device
        backend_for(dev).check_status(dev)      # the ensembler's lazy map builds report into the same stream's word
        return ssc_confidences, sem_prob_denses, panop

    def step_inference(self, in_feats, in_coords, Ts, global_min_coords, global_max_coords, min_Cs, max_Cs,
                       keep_override=None, eval_list=None):
        """`Net.step_inference` (net_panoptic_sparse.py:539-608) without the metric bookkeeping: point MLP,
        merge, U-Net + transformer, ensembling, panoptic post-processing of every subnet + the ensemble."""
        x = self.prepare_input(in_feats, in_coords)
        ret = self(x, global_min_coords, global_max_coords, min_Cs, max_Cs, keep_override=keep_override)
        ssc_conf, sem_probs, panop = self.ensemble(ret, Ts)
        outs = self.panoptic(panop, ssc_conf, eval_list)
        return outs, sem_probs, panop

    def panoptic(self, panop, ssc_conf=None, eval_list=None):
        """`panoptic_inference` of the M subnets' outputs and the ensemble's (net_panoptic_sparse.py:578-608): every output's
        launches first, then one device->host copy for all their segment tables."""
        ids = list(range(len(panop)) if eval_list is None else eval_list)
        dev = panop[0]["voxel_probs"].F.device
        outs = panoptic_inference_many([(panop[i]["voxel_probs"], panop[i]["query_probs"]) for i in ids],
                                       overlap_threshold=self.overlap_threshold,
                                       object_mask_threshold=self.object_mask_threshold, thing_ids=self.thing_ids,
                                       scene_size=self.ensembler.scene_size,
                                       min_C=torch.zeros(3, dtype=torch.int32, device=dev),
                                       input_query_logit=False, input_voxel_logit=False)
        if ssc_conf is not None:
            for i, o in zip(ids, outs):
                o["ssc_confidence"] = ssc_conf[i]
        return outs
