"""Subnet ensembling on the canonical grid (reference: pasco/models/ensembler.py:20-131,159-187,
pasco/models/transform_utils.py:60-74,95-117,160-181, pasco/models/utils.py:153-198,
pasco/models/misc.py:46-57).

The reference resamples every subnet's per-voxel outputs onto the canonical 256x256x32 grid through
dense tensors ([20,256,256,32] per subnet for the semantic probabilities, [100,256,256,32] = 838 MB per
subnet for the mask probabilities) and `grid_sample(nearest)`.  Restated sparsely, same results:

  * the canonical voxel centres are pushed through T_i once (`project_canonical`, the reference's
    `transform`: metres, voxel 0.2, origin (0,-25.6,-2), round) and looked up in the subnet's coordinate
    hash map - "nearest sample with zero padding" of an integer coordinate is exactly a hash lookup;
  * mask probabilities live on the compacted union of occupied canonical sites [U, Q] (U ~ 10 % of the
    grid); soft-IoU matching is the same [Q,U] x [U,Q] product, the Hungarian step runs on the host
    with scipy as in the reference (utils.py:191);
  * `ME.to_sparse` of a dense tensor whose occupied sites are known is an order-preserving compaction.
Dense [C,256,256,32] tensors are produced only where they are the returned format (semantic
probabilities), as a view of channels-last rows.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from .. import me as ME
from ..me.backend import backend_for

CANONICAL_SIZE = (256, 256, 32)
RESOLUTION = 0.2
MIN_BOUND = (0.0, -25.6, -2.0)


def canonical_sites(size: Sequence[int], device) -> torch.Tensor:
    """int32 [X*Y*Z, 3] voxel indices of the canonical grid, lexicographic (x, y, z)."""
    ax = [torch.arange(n, dtype=torch.int32, device=device) for n in size]
    return torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1).reshape(-1, 3)


_MB_CACHE = {}


def _min_bound(device) -> torch.Tensor:
    key = str(device)
    if key not in _MB_CACHE:
        _MB_CACHE[key] = torch.tensor(MIN_BOUND, dtype=torch.float32, device=device)
    return _MB_CACHE[key]


def project_canonical(sites: torch.Tensor, T: torch.Tensor, resolution: float = RESOLUTION) -> torch.Tensor:
    """Voxel index -> voxel centre in metres -> T -> voxel index (round half to even), fp32 like the
    reference's `transform` (transform_utils.py:60-74).  The 4x4 product is written out with a fixed
    operation order so that CPU and GPU give bit-identical coordinates."""
    T = T.to(device=sites.device, dtype=torch.float32)
    mb = _min_bound(sites.device)
    # float64 until the cast, as in the reference (numpy grid is float64, min_bound float32 -> promoted)
    p = (sites.to(torch.float64) * resolution + resolution / 2 + mb.to(torch.float64)).to(torch.float32)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    out = []
    for r in range(3):
        v = T[r, 0] * x + T[r, 1] * y
        v = v + T[r, 2] * z
        v = v + T[r, 3]
        out.append(v)
    q = torch.stack(out, dim=1)
    q = (q - mb - resolution / 2) / resolution
    return torch.round(q).to(torch.int32)


def _lookup_rows(st: ME.SparseTensor, coords4: torch.Tensor) -> torch.Tensor:
    """row of each (0, x, y, z) coordinate (int32 [N, 4], batch column 0) in st's coordinate map, or -1."""
    return st.coordinate_manager.find(st.coordinate_map_key, coords4)


def _gram(a: torch.Tensor, b: torch.Tensor, chunks: int = 256) -> torch.Tensor:
    """a^T b for tall [U, Q] operands.  A [Q, U] x [U, Q] GEMM has a 100 x 100 output and a 440 k long
    reduction: one library call runs on a handful of workgroups (0.65 ms measured); cutting U into `chunks`
    slabs makes it a batched GEMM over the slabs plus a sum (same products, reassociated)."""
    U = a.shape[0]
    per = U // chunks
    if not a.is_cuda or per < 64:
        return a.t() @ b
    main = per * chunks
    out = torch.bmm(a[:main].view(chunks, per, -1).transpose(1, 2), b[:main].view(chunks, per, -1)).sum(0)
    if main < U:
        out = out + a[main:].t() @ b[main:]
    return out


GRAM_SLABS = 8       # config C4 (dist.site_sharded_ensemble) cuts the union rows into this many contiguous slabs owned by the
                     # ranks; every reduction over the rows (the matching's Gram matrix and column sums) is formed per slab and
                     # the slab results are added in slab order.  The order is part of the result: `Ensembler(gram_slabs=8)`
                     # adds the same partial results in the same order in ONE process -> bit-identical to the sharded run.
                     # The default single-GPU ensembler keeps one slab (one batched GEMM per matching: 8 slabs cost 8 x the
                     # launches, +1.0 ms per step measured in profiles/r5z_bench_kernel_stats.csv's predecessor).


def slab_bounds(n_rows: int, slabs: int = GRAM_SLABS) -> List[int]:
    """Row boundaries of the slabs: slab k = rows [b[k], b[k + 1])."""
    return [(k * n_rows) // slabs for k in range(slabs + 1)]


def match_partials(anchor_mask: torch.Tensor, aux_mask: torch.Tensor) -> torch.Tensor:
    """What ONE slab of rows contributes to the soft-IoU matching: [Q * Q + 2 Q] = anchor^T aux | column sums of the
    anchor | column sums of the aux masks (fp32, rows [n, Q] each)."""
    q = anchor_mask.shape[1]
    if anchor_mask.shape[0] == 0:
        return anchor_mask.new_zeros(q * q + 2 * q)
    return torch.cat([_gram(anchor_mask, aux_mask).reshape(-1), anchor_mask.sum(0), aux_mask.sum(0)])


def match_from_partials(partials: Sequence[torch.Tensor], q: int, iou_threshold: float):
    """Slab contributions (in slab order) -> the Hungarian matching: (a_idx, b_idx on the partials' device, matched IoUs
    on the host) (utils.py:153-198)."""
    if len(partials) == 1:
        tot = partials[0]
    else:
        tot = partials[0].clone()
        for p in partials[1:]:
            tot += p                                                         # fixed order: slab 0, 1, 2, ...
    dev = tot.device
    # the one host read of the matching step brings the SUMS down (Q Q + 2 Q floats); the Q x Q soft IoUs are formed on the host:
    # fp32 +, -, / are correctly rounded on either side - the same matrix as the device's, bit for bit, without nine small launches
    tot_h = tot.cpu().numpy()
    inter = tot_h[:q * q].reshape(q, q)
    union = (tot_h[q * q:q * q + q][:, None] + tot_h[q * q + q:][None, :]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.where(union != 0, inter / union, np.float32(0.0)).astype(np.float32)
    iou = iou * (iou > np.float32(iou_threshold))
    a_idx, b_idx = linear_sum_assignment(np.float32(1.0) - iou)
    matched_h = torch.from_numpy(iou[a_idx, b_idx])                          # host copy: the query filter is decided there too
    ab = torch.as_tensor(np.stack([a_idx, b_idx]))
    if dev.type == "cuda":                                                   # pinned staging: the upload does not synchronise
        ab = ab.pin_memory().to(dev, non_blocking=True)
    return ab[0], ab[1], matched_h


ENS_KERNEL_MAX_Q, ENS_KERNEL_MAX_C = 128, 64     # shapes the ph_ens_* row kernels take (one wave64 per row, two columns a lane)


def _ens_resample_torch(logits, rows, sel):
    """ph_ens_resample in torch (query counts the row kernels do not take): sigmoid(logits[rows[sel]]), zero rows where
    the site has no voxel; flag = the row has a non-zero entry."""
    r = rows.index_select(0, sel.long())
    out = torch.sigmoid(logits.index_select(0, r.clamp(min=0).long()))
    out = torch.where((r >= 0)[:, None], out, torch.zeros_like(out))
    return out, (out != 0).any(dim=1).to(torch.uint8)


def _ens_merge_torch(anchor, m, perm, i):
    anchor.copy_((anchor * i + m.index_select(1, perm.long())) / (i + 1))
    return anchor


def _ens_finish_torch(anchor, keep, sem, sel):
    nonempty = sem.index_select(0, sel.long()).argmax(dim=1) != 0
    out = anchor.index_select(1, keep.long()) * nonempty[:, None].to(anchor.dtype)
    return out, (out != 0).any(dim=1).to(torch.uint8)


class Ensembler(torch.nn.Module):
    def __init__(self, scene_size=CANONICAL_SIZE, gram_slabs: int = 1):
        super().__init__()
        self.scene_size = tuple(scene_size)
        self.gram_slabs = int(gram_slabs)      # 1 = one reduction over all union rows; GRAM_SLABS = the order of config C4
        self._sites = {}

    def projected(self, T: torch.Tensor, device, cache: dict) -> torch.Tensor:
        """canonical sites seen through T as (0, x, y, z) rows (shared by the semantic and the mask resampling
        of a scene)."""
        key = id(T)
        if key not in cache:
            be = backend_for(device)
            if be.has("project_canonical"):
                # one kernel instead of ~20 elementwise passes over [2 M, 3] tensors (same arithmetic, same order)
                cache[key] = be.project_canonical(T.to(device), self.scene_size, RESOLUTION, MIN_BOUND)
            else:
                p = project_canonical(self.sites(device), T)
                cache[key] = torch.cat([torch.zeros((p.shape[0], 1), dtype=torch.int32, device=p.device), p], dim=1).contiguous()
        return cache[key]

    def sites(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._sites:
            self._sites[key] = canonical_sites(self.scene_size, device)
        return self._sites[key]

    # -- a21 -----------------------------------------------------------------------------------------
    def ensemble_sem_compl(self, sem_logits_at_scales: Dict[int, List[ME.SparseTensor]], Ts,
                           cache: dict = None) -> List[torch.Tensor]:
        """-> list of [n_classes, X, Y, Z] probabilities, one per subnet + their mean (ensembler.py:159-187).
        Sites a subnet does not cover get class-0 probability 1."""
        logits_1 = sem_logits_at_scales[1]
        dev = logits_1[0].device
        cache = {} if cache is None else cache
        X, Y, Z = self.scene_size
        be = backend_for(dev)
        c = logits_1[0].F.shape[1]
        if c in (19, 20) and len(logits_1) <= 8 and (dev.type == "cuda" or be.has("sem_ensemble")):
            # softmax, resampling, class-0 fill, mean and the confidence maps in ONE pass (ph_sem_ensemble)
            rows = [_lookup_rows(st, self.projected(Ts[i], dev, cache)) for i, st in enumerate(logits_1)]
            outs, confs = be.sem_ensemble([st.F.contiguous() for st in logits_1], rows)
            cache["sem_rows"] = outs
            cache["sem_conf"] = confs
            return [o.reshape(X, Y, Z, -1).permute(3, 0, 1, 2) for o in outs]
        outs = []
        for i, st in enumerate(logits_1):
            probs = F.softmax(st.F, dim=-1)
            rows = _lookup_rows(st, self.projected(Ts[i], dev, cache))
            dense_rows = backend_for(dev).gather_rows(probs.contiguous(), rows)          # [XYZ, C], -1 -> 0
            # the reference tests `probs.sum(channels) == 0` on the resampled grid; a covered site holds a softmax
            # row (sum 1), so that is exactly "no source voxel"
            dense_rows[:, 0] = torch.where(rows < 0, torch.ones_like(dense_rows[:, 0]), dense_rows[:, 0])
            outs.append(dense_rows)
        outs.append(torch.stack(outs, dim=0).mean(0))
        if cache is not None:
            cache["sem_rows"] = outs          # channels-last rows [XYZ, C] of the returned dense views
        return [o.reshape(X, Y, Z, -1).permute(3, 0, 1, 2) for o in outs]

    # -- a22 -----------------------------------------------------------------------------------------
    @staticmethod
    def match_queries(anchor_mask: torch.Tensor, aux_mask: torch.Tensor, iou_threshold: float, slabs: int = 1):
        """Soft-IoU Hungarian matching of query masks given as [U, Q] site rows (utils.py:153-198).  `slabs` > 1: the sums
        over the rows are formed slab by slab and added in slab order (what the site-sharded run of config C4 does)."""
        b = slab_bounds(anchor_mask.shape[0], slabs)
        parts = [match_partials(anchor_mask[b[k]:b[k + 1]], aux_mask[b[k]:b[k + 1]]) for k in range(slabs)]
        return match_from_partials(parts, anchor_mask.shape[1], iou_threshold)

    def ensemble_panop(self, panop_predictions, ensemble_sem_prob_denses, Ts, iou_threshold=0.2, cache: dict = None):
        """-> one dict per subnet + the ensemble: {"sem_probs", "voxel_probs" (SparseTensors on the
        canonical grid), "query_probs"} (ensembler.py:20-131).
        The [U, Q] passes run as three fused row kernels (ph_ens_resample / ph_ens_merge / ph_ens_finish): sigmoid +
        resampling + "row is non-zero" flag; running mean of the matched masks; matched-IoU column filter + empty-class
        zeroing + flag."""
        n_sub = len(panop_predictions)
        dev = panop_predictions[0]["query_logits"].device
        be = backend_for(dev)
        sites = self.sites(dev)
        cache = {} if cache is None else cache
        rows_per_subnet, query_probs = [], []
        for i in range(n_sub):
            vl = panop_predictions[i]["voxel_logits"]
            rows_per_subnet.append(_lookup_rows(vl, self.projected(Ts[i], dev, cache)))
            query_probs.append(F.softmax(panop_predictions[i]["query_logits"], dim=-1))
        # a canonical site is occupied where some subnet has a row for it (row >= 0): one pass over the lookups where the library
        # serves it (ph_keep_mask on int32 sources), else the element-wise form
        if be.has("keep_mask") and n_sub <= 8:
            occupied = be.keep_mask([r.contiguous() for r in rows_per_subnet])
        else:
            occupied = None
            for rows in rows_per_subnet:
                occupied = (rows >= 0) if occupied is None else (occupied | (rows >= 0))
        union_sites = be.mask_compact(occupied.contiguous())                # canonical site ids (int32), lexicographic
        union_long = union_sites.long()
        site_coords = sites.index_select(0, union_long)                     # [U, 3]
        masks, flags = [], []                                               # per subnet [U, Q] (0 where absent), [U] uint8
        # the row kernels serve <= 128 queries and <= 64 classes (the reference's 100 / 20); a checkpoint with more runs
        # the same steps as torch passes instead of failing (num_queries comes from the checkpoint's hyper-parameters)
        nq = panop_predictions[0]["voxel_logits"].F.shape[1]
        rowk = be.has("ens_resample") and nq <= ENS_KERNEL_MAX_Q and ensemble_sem_prob_denses[-1].shape[0] <= ENS_KERNEL_MAX_C
        ens_resample = be.ens_resample if rowk else _ens_resample_torch
        ens_merge = be.ens_merge if rowk else _ens_merge_torch
        ens_finish = be.ens_finish if rowk else _ens_finish_torch
        for i in range(n_sub):
            m, fl = ens_resample(panop_predictions[i]["voxel_logits"].F.contiguous(), rows_per_subnet[i].contiguous(),
                                    union_sites)
            masks.append(m)
            flags.append(fl)
        anchor_q = query_probs[0].clone()
        anchor_m = masks[0].clone() if n_sub > 1 else masks[0]
        ious = []
        for i in range(1, n_sub):
            if self.gram_slabs > 1:
                a_idx, b_idx, iou = self.match_queries(anchor_m, masks[i], iou_threshold, self.gram_slabs)
            else:
                a_idx, b_idx, iou = self.match_queries(anchor_m, masks[i], iou_threshold)
            # the assignment of a square cost matrix lists every anchor query once, in order (a_idx = 0..Q-1)
            anchor_q = (anchor_q * i + query_probs[i][:, b_idx, :]) / (i + 1)
            ens_merge(anchor_m, masks[i], b_idx.to(torch.int32).contiguous(), i)
            ious.append(iou)
        Q = anchor_m.shape[1]
        if ious:       # matched IoUs are host tensors (they came down with the cost matrix): the filter costs no device read
            keep_h = torch.stack(ious, dim=0).mean(0) > iou_threshold
            keep_cols = keep_h.nonzero().reshape(-1)
            if dev.type == "cuda":
                keep_cols = keep_cols.pin_memory().to(dev, non_blocking=True)
            anchor_q = anchor_q.index_select(1, keep_cols)
        else:
            keep_cols = torch.arange(Q, device=dev)
        # zero the ensemble where the ensembled semantic class is "empty"
        sem_rows = cache.get("sem_rows")
        if sem_rows is not None and (len(sem_rows) != len(ensemble_sem_prob_denses) or any(
                r.data_ptr() != d.data_ptr() for r, d in zip(sem_rows, ensemble_sem_prob_denses))):
            sem_rows = None                      # denses did not come from this cache's ensemble_sem_compl
        if sem_rows is None:                     # channels-last rows of the dense [C, X, Y, Z] tensors
            sem_rows = [d.permute(1, 2, 3, 0).reshape(-1, d.shape[0]).contiguous() for d in ensemble_sem_prob_denses]
        ens_m, ens_flag = ens_finish(anchor_m, keep_cols.to(torch.int32).contiguous(), sem_rows[-1].contiguous(), union_sites)
        masks.append(ens_m)
        flags.append(ens_flag)
        query_probs.append(anchor_q)
        out = []
        coords4 = torch.cat([torch.zeros((site_coords.shape[0], 1), dtype=torch.int32, device=dev), site_coords], dim=1)
        nz_rows = be.mask_compact_many(flags)                                 # one host read for all outputs' row counts
        for i, m in enumerate(masks):
            nz32 = nz_rows[i]                                                 # ME.to_sparse keeps non-zero sites
            c = be.gather_rows(coords4, nz32)
            mgr = ME.CoordinateManager(D=3, device=dev)
            key = mgr.insert_unique(c, 1)                                     # canonical sites are unique
            vf = be.gather_rows(m, nz32) if m.shape[1] > 0 else m.new_zeros((nz32.shape[0], 0))   # no query survived
            voxel_prob = ME.SparseTensor(vf, coordinate_map_key=key, coordinate_manager=mgr)
            sem_f = be.gather_rows(sem_rows[i], be.gather_rows(union_sites.reshape(-1, 1), nz32).reshape(-1))
            sem_prob = ME.SparseTensor(sem_f, coordinate_map_key=key, coordinate_manager=mgr)
            out.append({"sem_probs": sem_prob, "voxel_probs": voxel_prob, "query_probs": query_probs[i]})
        return out
