"""Panoptic post-processing (reference: pasco/models/helper.py:91-303 `panoptic_inference`).

Queries with a non-empty, non-dustbin class and probability > object_mask_threshold compete per voxel
(argmax of query prob x mask prob); a query keeps its voxels if enough of its own mask
(mask prob >= vox_occ_threshold) survived the competition; stuff segments of one class merge.

The reference walks the kept queries in a Python loop with ~6 `.item()` host syncs per query and returns dense
[X, Y, Z] maps plus a [K, X, Y, Z] tensor of all kept masks (838 MB at K = 100) for every output.

Device path (`panoptic_inference_many`, round 5; include/pasco_hip.h `panop_*`, csrc/panop.hip): per output three launches
on the SPARSE rows - classify the queries, one pass over the [N, Q] mask probabilities (winner per voxel + per-query areas),
one element-wise pass that replays the reference's sequential walk over the kept queries on the device and writes the
per-voxel results - no host read in between; the segment tables of ALL outputs of a step (the M subnets + the ensemble,
net_panoptic_sparse.py:578-608) come back in ONE device->host copy.  The dense maps are built when somebody asks for them
(`PanopticResult.__missing__`), not per step.

Torch path (`_panoptic_inference_torch`: tensors no backend serves, batches of more than one scene): per-query areas from
one pass over the voxels (bincount), the segment-id bookkeeping on the host on those counts, per-voxel outputs as one gather
through per-query tables.  Both give the reference's results, including its quirk that voxels of a *merged* stuff segment
keep semantic class 0.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn.functional as F

from .. import me as ME


def to_dense(values: torch.Tensor, coords: torch.Tensor, scene_size, min_coords=None) -> torch.Tensor:
    """[N, C] rows -> [C, X, Y, Z] (reference: pasco/models/misc.py:46-57)."""
    c = coords[:, 1:].long() if coords.shape[1] == 4 else coords.long()
    if min_coords is not None:
        c = c - torch.as_tensor(min_coords, device=c.device).long()
    out = torch.zeros((values.shape[1], int(scene_size[0]), int(scene_size[1]), int(scene_size[2])),
                      dtype=values.dtype, device=values.device)
    out[:, c[:, 0], c[:, 1], c[:, 2]] = values.t()
    return out


DENSE_KEYS = ("panoptic_seg_denses", "semantic_seg_denses", "ins_uncertainty_denses", "vox_confidence_denses",
              "vox_uncertainty_denses")


class PanopticResult(dict):
    """What `panoptic_inference` returns (helper.py:291-303), with the sparse rows computed and the dense maps made on first
    access: `res["semantic_seg_denses"]` etc. [bs, X, Y, Z], `res["vox_all_mask_probs_denses"]` list of [K, X, Y, Z].  Extra
    keys of the device path: "semantic_seg_sparses", "ins_uncertainty_sparses", "vox_confidence_sparses",
    "vox_uncertainty_sparses" (lists of [N] rows, like the reference's "panoptic_seg_sparses")."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._lazy = None      # (coords, scene_size, min_C, rows dict, masks, K)

    def __missing__(self, key):
        if self._lazy is None or key not in DENSE_KEYS + ("vox_all_mask_probs_denses",):
            raise KeyError(key)
        coords, scene_size, min_C, rows, masks, kept = self._lazy
        dense = lambda v: to_dense(v.unsqueeze(-1) if v.dim() == 1 else v, coords, scene_size, min_C).squeeze()
        if key == "vox_all_mask_probs_denses":
            kept_t = torch.as_tensor(kept, dtype=torch.long, device=masks.device)
            full = (rows["semantic"] != 0)[:, None]          # a written class is never 0 (class 0 queries are not kept)
            val = [dense(torch.where(full, masks[:, kept_t], torch.zeros((), dtype=masks.dtype, device=masks.device)))]
        else:
            src = {"panoptic_seg_denses": "panoptic", "semantic_seg_denses": "semantic", "ins_uncertainty_denses": "ins_unc",
                   "vox_confidence_denses": "vox_conf", "vox_uncertainty_denses": "vox_unc"}[key]
            val = torch.stack([dense(rows[src])])
        self[key] = val
        return val

    # the lazy keys behave like the real ones of the dict the reference returns (helper.py:291-303): membership, get, iteration
    def _lazy_keys(self):
        return () if self._lazy is None else tuple(k for k in DENSE_KEYS + ("vox_all_mask_probs_denses",) if not dict.__contains__(self, k))

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy_keys()

    def get(self, key, default=None):
        return self[key] if key in self else default

    def materialize(self) -> "PanopticResult":
        """Build every dense map now (what the reference's function always does)."""
        for k in self._lazy_keys():
            self[k]
        return self

    def keys(self):
        return list(dict.keys(self)) + list(self._lazy_keys())

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return dict.__len__(self) + len(self._lazy_keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __reduce__(self):          # pickling / copying: a plain dict with everything built
        return (dict, (dict(self.materialize().items()),))


def _device_backend(t: torch.Tensor):
    import os
    if os.environ.get("PASCO_PANOPTIC_KERNEL", "1") == "0":
        return None
    from ..me.backend import backend_for
    try:
        be = backend_for(t.device)
    except RuntimeError:
        return None
    return be if be.has("panop_write") else None


def panoptic_inference_many(outputs, overlap_threshold: float, object_mask_threshold: float, thing_ids: Sequence[int], min_C,
                            scene_size, input_query_logit: bool = True, input_voxel_logit: bool = False,
                            vox_occ_threshold: float = 0.3) -> List[dict]:
    """`panoptic_inference` for several (voxel_output, query_output) pairs - the M subnets' outputs and the ensemble's of one
    step (net_panoptic_sparse.py:578-608) - with ONE device->host copy for all their segment tables."""
    outputs = list(outputs)
    be = _device_backend(outputs[0][0].F) if outputs else None
    torch_form = lambda v, q: _panoptic_inference_torch(v, q, overlap_threshold, object_mask_threshold, thing_ids, min_C,
                                                        scene_size, input_query_logit, input_voxel_logit, vox_occ_threshold)
    if be is None:
        return [torch_form(v, q) for v, q in outputs]
    # the row kernels take one scene, 1 .. 128 queries, <= 64 classes; anything else (a batch of scenes, an ensemble whose
    # queries were all filtered out, a checkpoint with more queries) is served by the torch form of the same function
    served = [q.shape[0] == 1 and 1 <= q.shape[1] <= be.PANOP_QMAX and 2 <= q.shape[2] <= 64 for _, q in outputs]
    work = []
    for (v, q), ok in zip(outputs, served):
        if not ok:
            work.append(None)
            continue
        masks = (torch.sigmoid(v.F) if input_voxel_logit else v.F).contiguous()
        qp = (F.softmax(q[0], dim=-1) if input_query_logit else q[0]).contiguous().float()
        rows = be.panoptic_rows(masks, qp, object_mask_threshold, overlap_threshold, vox_occ_threshold, thing_ids)
        work.append((v, masks, qp, rows))
    # one copy for every output's tables: qtab (rows 0-3) | nk (4) | seg (5-9) | areas (10-11), 128 int32 each
    live = [w for w in work if w is not None]
    tabs = iter(torch.stack([w[3]["tabs"] for w in live]).cpu()) if live else iter(())
    results = []
    for w, (v0, q0) in zip(work, outputs):
        if w is None:
            results.append(torch_form(v0, q0))
            continue
        (v, masks, qp, rows), t = w, next(tabs)
        probs = t[3].view(torch.float32)
        K, n_seg = int(t[4, 0]), int(t[9, 0])
        infos = [{"id": int(t[5, s]), "isthing": bool(t[6, s]), "category_id": int(t[7, s]), "query_id": int(t[8, s]),
                  "confidence": float(probs[int(t[8, s])]), "all_class_probs": qp[int(t[8, s])]} for s in range(n_seg)]
        res = PanopticResult(panoptic_seg_sparses=[rows["panoptic"]], segments_infos=[infos],
                             semantic_seg_sparses=[rows["semantic"]], ins_uncertainty_sparses=[rows["ins_unc"]],
                             vox_confidence_sparses=[rows["vox_conf"]], vox_uncertainty_sparses=[rows["vox_unc"]])
        res._lazy = (v.C, scene_size, min_C, rows, masks, [int(k) for k in t[1, :K]])
        results.append(res)
    return results


def panoptic_inference(voxel_output: ME.SparseTensor, query_output: torch.Tensor, overlap_threshold: float,
                       object_mask_threshold: float, thing_ids: Sequence[int], min_C, scene_size,
                       input_query_logit: bool = True, input_voxel_logit: bool = False,
                       vox_occ_threshold: float = 0.3):
    """The reference's signature (helper.py:91-103); one output = `panoptic_inference_many` of one pair."""
    return panoptic_inference_many([(voxel_output, query_output)], overlap_threshold, object_mask_threshold, thing_ids, min_C,
                                   scene_size, input_query_logit, input_voxel_logit, vox_occ_threshold)[0]


def _panoptic_inference_torch(voxel_output: ME.SparseTensor, query_output: torch.Tensor, overlap_threshold: float,
                              object_mask_threshold: float, thing_ids: Sequence[int], min_C, scene_size,
                              input_query_logit: bool = True, input_voxel_logit: bool = False,
                              vox_occ_threshold: float = 0.3):
    bs = query_output.shape[0]
    n_classes = query_output.shape[-1] - 1
    vF = torch.sigmoid(voxel_output.F) if input_voxel_logit else voxel_output.F
    vC = voxel_output.C
    dev = vF.device
    thing = set(int(t) for t in thing_ids)
    res = {k: [] for k in ("vox_all_mask_probs_denses", "panoptic_seg_denses", "semantic_seg_denses",
                           "ins_uncertainty_denses", "vox_confidence_denses", "vox_uncertainty_denses",
                           "panoptic_seg_sparses", "segments_infos", "semantic_seg_sparses", "ins_uncertainty_sparses",
                           "vox_confidence_sparses", "vox_uncertainty_sparses")}
    for b in range(bs):
        qp = F.softmax(query_output[b], dim=-1) if input_query_logit else query_output[b]
        probs, labels = qp.max(-1)
        keep = labels.ne(0) & labels.ne(n_classes) & (probs > object_mask_threshold)
        kq_probs, kq_classes, kq_all = probs[keep], labels[keep], qp[keep]
        kq_ids = torch.arange(keep.shape[0], device=dev)[keep]
        sel = vC[:, 0] == b
        coords = vC[sel]
        masks = vF[sel][:, keep]                                       # [N, K]
        N, K = masks.shape
        panoptic = torch.zeros(N, dtype=torch.int32, device=dev)
        semantic = torch.zeros(N, dtype=torch.int32, device=dev)
        ins_unc = torch.zeros(N, dtype=torch.float32, device=dev)
        vox_unc = torch.zeros(N, dtype=torch.float32, device=dev)
        vox_conf = torch.zeros(N, dtype=torch.float32, device=dev)
        all_mask = torch.zeros((N, K), dtype=torch.float32, device=dev)
        segments_info: List[dict] = []
        if K != 0 and N != 0:
            combined = kq_probs.view(1, -1) * masks
            winner = combined.argmax(dim=1)                              # query (kept index) owning the voxel
            own = masks.gather(1, winner[:, None]).squeeze(1) >= vox_occ_threshold
            mask_area = torch.bincount(winner[own], minlength=K)
            orig_area = (masks >= vox_occ_threshold).sum(dim=0)
            host = torch.stack([mask_area, orig_area, kq_classes.to(mask_area.dtype)]).cpu().tolist()
            probs_host = kq_probs.cpu().tolist()
            ids_host = kq_ids.cpu().tolist()
            seg_of = [0] * K          # segment id written for voxels of query k (0 = nothing written)
            full = [False] * K        # query opened a segment (class / confidences are written too)
            stuff_memory = {}
            current = 0
            for k in range(K):
                ma, oa, cls = int(host[0][k]), int(host[1][k]), int(host[2][k])
                if not (ma > 0 and oa > 0) or ma / oa < overlap_threshold:
                    continue
                isthing = cls in thing
                if not isthing:
                    if cls in stuff_memory:
                        seg_of[k] = stuff_memory[cls]      # merged: only the panoptic id is written
                        continue
                    stuff_memory[cls] = current + 1
                current += 1
                seg_of[k], full[k] = current, True
                segments_info.append({"id": current, "isthing": bool(isthing), "category_id": cls,
                                      "query_id": ids_host[k], "confidence": probs_host[k],
                                      "all_class_probs": kq_all[k]})
            seg_t = torch.tensor(seg_of, dtype=torch.int32, device=dev)
            full_t = torch.tensor(full, dtype=torch.bool, device=dev)
            hit = own & (seg_t[winner] != 0)
            fullhit = own & full_t[winner]
            panoptic = torch.where(hit, seg_t[winner], panoptic)
            semantic = torch.where(fullhit, kq_classes[winner].to(torch.int32), semantic)
            norm = masks / (masks.sum(1, keepdim=True) + 1e-8)
            vox_conf = torch.where(fullhit, norm.gather(1, winner[:, None]).squeeze(1), vox_conf)
            all_mask = torch.where(fullhit[:, None], masks, all_mask)
            ins_unc = torch.where(fullhit, kq_probs[winner], ins_unc)
            vox_unc = torch.where(fullhit, (combined / combined.sum(1, keepdim=True)).max(1)[0], vox_unc)
        dense = lambda v: to_dense(v.unsqueeze(-1) if v.dim() == 1 else v, coords, scene_size, min_C).squeeze()
        res["semantic_seg_denses"].append(dense(semantic))
        res["panoptic_seg_sparses"].append(panoptic)
        res["semantic_seg_sparses"].append(semantic)
        res["ins_uncertainty_sparses"].append(ins_unc)
        res["vox_confidence_sparses"].append(vox_conf)
        res["vox_uncertainty_sparses"].append(vox_unc)
        res["panoptic_seg_denses"].append(dense(panoptic))
        res["segments_infos"].append(segments_info)
        res["ins_uncertainty_denses"].append(dense(ins_unc))
        res["vox_uncertainty_denses"].append(dense(vox_unc))
        res["vox_confidence_denses"].append(dense(vox_conf))
        res["vox_all_mask_probs_denses"].append(dense(all_mask))
    for k in ("panoptic_seg_denses", "semantic_seg_denses", "ins_uncertainty_denses", "vox_confidence_denses",
              "vox_uncertainty_denses"):
        res[k] = torch.stack(res[k])
    return res
