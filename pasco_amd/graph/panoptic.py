"""Panoptic post-processing (reference: pasco/models/helper.py:91-303 `panoptic_inference`).

Queries with a non-empty, non-dustbin class and probability > object_mask_threshold compete per voxel
(argmax of query prob x mask prob); a query keeps its voxels if enough of its own mask
(mask prob >= vox_occ_threshold) survived the competition; stuff segments of one class merge.

The reference walks the kept queries in a Python loop with ~6 `.item()` host syncs per query.  Here
the per-query areas come from one pass over the voxels (bincount), the segment-id bookkeeping - a
sequential walk over <= 100 queries - runs on the host on those counts (one device->host copy), and
the per-voxel outputs are one gather through per-query tables.  Results are identical, including the
reference's quirk that voxels of a *merged* stuff segment keep semantic class 0.
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


def panoptic_inference(voxel_output: ME.SparseTensor, query_output: torch.Tensor, overlap_threshold: float,
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
                           "panoptic_seg_sparses", "segments_infos")}
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
