"""`ME.utils` names used by the reference (transformer_predictor_v2.py:230,254; ensembler.py:54)."""
from .core import batched_coordinates  # noqa: F401


def sparse_collate(coords, feats, labels=None, dtype=None, device=None):
    import torch
    bc = batched_coordinates(coords, device=device)
    f = torch.cat(feats, dim=0)
    if labels is None:
        return bc, f
    return bc, f, torch.cat(labels, dim=0)
