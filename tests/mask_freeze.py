"""Test helper: record / force the mask transformer's attention-mask DECISIONS (`mask logit > 0`,
transformer_predictor_v2.py:224 `keep_mask = outputs_mask.sigmoid() > 0.5`).

The masks of decoder layer l are thresholds of layer l - 1's prediction, so two correct implementations whose logits
differ by rounding can decide a near-zero logit differently and then see different attention masks.  Recording the
decisions of one run and forcing them into another separates that effect from arithmetic error: with the decisions
frozen the remaining difference is the arithmetic's alone."""
import contextlib

import torch

from pasco_amd.graph.transformer import TransformerPredictorV2


@contextlib.contextmanager
def recording(store: list):
    """Append (outputs_mask > 0) [B, P, Q] (bool, CPU) of every `compute_mask_bits` call to `store`."""
    inner = TransformerPredictorV2.compute_mask_bits

    def compute_mask_bits(self, outputs_mask, voxel_coord, *a, **k):
        store.append((outputs_mask > 0).cpu())
        return inner(self, outputs_mask, voxel_coord, *a, **k)
    TransformerPredictorV2.compute_mask_bits = compute_mask_bits
    try:
        yield store
    finally:
        TransformerPredictorV2.compute_mask_bits = inner


@contextlib.contextmanager
def forcing(store: list, stats: dict, noise: float = 1e-3):
    """Replace the decisions of every `compute_mask_bits` call by the recorded ones (same call order, same rows).
    stats: decisions / differ (own decision != recorded) / differ_above_noise (... with |own logit| > noise * mean |logit|)."""
    inner = TransformerPredictorV2.compute_mask_bits
    it = iter(store)
    stats.update(decisions=0, differ=0, differ_above_noise=0, calls=0)

    def compute_mask_bits(self, outputs_mask, voxel_coord, *a, **k):
        ref = next(it).to(outputs_mask.device)
        assert ref.shape == outputs_mask.shape, "the forced run has other rows than the recorded one"
        diff = (outputs_mask > 0) != ref
        stats["calls"] += 1
        stats["decisions"] += int(diff.numel())
        stats["differ"] += int(diff.sum())
        stats["differ_above_noise"] += int((diff & (outputs_mask.abs() > noise * outputs_mask.abs().mean())).sum())
        forced = torch.where(ref, 1.0, -1.0).to(outputs_mask.dtype)
        return inner(self, forced, voxel_coord, *a, **k)
    TransformerPredictorV2.compute_mask_bits = compute_mask_bits
    try:
        yield stats
    finally:
        TransformerPredictorV2.compute_mask_bits = inner
