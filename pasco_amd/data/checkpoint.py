"""Lightning checkpoint of the reference `Net` -> `PascoNet` (scripts/eval.py:69-71 `Net.load_from_checkpoint`).

A Lightning `.ckpt` is a torch pickle {"state_dict": ..., "hyper_parameters": ..., ...}.  Of the reference's modules
(net_panoptic_sparse.py:108-175) the inference graph holds `feat`, `unet3d` and `transformer_predictor`; the
predictor is registered under three parents (`transformer_predictor.`, `unet3d.transformer_predictor.`,
`unet3d.decoder_generative.transformer_predictor.`: one module, three key prefixes); `criterion.*` (loss buffers) has
no counterpart here and is dropped.  Everything else must match exactly (strict load).

torch.load unpickles arbitrary objects (the hyper-parameters hold numpy arrays): load checkpoints only from sources
you trust."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import torch

DROPPED_PREFIXES = ("criterion.",)
PREDICTOR_ALIASES = ("transformer_predictor.", "unet3d.transformer_predictor.",
                     "unet3d.decoder_generative.transformer_predictor.")


def load_lightning_state_dict(path: str) -> Tuple[Dict[str, torch.Tensor], Dict]:
    """-> (state_dict, hyper_parameters) of a Lightning checkpoint (or of a bare state-dict file)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(ckpt, dict) and "state_dict" in ckpt:
        return ckpt["state_dict"], dict(ckpt.get("hyper_parameters", {}) or {})
    return ckpt, {}


def remap_reference_state_dict(sd: Dict[str, torch.Tensor], net) -> "OrderedDict[str, torch.Tensor]":
    """Reference `Net` keys -> `PascoNet` keys: loss buffers dropped, the predictor's aliases reconciled (all three
    must agree where present; missing aliases are filled from the one that exists)."""
    out = OrderedDict((k, v) for k, v in sd.items() if not k.startswith(DROPPED_PREFIXES))
    canon = {}
    for k, v in out.items():
        for p in PREDICTOR_ALIASES:
            if k.startswith(p):
                tail = k[len(p):]
                if tail in canon and not torch.equal(canon[tail], v):
                    raise ValueError(f"checkpoint: aliases of transformer_predictor disagree on '{tail}'")
                canon.setdefault(tail, v)
    want = net.state_dict().keys()
    for tail, v in canon.items():
        for p in PREDICTOR_ALIASES:
            if p + tail in want:
                out.setdefault(p + tail, v)
    return out


def net_from_checkpoint(path: str, device="cpu", **overrides):
    """Build `PascoNet` from the checkpoint's hyper-parameters (n_classes, n_infers, in_channels, f, num_queries,
    heavy_decoder, the three thresholds; `overrides` win, as the keyword arguments of `load_from_checkpoint` do) and
    load the weights strictly."""
    from ..graph import PascoNet
    sd, hp = load_lightning_state_dict(path)
    hp.update(overrides)
    kw = dict(n_classes=int(hp.get("n_classes", 20)), n_infers=int(hp.get("n_infers", 1)),
              in_channels=int(hp.get("in_channels", 27 + 256)), f=int(hp.get("f", 64)),
              num_queries=int(hp.get("num_queries", 100)), heavy_decoder=bool(hp.get("heavy_decoder", True)),
              iou_threshold=float(hp.get("iou_threshold", 0.2)), overlap_threshold=float(hp.get("overlap_threshold", 0.4)),
              object_mask_threshold=float(hp.get("object_mask_threshold", 0.7)))
    # the reference fixes the predictor's widths in code (384 / 1024, net_panoptic_sparse.py:111-124); read them off
    # the weights so that reduced test checkpoints load as well
    for p in PREDICTOR_ALIASES:
        if p + "query_feat.weight" in sd:
            kw["hidden_dim"] = int(sd[p + "query_feat.weight"].shape[1])
            kw["dim_feedforward"] = int(sd[p + "transformer_ffn_layers.0.linear1.weight"].shape[0])
            break
    net = PascoNet(**kw)
    missing, unexpected = net.load_state_dict(remap_reference_state_dict(sd, net), strict=False)
    if missing or unexpected:
        raise RuntimeError(f"checkpoint does not match PascoNet: missing {list(missing)[:6]}, unexpected {list(unexpected)[:6]}")
    return net.eval().to(device)
