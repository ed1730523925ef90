"""Procedural weights for fixtures whose state dict is too large to commit (the f = 64 graph holds ~120 M parameters).

Every tensor of a state dict is a deterministic function of (canonical key, shape, seed): a CPU generator seeded by a CRC of
the key fills it, with a scale derived from the shape (fan-in scaling keeps the activations of a random net in range).  The
fixture generator (tests/golden/make_golden.py, build container only) loads these tensors into the REFERENCE's modules and
stores inputs + outputs; the test calls the same function for its own modules, so only the outputs travel.  Test
infrastructure: nothing under pasco_amd/ imports this file.
"""
import zlib

import torch

# the one transformer predictor is registered under three parents: one set of numbers for all aliases
_ALIASES = ("unet3d.decoder_generative.transformer_predictor.", "unet3d.transformer_predictor.",
            "decoder_generative.transformer_predictor.")
HEAD_GAIN = 8.0     # class-0 column of the completion heads: `argmax != 0` then keeps a stable ~half of the voxels


def canonical(key: str) -> str:
    for a in _ALIASES:
        if key.startswith(a):
            return "transformer_predictor." + key[len(a):]
    return key


def tensor_for(key: str, shape, dtype, seed: int) -> torch.Tensor:
    k = canonical(key)
    g = torch.Generator().manual_seed((zlib.crc32(k.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(shape)
    leaf = k.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=dtype)
    if not dtype.is_floating_point:
        return torch.zeros(shape, dtype=dtype)
    if leaf == "running_var":
        return (torch.rand(shape, generator=g) + 0.5).to(dtype)
    if leaf == "running_mean":
        return (torch.randn(shape, generator=g) * 0.1).to(dtype)
    if len(shape) <= 1 or (len(shape) == 2 and shape[0] == 1 and leaf == "bias"):
        if leaf == "weight":                 # every 1-D weight of this net is a BatchNorm / LayerNorm gain
            return (torch.rand(shape, generator=g) * 0.4 + 0.8).to(dtype)
        if leaf in ("bias", "in_proj_bias"):
            return (torch.randn(shape, generator=g) * 0.05).to(dtype)
        return (torch.randn(shape, generator=g) * 0.1).to(dtype)
    # every multi-dimensional tensor at the scale of the modules' default inits (uniform +-1/sqrt(fan_in): variance
    # 1 / (3 fan_in)), which keeps the activations of the random net - and with them the attention logits - moderate
    if leaf == "kernel":                     # ME convolution: [K, cin, cout] or [cin, cout]
        fan_in = shape[0] * shape[1] if len(shape) == 3 else shape[0]
        t = torch.randn(shape, generator=g) * (1.0 / (3.0 * fan_in)) ** 0.5
        if ".completion_heads." in k:
            t[..., 0] *= HEAD_GAIN
        return t.to(dtype)
    if "query_feat" in k or "query_embed" in k:
        return torch.randn(shape, generator=g).to(dtype)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return (torch.randn(shape, generator=g) * (1.0 / (3.0 * fan_in)) ** 0.5).to(dtype)


def fill_state_dict(module, seed: int) -> None:
    """Overwrite every entry of module.state_dict() in place with its procedural tensor."""
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(tensor_for(k, v.shape, v.dtype, seed))
