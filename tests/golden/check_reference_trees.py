"""Build container only (needs /root/reference): re-run the REFERENCE's own module trees (`UNet3DV2`, `TransformerPredictorV2`
with `MinkowskiEngine := pasco_amd.me`) under `torch.no_grad()` - where the plain modules defer BatchNorm / ReLU into the next
convolution's prologue (SparseTensor.deferred) - and compare with the stored fixtures, which were generated before that
existed: integer outputs identical, floating-point outputs to a few 1e-6 of their mean magnitude.

    cd tests/golden && python check_reference_trees.py"""
import numpy as np
import torch

import make_golden as G
import pasco_amd.me.modules as M

captured = {}


def save(name, **arrays):
    captured[name] = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in arrays.items()}


G.save = save
n_def = [0]
orig = M.SparseTensor.deferred.__func__


def counting(cls, *a, **k):
    n_def[0] += 1
    return orig(cls, *a, **k)


M.SparseTensor.deferred = classmethod(counting)
with torch.no_grad():
    for tag, args in (("m1_light", (1, False)), ("m2_light", (2, False)), ("m1_heavy", (1, True)), ("m2_fallback", (2, False))):
        old = np.load(f"unet_{tag}.npz")
        n_def[0] = 0
        G._golden_unet(args[0], args[1], tag, int(old["seed"]), 1 if tag == "m2_fallback" else None)
        new = captured[f"unet_{tag}.npz"]
        worst = 0.0
        for k in old.files:
            a, b = old[k], new[k]
            assert a.shape == b.shape, (k, a.shape, b.shape)
            if a.dtype.kind == "f":
                worst = max(worst, float(np.abs(a - b).max() / (np.abs(a).mean() + 1e-12)))
            else:
                assert np.array_equal(a, b), k
        assert n_def[0] > 100 and worst < 5e-5, (tag, n_def[0], worst)
        print(f"{tag}: {n_def[0]} deferred operations; the reference's module trees vs the stored fixture: worst |diff| / mean |y| = {worst:.2e}")
