"""Bitwise comparison of the three split-precision kernels on identity-map / k=3 problems (development)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasco_amd.graph.synth import make_occupancy
from pasco_amd.me.backend import hip_backend
from pasco_amd.me.core import kernel_offsets

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from devlib import use_dev_library   # noqa: E402
use_dev_library()      # the hooks below exist only in the development build (-DPH_DEV)
be = hip_backend()
lib = be.lib
lib.ph_conv_dma_set_ablate.argtypes = [C.c_int]
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(1)
g1 = np.argwhere(make_occupancy(0))
c1 = torch.from_numpy(np.concatenate([np.zeros((g1.shape[0], 1), np.int64), g1], 1)).int().to(dev).contiguous()
tk, tv, _, _, _ = be.map_insert(c1, dedup=False)
nbr27 = be.nbr_build(c1, tk, tv, kernel_offsets(3, 1))
for (kvol, cin, cout, n) in ((1, 64, 128, 380049), (1, 64, 64, 100000), (27, 64, 64, c1.shape[0]), (27, 128, 128, c1.shape[0]),
                             (1, 384, 384, 300000), (27, 256, 256, 60000)):
    nbr = None
    if kvol == 27:
        n = min(n, c1.shape[0])
        nbr = nbr27[:, :n].contiguous()
        nbr = torch.where(nbr < n, nbr, torch.full_like(nbr, -1)).contiguous()
    x = torch.randn(n, cin, device=dev, generator=g)
    w = torch.randn(kvol, cin, cout, device=dev, generator=g) / (cin ** 0.5)
    wq = w if kvol > 1 else w[0]
    s2, s1 = be.split_weight_rows(w), be.split_weight_f16(w)
    outs = {}
    os.environ.pop("X", None)
    for name, mask in (("dma_tall", 0), ("dma_128", 0x100)):
        lib.ph_conv_dma_set_ablate(mask)
        outs[name] = be.conv_fwd(x, wq, nbr, n, split=s2)
        outs[name + "_cfg"] = be.conv_last_config()
    lib.ph_conv_dma_set_ablate(0)
    outs["mode1"] = be.conv_fwd(x, wq, nbr, n, split=s1)
    outs["f32"] = be.conv_fwd(x, wq, nbr, n)
    ref = outs["mode1"]
    for name in ("dma_tall", "dma_128", "f32"):
        d = (outs[name] - ref).abs()
        nz = int((d > 0).sum())
        print(f"k{kvol} {cin}->{cout} n={n}: {name:9s} vs mode1: max {float(d.max()):.3e} differing {nz} of {d.numel()}",
              outs.get(name + "_cfg", ""), flush=True)
        if nz and name != "f32":
            idx = (d > 0).nonzero()[:8]
            print("   first differing (row, col):", idx.tolist())
