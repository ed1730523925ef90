"""k = 1 streams on k_conv_lin: the launch against k_conv_dma, and with 1 - 3 workgroups per CU (development hook ph_conv_lin_set).

    python tools/lin_time.py > gpurun_out/lin_time.txt
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasco_amd.me.backend import hip_backend   # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from devlib import use_dev_library   # noqa: E402
use_dev_library()      # the hooks below exist only in the development build (-DPH_DEV)
be = hip_backend()
lib = be.lib
lib.ph_conv_lin_set.argtypes = [C.c_int]
lib.ph_conv_lin_set.restype = None
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
SHAPES = [(380049, 64, 128, True, False), (380049, 128, 256, True, False), (136887, 128, 384, True, True),
          (210542, 64, 100, False, True), (127240, 128, 128, False, True)]
VARIANTS = [("k_conv_dma", 0), ("lin", 1), ("lin 1 wg/CU", 1 | (1 << 8)), ("lin 3 wg/CU", 1 | (3 << 8))]
if len(sys.argv) > 1 and sys.argv[1] == "first":      # counter passes: the 64 -> 128 stream only
    SHAPES, VARIANTS = SHAPES[:1], VARIANTS[:2]
for n, cin, cout, emit, axis in SHAPES:
    x = torch.randn(n, cin, generator=g).to(dev)
    w = (torch.randn(1, cin, cout, generator=g) / cin ** 0.5).to(dev)
    split, xs = be.split_weight_rows(w), be.split_rows(x)
    kw = dict(bias=torch.zeros(cout, device=dev))
    if axis:
        tab = torch.randn(3, 260, cout, generator=g).to(dev)
        ac = torch.cat([torch.zeros(n, 1, dtype=torch.int32), torch.randint(0, 256, (n, 3), generator=g, dtype=torch.int32)], dim=1).contiguous().to(dev)
        kw["axis"] = (tab, ac, 0)
    if emit:
        kw.update(emit_split=(None, None, 0), want_out=False)
    alg = n * cin * 4 + n * cout * 4
    line = f"k1 {cin:3d}->{cout:<3d} n={n:6d} {'E' if emit else '-'}{'A' if axis else '-'} ({alg / 1e6:5.0f} MB):"
    for name, v in VARIANTS:
        lib.ph_conv_lin_set(v)
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            be.conv_fwd(None, None, None, n, xshape=(n, cin), wshape=(1, cin, cout), split=split, in_split=xs, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        t = min(ts[1:])
        line += f"  [{name}] {t:6.1f} us ({alg / t / 1e6:4.2f} TB/s)"
    lib.ph_conv_lin_set(1)
    print(line, flush=True)
