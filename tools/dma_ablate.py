"""Phase ablation of k_conv_dma on layers captured from one benchmark step.  NOTE: the ablation branches were removed from
the kernel after round 2 (profiles/r2c - r2e hold the results); only the 256-row tile switch (0x100) still acts.  python tools/dma_ablate.py [out.txt]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import TeacherKeep, make_scene
from pasco_amd.me.backend import hip_backend

dev = torch.device("cuda", 0)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from devlib import use_dev_library   # noqa: E402
use_dev_library()      # the hooks below exist only in the development build (-DPH_DEV)
be = hip_backend()
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
want = {(27, 64, 64): None, (27, 128, 128): None, (27, 256, 256): None, (1, 384, 384): None}
inner = be.conv_fwd


def spy(x, weight, nbr, n_out, **kw):
    out = inner(x, weight, nbr, n_out, **kw)
    if weight is not None and kw.get("split") is not None and len(kw["split"]) == 2:
        w = weight if weight.dim() == 3 else weight[None]
        key = (w.shape[0], w.shape[1], w.shape[2])
        if key in want and (want[key] is None or want[key][3] < n_out):
            want[key] = (x, weight, nbr, n_out, dict(kw))
    return out


be.conv_fwd = spy
with torch.no_grad():
    bench.run_scene(net, scene, tk)
be.conv_fwd = inner
lib = be.lib
lib.ph_conv_dma_set_ablate.argtypes = [C.c_int]
lines = []
for key, rec in want.items():
    if rec is None:
        continue
    x, weight, nbr, n_out, kw = rec
    for mask, name in ((0x100, "256-row tiles: full"), (0x10F, "256-row tiles: hot-line DMA only"), (0, "full"), (1, "no MFMA"), (2, "A from zero line"), (4, "W one row"), (6, "A zero + W one row"),
                       (8, "no fragment reads"), (9, "no MFMA, no frag reads"), (15, "DMA of hot lines only")):
        lib.ph_conv_dma_set_ablate(mask)
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            inner(x, weight, nbr, n_out, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        cfg = be.conv_last_config()
        line = f"k{key[0]:<3d} {key[1]:3d}->{key[2]:<3d} n={n_out:7d} kernel={cfg['kernel']} bn={cfg['bn']} ksplit={cfg['ksplit']}  {name:28s} {min(ts[1:]):8.1f} us"
        print(line, flush=True)
        lines.append(line)
    lib.ph_conv_dma_set_ablate(0)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
