"""Phase ablation of k_conv_win (forced onto the window kernel) on layers captured from one benchmark step."""
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
want = {(27, 64, 64): None, (27, 128, 128): None, (27, 256, 256): None}
inner = be.conv_fwd


def spy(x, weight, nbr, n_out, **kw):
    out = inner(x, weight, nbr, n_out, **kw)
    if weight is not None and kw.get("win") is not None:
        key = tuple(weight.shape)
        if key in want and (want[key] is None or want[key][3] < n_out):
            want[key] = (x, weight, nbr, n_out, dict(kw))
    return out


be.conv_fwd = spy
with torch.no_grad():
    bench.run_scene(net, scene, tk)
be.conv_fwd = inner
lib = be.lib
lib.ph_conv_dma_set_ablate.argtypes = [C.c_int]
for key, rec in want.items():
    if rec is None:
        continue
    x, weight, nbr, n_out, kw = rec
    print(key, n_out, "window stats", kw["win"]["stats"].tolist(), "tiles", (n_out + 127) // 128,
          "mean cnt", float(kw["win"]["cnt"].float().mean()))
    for force, fname in ((1, "windows"), (-1, "gather")):
        be.set_route({1: 1, -1: 2, 0: 0}[force])
        for mask, name in ((0, "full"), (1, "no MFMA"), (8, "no fragment reads"), (9, "no MFMA, no frag reads"),
                           (2, "no window reload"), (4, "no weight DMA"), (15, "loop skeleton only")):
            lib.ph_conv_dma_set_ablate(mask)
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                inner(x, weight, nbr, n_out, **kw)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            print(f"  {fname:8s} {name:26s} {min(ts[1:]):8.1f} us", flush=True)
    lib.ph_conv_dma_set_ablate(0)
    be.set_route(0)
