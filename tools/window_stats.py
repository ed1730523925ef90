"""Per-tile window sizes of the k=3 kernel maps of one benchmark step: how many distinct input rows do the 27 x BM
neighbour entries of BM consecutive output rows touch?  (Design input for the LDS-window convolution.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import TeacherKeep, make_scene
from pasco_amd.me.backend import hip_backend

dev = torch.device("cuda", 0)
be = hip_backend()
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
inner = be.conv_fwd
maps = {}


def spy(x, weight, nbr, n_out, **kw):
    if nbr is not None and nbr.shape[0] == 27:
        w = kw.get("wshape") or tuple(weight.shape)
        rec = maps.setdefault(nbr.data_ptr(), [nbr, n_out, 0, set()])
        rec[2] += 1
        rec[3].add((w[1], w[2]))
    return inner(x, weight, nbr, n_out, **kw)


be.conv_fwd = spy
with torch.no_grad():
    bench.run_scene(net, scene, tk)
be.conv_fwd = inner
for nbr, n, uses, shapes in sorted(maps.values(), key=lambda r: -r[1]):
    for BM in (128, 256):
        T = n // BM
        if T == 0:
            continue
        v = nbr[:, : T * BM].reshape(27, T, BM).permute(1, 0, 2).reshape(T, 27 * BM).long()
        big = torch.iinfo(torch.int64).max
        v = torch.where(v >= 0, v, torch.full_like(v, big))
        s, _ = v.sort(dim=1)
        distinct = ((s[:, 1:] != s[:, :-1]) & (s[:, 1:] != big)).sum(dim=1) + (s[:, 0] != big).long()
        pairs = (v != big).sum(dim=1)
        span = torch.where(s == big, torch.zeros_like(s), s).max(dim=1)[0] - s[:, 0]
        q = torch.quantile(distinct.float(), torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev)).tolist()
        print(f"n={n:7d} uses={uses} {sorted(shapes)} BM={BM}: pairs/row {float(pairs.float().mean()) / BM:5.1f}  "
              f"window rows median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} max {q[3]:.0f}  "
              f"= {q[0] / BM:.2f}x tile; gather reduction {float(pairs.float().mean()) / float(distinct.float().mean()):.1f}x; "
              f"median index span {float(span.float().median()):.0f}", flush=True)
