"""Time ph_win_build on the 27-offset kernel maps of one S10 M=3 step (the maps the 64-channel layers run on)."""
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
seen = {}
inner = be.win_build


def spy(nbr, *a, **k):
    seen[(nbr.data_ptr(), tuple(nbr.shape))] = nbr
    return inner(nbr, *a, **k)


with torch.no_grad():
    be.win_build = spy
    bench.run_scene(net, scene, tk)
    be.win_build = inner
tot = 0.0
for (_, shape), nbr in sorted(seen.items(), key=lambda kv: -kv[0][1][1]):
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        inner(nbr)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"n = {shape[1]:7d}: {min(ts[1:]):7.1f} us")
    tot += min(ts[1:])
print(f"{len(seen)} maps, {tot:.1f} us per step, {tot / max(len(seen), 1):.1f} us per map")
