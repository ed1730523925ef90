"""Who builds coordinate hash maps / neighbour tables in one step, and how big: wraps CBackend.map_insert / map_find /
nbr_build, records rows and the innermost caller outside pasco_amd/me."""
import collections
import os
import sys
import traceback

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
log = collections.OrderedDict()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/pasco_amd/me/" not in fr.filename and "map_audit" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


def wrap(name, rows_of):
    inner = getattr(be, name)

    def f(*a, **k):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        out = inner(*a, **k)
        t1.record()
        torch.cuda.synchronize()
        key = (name, site())
        e = log.setdefault(key, [0, 0, 0.0])
        e[0] += 1
        e[1] += rows_of(*a, **k)
        e[2] += t0.elapsed_time(t1)
        return out
    setattr(be, name, f)


wrap("map_insert", lambda coords, *a, **k: int(coords.shape[0]))
wrap("map_find", lambda table, q, *a, **k: int(q.shape[0]) if hasattr(q, "shape") else 0)
wrap("nbr_build", lambda *a, **k: int(a[1].shape[0]) if len(a) > 1 and hasattr(a[1], "shape") else 0)
with torch.no_grad():
    bench.run_scene(net, scene, tk)
    log.clear()
    bench.run_scene(net, scene, tk)
tot = collections.Counter()
for (name, where), (calls, rows, ms) in sorted(log.items(), key=lambda kv: -kv[1][2]):
    print(f"{name:11s} {calls:3d} calls {rows:9d} rows {ms:7.3f} ms  {where}")
    tot[name] += ms
print(dict(tot))
