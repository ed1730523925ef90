"""Where does a tall projection's time go?  The 64 -> 384 launches of one benchmark step (K / V / mask-feature projections,
631 k rows) re-run with pieces of the epilogue removed: table residual, operand emission, fp32 output.
python tools/lin_variants.py [out.txt]"""
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
caught = {}
inner = be.conv_fwd


def spy(x, weight, nbr, n_out, **kw):
    out = inner(x, weight, nbr, n_out, **kw)
    if nbr is None and kw.get("split") is not None and len(kw["split"]) == 2 and n_out > 100000:
        shape = tuple(weight.shape[-2:]) if weight is not None else tuple(kw["wshape"][-2:])
        key = (shape, n_out, kw.get("axis") is not None, kw.get("emit_split") is not None, kw.get("want_out", True))
        caught.setdefault(key, (x, weight, nbr, n_out, dict(kw)))
    return out


be.conv_fwd = spy
with torch.no_grad():
    bench.run_scene(net, scene, tk)
be.conv_fwd = inner


def t_of(fn):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts[1:])


lines = []
for key, (x, weight, nbr, n_out, kw) in sorted(caught.items(), key=lambda kv: -kv[0][1]):
    (cin, cout), n, has_axis, has_emit, want_out = key
    variants = [("as launched", dict())]
    if has_axis:
        variants.append(("no table residual", dict(axis=None)))
        tab, coords, lo = kw["axis"]
        variants.append(("table residual, all rows at one coordinate", dict(axis=(tab, torch.zeros_like(coords) + lo + 3, lo))))
        perm = torch.randperm(coords.shape[0], device=coords.device)
        variants.append(("table residual, coordinates shuffled", dict(axis=(tab, coords[perm].contiguous(), lo))))
    if has_emit:
        variants.append(("fp32 output instead of the operand", dict(emit_split=None, want_out=True)))
        if has_axis:
            variants.append(("fp32 output, no table residual", dict(emit_split=None, want_out=True, axis=None)))
    for name, over in variants:
        k2 = dict(kw)
        k2.update(over)
        us = t_of(lambda: inner(x, weight, nbr, n_out, **k2))
        cfg = be.conv_last_config()
        line = f"k1 {cin:3d}->{cout:<3d} n={n:7d} axis={int(has_axis)} emit={int(has_emit)} out={int(want_out)} kernel={cfg['kernel']:2d}  {name:46s} {us:8.1f} us"
        print(line, flush=True)
        lines.append(line)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
