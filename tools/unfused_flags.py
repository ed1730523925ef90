"""Which guarded module convolutions of the unfused MIMO-1 step raise their range flag (i.e. are redone by the exact kernel)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph import fused
from pasco_amd.graph.synth import TeacherKeep, make_scene
from pasco_amd.me import modules as M
from pasco_amd.me.backend import hip_backend

dev = torch.device("cuda", 0)
net = bench.build_net(1, 283, dev)
scene = make_scene(seed=0, n_infers=1, in_channels=283).to(dev)
tk = TeacherKeep(scene, dev)
be = hip_backend()
rec = []
inner_split, inner_conv = be.split_rows, be.conv_fwd


def split_rows(x, **kw):
    if kw.get("status") is not None:
        rec.append([kw["status"], tuple(x.shape), float(x.abs().max()), kw.get("pro_scale") is not None, kw.get("pro_act", 0), None])
    return inner_split(x, **kw)


def conv_fwd(x, w, nbr, n_out, **kw):
    if kw.get("exact_if") is not None and rec:
        rec[-1][5] = (tuple(w.shape), n_out)
    return inner_conv(x, w, nbr, n_out, **kw)


fused.set_fusion(False)
fused.set_conv_precision("f32")
with torch.no_grad():
    bench.run_scene(net, scene, tk)
    be.split_rows, be.conv_fwd = split_rows, conv_fwd
    bench.run_scene(net, scene, tk)
torch.cuda.synchronize()
fired = 0
for flag, shape, amax, has_bn, act, conv in rec:
    v = int(flag.item())
    fired += v & 1
    if v & 1:
        print("FIRED", shape, "max |x| %.1f" % amax, "bn" if has_bn else "-", "act", act, conv)
print(f"{len(rec)} guarded convolutions, {fired} redone by the exact kernel")
