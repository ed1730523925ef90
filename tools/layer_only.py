"""Replay ONE captured convolution launch of a benchmark step a few times (for rocprofv3 --pmc passes on a single kernel):
    python tools/layer_only.py <kernel id (5 = window / gather pair, 4 = k_conv_dma, 6 = k_conv_wide)> <min rows> [iters]
The launch with the most rows among those the filter admits is replayed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import TeacherKeep, make_scene
from pasco_amd.me.backend import hip_backend

kid, min_rows = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
be = hip_backend()
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
best = [None]
inner = be.conv_fwd


def spy(x, weight, nbr, n_out, **kw):
    out = inner(x, weight, nbr, n_out, **kw)
    cfg = be.conv_last_config()
    kvol = (weight.shape[0] if weight is not None and weight.dim() == 3 else 1) if weight is not None else kw["wshape"][0]
    if cfg["kernel"] == kid and n_out >= min_rows and kvol >= int(os.environ.get("LAYER_ONLY_MIN_KVOL", "0")) and \
            (best[0] is None or n_out > best[0][3]):
        best[0] = (x, weight, nbr, n_out, dict(kw))
    return out


with torch.no_grad():
    bench.run_scene(net, scene, tk)
    be.conv_fwd = spy
    bench.run_scene(net, scene, tk)
    be.conv_fwd = inner
x, weight, nbr, n_out, kw = best[0]
torch.cuda.synchronize()
print("REPLAY n_out", n_out, "flags", sorted(k for k, v in kw.items() if v is not None), flush=True)
for _ in range(iters):
    inner(x, weight, nbr, n_out, **kw)
torch.cuda.synchronize()
print("done")
