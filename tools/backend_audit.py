"""Which C-ABI entry points one benchmark step calls, by calling line inside pasco_amd.graph / pasco_amd.me.core, with the
rows they move: the launch-count side of the non-convolution kernel time (profiles/README.md, round 5).

    python tools/backend_audit.py [out.txt]"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pasco_amd.graph.synth import TeacherKeep, make_scene
from pasco_amd.me.backend import hip_backend

out_path = sys.argv[1] if len(sys.argv) > 1 else None
dev = torch.device("cuda", 0)
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
be = hip_backend()
counts = collections.Counter()
rows = collections.Counter()
NAMES = ["split_rows", "gather_rows", "scatter_add_rows", "map_insert_launch", "map_find", "nbr_build", "win_build", "kmap_compact",
         "rowlist_build", "mask_compact", "mask_compact_many", "keep_mask", "coords_floor", "coords_expand", "maxpool_fwd",
         "bits_block_or", "attn_mask_pack", "bits_or_reduce", "pos_aug", "sine_pe", "to_dense", "dense_gather"]


def wrap(name):
    inner = getattr(be, name)

    def f(*a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if x.filename.startswith(ROOT) and "backend_audit" not in x.filename
              and not x.filename.endswith("me/backend.py")]
        site = f"{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno}" if fr else "?"
        up = f"{os.path.relpath(fr[-2].filename, ROOT)}:{fr[-2].lineno}" if len(fr) > 1 else ""
        n = 0
        for t in a:
            if torch.is_tensor(t) and t.dim() >= 1:
                n = max(n, int(t.shape[0] if t.dim() == 1 or name != "nbr_build" else t.shape[0]))
        key = (name, site, up)
        counts[key] += 1
        rows[key] += n
        return inner(*a, **k)
    setattr(be, name, f)


with torch.no_grad():
    for _ in range(2):
        bench.run_scene(net, scene, tk)
    torch.cuda.synchronize()
    for n in NAMES:
        if hasattr(be, n):
            wrap(n)
    bench.run_scene(net, scene, tk)
lines = [f"backend entry points of one step (S10, M = 3): {sum(counts.values())} calls"]
by = collections.Counter()
for (name, _, _), c in counts.items():
    by[name] += c
lines.append("by entry: " + ", ".join(f"{k} {v}" for k, v in by.most_common()))
for key, c in sorted(counts.items(), key=lambda kv: (-by[kv[0][0]], kv[0][0], -kv[1])):
    lines.append(f"{c:3d} x {key[0]:18s} rows(sum) {rows[key]:9d}  {key[1]}  <- {key[2]}")
text = "\n".join(lines)
print(text)
if out_path:
    with open(out_path, "w") as f:
        f.write(text + "\n")
