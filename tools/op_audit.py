"""Which torch operators one benchmark step launches, by calling line inside pasco_amd (the glue between the library's own
kernels): a TorchDispatchMode counts every aten op of one step after warm-up.

    python tools/op_audit.py [out.txt] [top N]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import TeacherKeep, make_scene

out_path = sys.argv[1] if len(sys.argv) > 1 else None
top = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda", 0)
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts = collections.Counter()
VIEWS = {"view", "reshape", "_unsafe_view", "expand", "permute", "transpose", "t", "slice", "select", "unsqueeze", "squeeze", "alias",
         "detach", "as_strided", "unbind", "split", "split_with_sizes", "_reshape_alias", "empty", "empty_like", "empty_strided",
         "sym_size", "sym_stride", "sym_numel", "is_pinned", "_local_scalar_dense", "new_empty", "lift_fresh", "unfold", "narrow",
         "diagonal", "view_as_real", "stride", "size", "numel", "dim", "is_contiguous", "_to_copy_view", "new_empty_strided",
         "result_type", "chunk", "movedim", "flatten"}


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in VIEWS:
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if fr.filename.startswith(os.path.join(ROOT, "pasco_amd")):
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {(fr.line or '')[:80]}"
                    break
            counts[(name, site)] += 1
        return func(*args, **(kwargs or {}))


with torch.no_grad():
    bench.run_scene(net, scene, tk)
    bench.run_scene(net, scene, tk)
    torch.cuda.synchronize()
    with Mode():
        bench.run_scene(net, scene, tk)
    torch.cuda.synchronize()
by_op = collections.Counter()
for (name, site), c in counts.items():
    by_op[name] += c
lines = [f"aten ops of one step (views excluded; a replayed hipGraph of the query side shows as nothing here): {sum(counts.values())}",
         "by operator: " + ", ".join(f"{k} {v}" for k, v in by_op.most_common(40)), ""]
for (name, site), c in counts.most_common(top):
    lines.append(f"{c:4d} {name:22s} {site}")
print("\n".join(lines))
if out_path:
    open(out_path, "w").write("\n".join(lines) + "\n")
