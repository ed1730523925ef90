"""Count implicit host<->device synchronisations of one bench step by source line
(torch.cuda.set_sync_debug_mode("warn")); diagnostic only."""
import os, sys, warnings, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import make_scene, TeacherKeep

dev = torch.device("cuda", 0)
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
with torch.no_grad():
    for _ in range(2):
        bench.run_scene(net, scene, tk)
    torch.cuda.synchronize()
    counts = collections.Counter()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def hook(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" not in str(message):
            return
        frames = [fr for fr in reversed(traceback.extract_stack()[:-1])
                  if fr.filename.startswith(root) and "sync_audit" not in fr.filename]
        if not frames:
            counts["<outside repo>"] += 1
            return
        # innermost repo frame, plus (for the generic backend / coordinate-map helpers) the first caller outside them
        label = f"{os.path.relpath(frames[0].filename, root)}:{frames[0].lineno}"
        for fr in frames[1:]:
            if "/me/" not in fr.filename:
                label += f"  <- {os.path.relpath(fr.filename, root)}:{fr.lineno} {fr.line.strip()[:70]}"
                break
        counts[label] += 1

    warnings.showwarning = hook
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    bench.run_scene(net, scene, tk)
    torch.cuda.set_sync_debug_mode("default")
print("implicit syncs in one step:", sum(counts.values()))
for k, v in counts.most_common(60):
    print(f"{v:4d}  {k}")
