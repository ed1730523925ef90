"""Kernel-level profile of the ensembling stage alone (torch.profiler); diagnostic only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import make_scene, TeacherKeep
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
which = sys.argv[1] if len(sys.argv) > 1 else "ensemble"
with torch.no_grad():
    for _ in range(2):
        ret, _ = bench.run_scene(net, scene, tk)
    x = net.prepare_input(scene.in_feats, scene.in_coords)
    ret = net(x, scene.global_min_Cs, scene.global_max_Cs, scene.min_Cs, scene.max_Cs, keep_override=tk)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        if which == "ensemble":
            net.ensemble(ret, scene.Ts)
        elif which == "unet":
            net(x, scene.global_min_Cs, scene.global_max_Cs, scene.min_Cs, scene.max_Cs, keep_override=tk)
        else:
            net.prepare_input(scene.in_feats, scene.in_coords)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=int(sys.argv[2]) if len(sys.argv) > 2 else 45, max_name_column_width=60))
