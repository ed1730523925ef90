"""cProfile of the host side of one bench step (which Python functions cost CPU time); diagnostic only."""
import os, sys, cProfile, pstats, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import make_scene, TeacherKeep

dev = torch.device("cuda", 0)
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
with torch.no_grad():
    for _ in range(3):
        bench.run_scene(net, scene, tk)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        bench.run_scene(net, scene, tk)
    torch.cuda.synchronize()
    pr.disable()
st = io.StringIO()
ps = pstats.Stats(pr, stream=st).sort_stats(sys.argv[1] if len(sys.argv) > 1 else "tottime")
ps.print_stats(45)
print(st.getvalue())
