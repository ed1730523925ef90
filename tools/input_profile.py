"""Input stage alone (prepare_input) for a kernel trace: python tools/input_profile.py [iters] ; wall time per call printed."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import make_scene
dev = torch.device("cuda", 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
with torch.no_grad():
    for fused_stage in (True, False):
        for _ in range(2):
            x = net.prepare_input(scene.in_feats, scene.in_coords, fused_stage=fused_stage)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters):
            x = net.prepare_input(scene.in_feats, scene.in_coords, fused_stage=fused_stage)
        torch.cuda.synchronize()
        print(f"fused={fused_stage}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per call, rows {x.F.shape}")
