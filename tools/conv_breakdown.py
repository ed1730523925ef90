"""Per-shape breakdown of the conv launches of one bench step (HIP events)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.me.backend import hip_backend
from pasco_amd.graph.synth import make_scene, TeacherKeep
from pasco_amd.graph.profiling import ConvProfiler
from pasco_amd.graph import fused
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
fused.set_conv_precision(prec)
dev = torch.device("cuda", 0)
be = hip_backend()
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
prof = ConvProfiler(); prof.wrap(be)
with torch.no_grad():
    for _ in range(2):
        bench.run_scene(net, scene, tk)
    prof.enabled = True
    for _ in range(3):
        bench.run_scene(net, scene, tk)
    prof.enabled = False
torch.cuda.synchronize()
agg = collections.OrderedDict()
for r in prof.records:
    dt = r["e0"].elapsed_time(r["e1"])
    if r["kernel"] == "k_split_rows":
        a = agg.setdefault(("k_split_rows", 0, r["c"], 0, r["n"]), [0, 0.0, 0])
        a[0] += 1; a[1] += dt
        continue
    key = (r["kernel"], r["kvol"], r["cin"], r["cout"], r["n_out"])
    P = int((r["nbr"] >= 0).sum().item()) if r["nbr"] is not None else r["n_out"]
    a = agg.setdefault(key, [0, 0.0, P])
    a[0] += 1; a[1] += dt
tot = sum(a[1] for a in agg.values()) / 3
print(f"precision {prec}: conv total {tot:.2f} ms/step")
for key, (cnt, t, P) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    kern, kvol, cin, cout, n = key
    fl = 2.0 * P * cin * cout
    print(f"{kern:13s} k{kvol:<3d} {cin:4d}->{cout:<4d} n={n:7d} pairs/row={P/max(n,1):5.1f}  x{cnt//3:2d}/step  {t/3:7.3f} ms/step  "
          f"{t/cnt*1e3:8.1f} us  {fl/(t/cnt*1e-3)/1e12:6.1f} TF")
