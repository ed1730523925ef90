"""How many (row tile, kernel offset) stages of a step's gather convolutions have NO neighbour at all (every entry of the
tile's column of the neighbour table is -1)?  Such a stage multiplies zero rows: skipping it changes nothing but the time.
Per distinct launch: rows, offsets, pairs per row, and the empty fraction for 128- and 256-row tiles, weighted by launches.
    python tools/empty_stage_stats.py > gpurun_out/empty_stage_stats.txt"""
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
seen = {}
inner = be.conv_fwd
KN = {0: "mfma", 1: "f16x3", 2: "h2", 3: "rl", 4: "dma", 5: "win|dma", 6: "wide"}


def spy(x, weight, nbr, n_out, **kw):
    out = inner(x, weight, nbr, n_out, **kw)
    if nbr is not None and nbr.shape[0] >= 8:
        cfg = be.conv_last_config()
        w = weight if weight is not None else None
        cin, cout = (w.shape[-2], w.shape[-1]) if w is not None else (kw["wshape"][1], kw["wshape"][2])
        key = (nbr.shape[0], cin, cout, n_out, cfg["kernel"])
        rec = seen.get(key)
        if rec is None:
            valid = nbr >= 0
            stats = {}
            for bm in (32, 64, 128, 256):
                t = (n_out + bm - 1) // bm
                pad = t * bm - n_out
                v = torch.nn.functional.pad(valid, (0, pad)).view(nbr.shape[0], t, bm).any(dim=2)      # [K, tiles]
                stats[bm] = 1.0 - float(v.float().mean())
            seen[key] = [1, float(valid.float().sum()) / n_out, stats]
        else:
            rec[0] += 1
    return out


with torch.no_grad():
    bench.run_scene(net, scene, tk)
    be.conv_fwd = spy
    bench.run_scene(net, scene, tk)
    be.conv_fwd = inner
print("kvol  cin->cout   rows     kernel   x/step  pairs/row  empty (row block, offset) stages at 32 / 64 / 128 / 256 rows per block")
for (kvol, cin, cout, n_out, kid), (cnt, ppr, st) in sorted(seen.items(), key=lambda kv: -kv[0][0] * kv[0][3] * kv[1][0]):
    print(f"k{kvol:<4d} {cin:3d}->{cout:<3d} n={n_out:7d}  {KN.get(kid, kid):8s} x{cnt:2d}   {ppr:6.1f}     {100 * st[32]:5.1f} %  {100 * st[64]:5.1f} %  {100 * st[128]:5.1f} %  {100 * st[256]:5.1f} %")
