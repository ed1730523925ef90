"""Run only the k=3 conv at one S10 level (for PMC passes): conv_only.py <level: 1|2|4|U4> [iters] [split|split2]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasco_amd.me.backend import hip_backend
from pasco_amd.me.core import kernel_offsets
from pasco_amd.graph.synth import make_occupancy
level = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
be = hip_backend()
if level == "gemm":      # plain dense GEMM through the same kernel: identity map, 631 626 x 384 -> 384
    n, c = 631626, 384
    x = torch.randn(n, c, device="cuda"); w = torch.randn(c, c, device="cuda") / 20; out = torch.empty(n, c, device="cuda")
    mode = sys.argv[3] if len(sys.argv) > 3 else ""
    split = be.split_weight_rows(w) if mode == "split2" else (be.split_weight_f16(w) if mode == "split" else None)
    xs2 = be.split_rows(x) if mode == "split2" else None
    for _ in range(iters):
        be.conv_fwd(x, w, None, n, out=out, split=split, in_split=xs2)
    torch.cuda.synchronize()
    print("done gemm", n, c)
    sys.exit(0)
if level == "U4":
    xs = np.stack(np.meshgrid(np.arange(64), np.arange(64), np.arange(8), indexing="ij"), -1).reshape(-1, 3) * 4
    ts, c = 4, 256
else:
    ts = int(level); c = {1: 64, 2: 128, 4: 256}[ts]
    g1 = np.argwhere(make_occupancy(0)); xs = np.unique(np.floor_divide(g1, ts) * ts, axis=0)
coords = torch.from_numpy(np.concatenate([np.zeros((xs.shape[0], 1), np.int64), xs], 1)).int().cuda()
tk, tv, _, _, _ = be.map_insert(coords, dedup=False)
nbr = be.nbr_build(coords, tk, tv, kernel_offsets(3, ts))
n = coords.shape[0]
x = torch.randn(n, c, device="cuda"); w = torch.randn(27, c, c, device="cuda") / 40; out = torch.empty(n, c, device="cuda")
mode = sys.argv[3] if len(sys.argv) > 3 else ""
split = be.split_weight_f16(w) if mode == "split" else (be.split_weight_rows(w) if mode == "split2" else None)
xs2 = be.split_rows(x) if mode == "split2" else None
for _ in range(iters):
    be.conv_fwd(x, w, nbr, n, out=out, split=split, in_split=xs2)
torch.cuda.synchronize()
print("done", level, n, c)
