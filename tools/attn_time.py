"""Time ph_attn_cross_split at the three levels' sizes of an S10 M=3 step (min of 5 launches, HIP events).
LAYER_AB_BASE=<other libpascohip .so> (tools/build_base_lib.sh): also through that library, same process, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasco_amd.me.backend import CBackend, hip_backend

hip = hip_backend()
libs = [("new", hip)]
if os.environ.get("LAYER_AB_BASE"):
    libs.insert(0, ("base", CBackend(os.path.abspath(os.environ["LAYER_AB_BASE"]), "ph_", "cuda")))
g = torch.Generator().manual_seed(0)
B, H, Q, Dh = 3, 8, 100, 48
for N in (210542, 45629, 9396):
    q = (torch.randn(B, H, Q, Dh, generator=g) * Dh ** -0.5).cuda()
    k = torch.randn(B * N, H * Dh, generator=g).cuda()
    v = torch.randn(B * N, H * Dh, generator=g).cuda()
    allow = (torch.rand(B * N, Q, generator=g) > 0.5).float().cuda()
    bits, any_ = hip.attn_mask_pack(allow, B, N)
    ks, vs = hip.split_rows(k), hip.split_rows(v)
    outs = {}
    line = [str(N)]
    for name, be in libs:
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            outs[name] = be.attn_cross_split(q, ks, vs, N, bits, any_)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        line.append(f"[{name}] {min(ts[1:]):.1f} us")
    if len(outs) == 2:
        d = (outs["new"] - outs["base"]).abs().max() / outs["base"].abs().mean()
        line.append(f"max |new - base| / mean |base| = {float(d):.2e}")
    print("  ".join(line))

# the same level through ph_attn_cross_feat (no K / V operands): 64-channel features + position columns
from pasco_amd.graph.transformer import PositionEmbeddingSineSparse
pe = PositionEmbeddingSineSparse(128, normalize=True)
eps = pe.angle_model(torch.device("cuda"))[0]
for N in (210542, 45629):
    x = torch.randn(B * N, 64, generator=g).cuda()
    coords = torch.randint(0, 256, (B * N, 4), generator=g, dtype=torch.int32).cuda()
    q2 = (torch.randn(B, H, Q, 80, generator=g) * Dh ** -0.5).cuda()
    allow = (torch.rand(B * N, Q, generator=g) > 0.5).float().cuda()
    bits, any_ = hip.attn_mask_pack(allow, B, N)
    xs = hip.split_rows(x)
    ts, ta = [], []
    for _ in range(6):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        aug = hip.pos_aug(coords, eps, pe.TABLE_LO)
        e1.record()
        hip.attn_cross_feat(q2, xs, aug, N, bits, any_)
        e2.record()
        torch.cuda.synchronize()
        ta.append(e0.elapsed_time(e1) * 1e3)
        ts.append(e1.elapsed_time(e2) * 1e3)
    print(f"{N}  [feat] {min(ts[1:]):.1f} us  (+ pos_aug {min(ta[1:]):.1f} us)")
