import sys, torch
sys.path.insert(0, ".")
from pasco_amd.me.backend import hip_backend
hip = hip_backend()
g = torch.Generator().manual_seed(0)
B, H, Q, Dh = 3, 8, 100, 48
for N in (210542, 45629, 9396):
    q = (torch.randn(B, H, Q, Dh, generator=g) * Dh ** -0.5).cuda()
    k = torch.randn(B * N, H * Dh, generator=g).cuda()
    v = torch.randn(B * N, H * Dh, generator=g).cuda()
    allow = (torch.rand(B * N, Q, generator=g) > 0.5).float().cuda()
    bits, any_ = hip.attn_mask_pack(allow, B, N)
    ks, vs = hip.split_rows(k), hip.split_rows(v)
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.attn_cross_split(q, ks, vs, N, bits, any_); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(N, f"{min(ts[1:]):.1f} us")
