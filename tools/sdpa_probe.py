"""Is torch's fused scaled_dot_product_attention (AOTriton `attn_fwd` on this ROCm build) fp32-exact?  The query
self-attention of the mask transformer ([M, 100, 384], 8 heads) through nn.MultiheadAttention's fast path, through the
written-out matmul form the graph uses since round 5, and in fp64.

    python tools/sdpa_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasco_amd.graph.transformer import SelfAttentionLayer

torch.manual_seed(0)
dev = torch.device("cuda", 0)
layer = SelfAttentionLayer(384, 8).eval().to(dev)
x = torch.randn(3, 100, 384, device=dev)
qp = torch.randn(3, 100, 384, device=dev)
with torch.no_grad():
    ours = layer(x, qp)
    qk = x + qp
    fast = layer.norm(x + layer.self_attn(qk, qk, value=x, need_weights=False)[0])
    l64 = SelfAttentionLayer(384, 8).eval().to(dev).double()
    l64.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
    ref = l64(x.double(), qp.double())
scale = float(ref.abs().mean())
for name, y in (("written-out matmuls (the graph)", ours), ("nn.MultiheadAttention fast path (SDPA)", fast)):
    print(f"{name:42s}: max |error| vs fp64 / mean |y| = {float((y.double() - ref).abs().max()) / scale:.2e}")
# larger logits (a trained net's attention is sharper than a random one's)
with torch.no_grad():
    for gain in (4.0, 16.0):
        xs = x * gain
        a = layer(xs, qp)
        f = layer.norm(xs + layer.self_attn(xs + qp, xs + qp, value=xs, need_weights=False)[0])
        r = l64(xs.double(), qp.double())
        s = float(r.abs().mean())
        print(f"input gain {gain:4.0f}: written-out {float((a.double() - r).abs().max()) / s:.2e}   SDPA fast path "
              f"{float((f.double() - r).abs().max()) / s:.2e}")
