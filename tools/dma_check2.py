import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import TeacherKeep, make_scene
from pasco_amd.me.backend import hip_backend
dev = torch.device("cuda", 0)
be = hip_backend()
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
inner = be.conv_fwd
cap = []
def spy(x, weight, nbr, n_out, **kw):
    out = inner(x, weight, nbr, n_out, **kw)
    if weight is not None and x is not None and kw.get("split") is not None and len(kw["split"]) == 2 and len(cap) < 6:
        cap.append((x, weight, nbr, n_out, kw["split"]))
    return out
be.conv_fwd = spy
with torch.no_grad():
    bench.run_scene(net, scene, tk)
be.conv_fwd = inner
for x, weight, nbr, n_out, split in cap:
    p2 = inner(x, weight, nbr, n_out, split=split)
    cfg = be.conv_last_config()
    s1 = be.split_weight_f16(weight)
    p1 = inner(x, weight, nbr, n_out, split=s1)
    s2b = be.split_weight_rows(weight)
    p2b = inner(x, weight, nbr, n_out, split=s2b)
    d = (p1 - p2).abs()
    print(tuple(weight.shape), n_out, cfg["bm"], cfg["bn"], "p1 vs p2 diff", int((d > 0).sum()), float(d.max()),
          "| p2 vs fresh-split p2b", int(((p2 - p2b).abs() > 0).sum()), "unscale", split[1], s1[2], s2b[1],
          "wsplit equal", torch.equal(split[0].view(torch.int16), s2b[0].view(torch.int16)),
          "x absmax", float(x.abs().max()), "x min nonzero", float(x[x != 0].abs().min()) if (x != 0).any() else 0)
    if (d > 0).any():
        idx = (d > 0).nonzero()
        print("   rows", idx[:5].tolist(), "cols unique", idx[:, 1].unique()[:10].tolist(), "n rows differing", idx[:, 0].unique().numel())
print("---- hypotheses")
x, weight, nbr, n_out, split = cap[0]
sp = be.split_rows(x)
xs = x * 32.0
hi = xs.half()
lo = (xs - hi.float()).half()
got = sp.reshape(x.shape[0], -1, 2, 32)
ghi = got[:, :, 0, :].reshape(x.shape[0], -1)[:, : x.shape[1]]
glo = got[:, :, 1, :].reshape(x.shape[0], -1)[:, : x.shape[1]]
print("split_rows hi == torch:", torch.equal(ghi.view(torch.int16), hi.view(torch.int16)), " lo == torch:", torch.equal(glo.view(torch.int16), lo.view(torch.int16)))
for thr in (0.0, 1e-6, 1e-4, 1e-2):
    xz = torch.where(x.abs() < thr, torch.zeros_like(x), x).contiguous()
    p2 = inner(xz, weight, nbr, n_out, split=split)
    p1 = inner(xz, weight, nbr, n_out, split=be.split_weight_f16(weight))
    d = (p1 - p2).abs()
    print(f"values below {thr:g} zeroed: differing {int((d > 0).sum())} max {float(d.max()):.3e}")
xr = torch.randn_like(x)
p2 = inner(xr, weight, nbr, n_out, split=split)
p1 = inner(xr, weight, nbr, n_out, split=be.split_weight_f16(weight))
print("randn x, real weights: differing", int(((p1 - p2).abs() > 0).sum()))
wr = torch.randn_like(weight) / 8
p2 = inner(x, wr, nbr, n_out, split=be.split_weight_rows(wr))
p1 = inner(x, wr, nbr, n_out, split=be.split_weight_f16(wr))
print("real x, randn weights: differing", int(((p1 - p2).abs() > 0).sum()))
xa = x.abs().contiguous()
p2 = inner(xa, weight, nbr, n_out, split=split); p1 = inner(xa, weight, nbr, n_out, split=be.split_weight_f16(weight))
print("|x|: differing", int(((p1 - p2).abs() > 0).sum()), " x has negatives:", bool((x < 0).any()), " frac zeros", float((x == 0).float().mean()))
print("---- weights")
x, weight, nbr, n_out, split = cap[0]
hi, lo, u1 = be.split_weight_f16(weight)
ws, u2 = be.split_weight_rows(weight)
cout, cin = hi.shape[1], hi.shape[2]
v = ws.reshape(cout, -1, 2, 32)
khi = v[:, :, 0, :].reshape(cout, -1)[:, :cin]
klo = v[:, :, 1, :].reshape(cout, -1)[:, :cin]
print("weights: hi equal", torch.equal(khi.view(torch.int16), hi[0].view(torch.int16)), "lo equal", torch.equal(klo.view(torch.int16), lo[0].view(torch.int16)),
      "n lo diff", int((klo.view(torch.int16) != lo[0].view(torch.int16)).sum()), "of", klo.numel())
wu = ((torch.rand_like(weight) - 0.5) * 0.25).contiguous()
p2 = inner(x, wu, nbr, n_out, split=be.split_weight_rows(wu)); p1 = inner(x, wu, nbr, n_out, split=be.split_weight_f16(wu))
print("uniform synthetic weights: differing", int(((p1 - p2).abs() > 0).sum()))
wc = weight.clone()
p2 = inner(x, wc, nbr, n_out, split=be.split_weight_rows(wc)); p1 = inner(x, wc, nbr, n_out, split=be.split_weight_f16(wc))
print("cloned real weights: differing", int(((p1 - p2).abs() > 0).sum()), "weight stride", weight.stride(), weight.is_contiguous(), weight.dtype, weight.shape)
# does the launch matter? run mode 1 twice
p1b = inner(x, wc, nbr, n_out, split=be.split_weight_f16(wc))
print("mode 1 repeatable:", torch.equal(p1, p1b))
f32 = inner(x, wc, nbr, n_out)
ref = x.double() @ wc.double()
print("err vs fp64: mode1 %.3e  mode2 %.3e  f32 %.3e" % (float((p1.double() - ref).abs().max()), float((p2.double() - ref).abs().max()), float((f32.double() - ref).abs().max())))
