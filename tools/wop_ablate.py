"""Where k_conv_wop2's time goes: the 64-channel 3^3 launches of one step replayed on the DEVELOPMENT library (PASCO_WOP=1: one
tile per workgroup) with parts of the kernel switched off (ph_conv_dma_set_ablate; wrong results by design, timing only):
0x1 no MFMAs, 0x2 no fragment reads, 0x4 no weight loads, 0x8 no window DMA, 0x10 no reduction / epilogue.
    python tools/wop_ablate.py [out.txt] [mask ...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("PASCO_WOP", "1")
from devlib import use_dev_library   # noqa: E402
use_dev_library()
import bench                           # noqa: E402
from pasco_amd.graph.synth import TeacherKeep, make_scene   # noqa: E402
from pasco_amd.me.backend import hip_backend                # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else None
masks = [int(v, 0) for v in sys.argv[2:]] or [0, 0x1, 0x2, 0x4, 0x8, 0x10, 0x3, 0x7, 0xf, 0x1e, 0x1f]
dev = torch.device("cuda", 0)
be = hip_backend()
lib = be.lib
lib.ph_conv_dma_set_ablate.argtypes = [C.c_int]
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
layers = {}
inner = be.conv_fwd


def spy(x, weight, nbr, n_out, **kw):
    out = inner(x, weight, nbr, n_out, **kw)
    cfg = be.conv_last_config()
    shape = tuple(kw["wshape"]) if weight is None else tuple((weight if weight.dim() == 3 else weight[None]).shape)
    if cfg["kernel"] == 5 and shape[2] <= 64:
        key = (shape, n_out, kw.get("emit_split") is not None, kw.get("residual") is not None)
        rec = layers.get(key)
        if rec is None:
            layers[key] = [1, (x, weight, nbr, n_out, dict(kw))]
        else:
            rec[0] += 1
    return out


with torch.no_grad():
    bench.run_scene(net, scene, tk)
    be.conv_fwd = spy
    bench.run_scene(net, scene, tk)
    be.conv_fwd = inner


def timed(rec, mask):
    x, weight, nbr, n_out, kw = rec
    print(f"  mask {hex(mask)} ...", file=sys.stderr, flush=True)
    lib.ph_conv_dma_set_ablate(mask)
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        inner(x, weight, nbr, n_out, **kw)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    lib.ph_conv_dma_set_ablate(0)
    return min(ts[1:])


lines = ["masks: " + " ".join(hex(m) for m in masks)]
for key, (cnt, rec) in sorted(layers.items(), key=lambda kv: -kv[0][1] * kv[1][0])[:5]:
    shape, n_out, emit, res = key
    t = [timed(rec, m) for m in masks]
    lines.append(f"k{shape[0]} {shape[1]}->{shape[2]} n={n_out:7d} {'E' if emit else '-'}{'R' if res else '-'} x{cnt}: " +
                 "  ".join(f"[{hex(m)}] {v:6.1f}" for m, v in zip(masks, t)))
    print(lines[-1], flush=True)
if out_path:
    open(out_path, "w").write("\n".join(lines) + "\n")
