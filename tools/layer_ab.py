"""A/B harness for convolution-kernel variants on the launches of ONE benchmark step.

    python tools/layer_ab.py [out.txt] [mask ...]         (masks: ints for ph_conv_dma_set_ablate, default "0 1")

Every distinct convolution launch of a step (kernel volume, channels, rows, flags) is captured with its operands and
replayed alone under each variant (min of 5 timed launches, HIP events); per layer the launches per step are counted so the
last line is the variant's convolution time per step.

LAYER_AB_BASE=<other libpascohip .so> (tools/build_base_lib.sh builds one from a git revision): every launch is also replayed
through that library in the same process, interleaved - column [base]; run-to-run differences between GPU boxes (several %)
do not enter the comparison.  LAYER_AB_PRODUCT=1: the working tree's PRODUCT library (no development hooks, mask 0) against the base."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import TeacherKeep, make_scene
from pasco_amd.me.backend import CBackend, hip_backend

out_path = sys.argv[1] if len(sys.argv) > 1 else None
masks = [int(v, 0) for v in sys.argv[2:]] or [0, 1]
min_us = float(os.environ.get("LAYER_AB_MIN_US", "40"))
n_infers = int(os.environ.get("LAYER_AB_M", "3"))
dev = torch.device("cuda", 0)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from devlib import use_dev_library   # noqa: E402
PRODUCT = os.environ.get("LAYER_AB_PRODUCT", "0") != "0"     # the working tree's PRODUCT library against LAYER_AB_BASE (mask 0 only)
if PRODUCT:
    masks = [0]
else:
    use_dev_library()      # the ablation hooks exist only in the development build (-DPH_DEV)
be = hip_backend()
net = bench.build_net(n_infers, 283, dev)
scene = make_scene(0, n_infers=n_infers).to(dev)
tk = TeacherKeep(scene, dev)
layers = {}
inner = be.conv_fwd


def spy(x, weight, nbr, n_out, **kw):
    out = inner(x, weight, nbr, n_out, **kw)
    cfg = be.conv_last_config()
    if weight is not None:
        w = weight if weight.dim() == 3 else weight[None]
        shape = tuple(w.shape)
    else:
        shape = tuple(kw["wshape"])
    key = (shape, n_out, cfg["kernel"], cfg["bn"], cfg["ksplit"], kw.get("emit_split") is not None, kw.get("axis") is not None,
           kw.get("residual") is not None)
    rec = layers.get(key)
    if rec is None:
        layers[key] = [1, (x, weight, nbr, n_out, dict(kw))]
    else:
        rec[0] += 1
    return out


with torch.no_grad():
    bench.run_scene(net, scene, tk)      # warm-up (operand caches, maps)
    be.conv_fwd = spy
    bench.run_scene(net, scene, tk)
    be.conv_fwd = inner
lib = be.lib
if PRODUCT:
    class _NoHooks:
        @staticmethod
        def ph_conv_dma_set_ablate(mask):
            pass
    lib = _NoHooks
else:
    lib.ph_conv_dma_set_ablate.argtypes = [C.c_int]
base = None
if os.environ.get("LAYER_AB_BASE"):
    base = CBackend(os.path.abspath(os.environ["LAYER_AB_BASE"]), "ph_", "cuda")
    masks = ["base"] + masks
KN = {0: "mfma", 1: "f16x3", 2: "h2", 3: "rl", 4: "dma", 5: "win|dma", 6: "wide", 7: "lin", 8: "grid"}


def timed(rec, mask):
    x, weight, nbr, n_out, kw = rec
    fwd = inner
    if mask == "base":
        fwd = base.conv_fwd
    else:
        lib.ph_conv_dma_set_ablate(mask)
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fwd(x, weight, nbr, n_out, **kw)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    lib.ph_conv_dma_set_ablate(0)
    return min(ts[1:])


lines = []
tot = {m: 0.0 for m in masks}
rows = []
for key, (cnt, rec) in layers.items():
    t = {m: timed(rec, m) for m in masks}
    rows.append((cnt * t[masks[0]], key, cnt, t))
for _, key, cnt, t in sorted(rows, key=lambda r: -r[0]):
    shape, n_out, kern, bn, ksplit, emit, axis, res = key
    for m in masks:
        tot[m] += cnt * t[m]
    if t[masks[0]] < min_us:
        continue
    flags = ("E" if emit else "-") + ("A" if axis else "-") + ("R" if res else "-")
    line = f"k{shape[0]:<3d} {shape[1]:3d}->{shape[2]:<3d} n={n_out:7d} {KN.get(kern, kern):7s} bn={bn:3d} ks={ksplit} {flags} x{cnt:2d}  " + \
        "  ".join(f"[{m if m == 'base' else hex(m)}] {t[m]:7.1f}" for m in masks)
    print(line, flush=True)
    lines.append(line)
line = "conv us/step: " + "  ".join(f"[{m if m == 'base' else hex(m)}] {tot[m]:9.1f}" for m in masks)
print(line)
lines.append(line)
if out_path:
    open(out_path, "w").write("\n".join(lines) + "\n")
