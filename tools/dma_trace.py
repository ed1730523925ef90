"""Where does a workgroup of a k = 1 launch on k_conv_dma spend its life?  Shader-clock stamps (development hook: bit 7 of
ph_conv_dma_set_ablate, ph_dma_trace_read) of the first 64 workgroups of the mask-head launch (64 -> 100, 210 k rows, per-axis
table residual) and of the 128 -> 384 K / V projection of an S10 step.
    python tools/dma_trace.py > gpurun_out/dma_trace.txt"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pasco_amd.graph.synth import TeacherKeep, make_scene
from pasco_amd.me.backend import hip_backend

dev = torch.device("cuda", 0)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from devlib import use_dev_library   # noqa: E402
use_dev_library()      # the hooks below exist only in the development build (-DPH_DEV)
be = hip_backend()
net = bench.build_net(3, 283, dev)
scene = make_scene(0, n_infers=3).to(dev)
tk = TeacherKeep(scene, dev)
picked = {}
inner = be.conv_fwd


def spy(x, weight, nbr, n_out, **kw):
    out = inner(x, weight, nbr, n_out, **kw)
    cfg = be.conv_last_config()
    if cfg["kernel"] == 4 and nbr is None and cfg["ksplit"] == 1:
        w = weight if weight is not None else None
        cin, cout = (w.shape[-2], w.shape[-1]) if w is not None else (kw["wshape"][1], kw["wshape"][2])
        key = (cin, cout, n_out, kw.get("axis") is not None, kw.get("emit_split") is not None)
        picked.setdefault(key, (x, weight, nbr, n_out, dict(kw)))
    return out


with torch.no_grad():
    bench.run_scene(net, scene, tk)
    be.conv_fwd = spy
    bench.run_scene(net, scene, tk)
    be.conv_fwd = inner
lib = be.lib
lib.ph_conv_dma_set_ablate.argtypes = [C.c_int]
lib.ph_dma_trace_read.argtypes = [C.c_void_p]
names = ["start -> index table in LDS (kernel arguments, 128 entries, barrier)", "index table -> first DMA issued (address arithmetic)", "first stage landed (DMA latency)",
         "stages multiplied (-> epilogue entry)", "epilogue: every wave out of the main loop (barrier)",
         "epilogue: per-channel vectors + table / residual addends staged in LDS", "epilogue: arithmetic + stores, drained"]
for key, rec in sorted(picked.items(), key=lambda kv: -kv[0][2] * kv[0][1]):
    cin, cout, n_out, axis, emit = key
    x, weight, nbr, n, kw = rec
    ts = []
    for it in range(3):
        lib.ph_conv_dma_set_ablate(0x80 if it == 2 else 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        inner(x, weight, nbr, n, **kw)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    lib.ph_conv_dma_set_ablate(0)
    buf = np.zeros(64 * 8, dtype=np.uint64)
    assert lib.ph_dma_trace_read(buf.ctypes.data) == 0
    st = buf.reshape(64, 8).astype(np.int64)
    tiles = (n + 127) // 128 * ((cout + 127) // 128)
    tot = (st[:, 4] - st[:, 0]).astype(float)
    print(f"k1 {cin} -> {cout}, n = {n}, table residual {axis}, operand emitted {emit}: {ts[1]:.1f} us plain, {ts[2]:.1f} us traced; "
          f"{tiles} tiles on 512 slots; workgroup life median {np.median(tot):.0f} clk (min {tot.min():.0f}, max {tot.max():.0f})")
    for nm, (a, b) in zip(names, [(0, 7), (7, 1), (1, 2), (2, 3), (3, 5), (5, 6), (6, 4)]):
        d = (st[:, b] - st[:, a]).astype(float)
        print(f"    {nm:62s} median {np.median(d):8.0f} clk ({100 * np.median(d) / np.median(tot):5.1f} %)  min {d.min():8.0f}  max {d.max():8.0f}")
