"""The literal drop-in route (reference-style module sequence on the plain pasco_amd.me modules), MIMO-1 at S10: the guarded
split path against the exact fp32 kernel only.    python tools/unfused_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
for mode in ("guarded", "exact", "guarded"):
    row = bench.short_row(1, 283, 20, dev, steps=4, unfused=True, me_conv=mode)
    print(f"unfused MIMO-1, pasco_amd.me convolutions {mode:8s}: {row['scenes_per_s']:.2f} scenes/s ({row['ms_per_step']:.2f} ms)")
row = bench.short_row(1, 283, 20, dev, steps=4)
print(f"fused MIMO-1 (pasco_amd.graph)                        : {row['scenes_per_s']:.2f} scenes/s ({row['ms_per_step']:.2f} ms)")
