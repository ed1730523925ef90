"""Kernel time of the literal drop-in route (unfused MIMO-1 at S10, guarded module convolutions): run under
    rocprofv3 --kernel-trace --stats -- python tools/unfused_profile.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

row = bench.short_row(1, 283, 20, torch.device("cuda", 0), steps=6, unfused=True, me_conv="guarded")
print(row["scenes_per_s"], row["ms_per_step"])
