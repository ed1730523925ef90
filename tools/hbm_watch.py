"""Does the box's HBM stream rate hold under load?  For `secs` seconds: alternate a matrix-heavy launch (fp16 GEMM) with a timed
1 GiB device copy; print per-second min / median / max copy rate and the memory-clock state the driver reports.
    python tools/hbm_watch.py [secs]"""
import glob
import sys
import time

import torch

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
dev = torch.device("cuda", 0)
a = torch.empty(1 << 28, dtype=torch.float32, device=dev)       # 1 GiB
b = torch.empty_like(a)
x = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
y = torch.randn(8192, 8192, device=dev, dtype=torch.float16)


def mclk():
    out = []
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk"):
        try:
            out += [l.strip() for l in open(p) if "*" in l]
        except OSError:
            pass
    return ";".join(out) or "?"


t_end = time.time() + secs
sec = int(time.time())
rates = []
while time.time() < t_end:
    for _ in range(3):
        torch.matmul(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    rates.append(2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    if int(time.time()) != sec:
        r = sorted(rates)
        print(f"t={int(time.time()) - int(t_end - secs):3d}s  copies {len(r):4d}  GB/s min {r[0]:7.0f}  median {r[len(r) // 2]:7.0f}  max {r[-1]:7.0f}  mclk {mclk()}",
              flush=True)
        rates = []
        sec = int(time.time())
