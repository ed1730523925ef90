"""Box calibration: HBM copy rate and one fixed split-precision convolution, to normalise A/B timings taken on
different gpurun boxes (diagnostic only)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasco_amd.me.backend import hip_backend
from pasco_amd.me.core import kernel_offsets
from pasco_amd.graph.synth import make_occupancy


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


be = hip_backend()
a = torch.empty(256 << 20, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a), iters=10)
g1 = np.argwhere(make_occupancy(0))
coords = torch.from_numpy(np.ascontiguousarray(np.concatenate([np.zeros((g1.shape[0], 1), np.int64), g1], 1))).int().contiguous().cuda()
tk, tv, _, _, _ = be.map_insert(coords, dedup=False)
nbr = be.nbr_build(coords, tk, tv, kernel_offsets(3, 1))
n = coords.shape[0]
x = torch.randn(n, 64, device="cuda"); w = torch.randn(27, 64, 64, device="cuda") / 40
sp, xs = be.split_weight_rows(w), be.split_rows(x)
out = torch.empty(n, 64, device="cuda")
tc = timeit(lambda: be.conv_fwd(x, w, nbr, n, out=out, split=sp, in_split=xs))
print(f"calib: copy {2 * a.numel() * 4 / t / 1e6:.0f} GB/s   conv k27 64->64 n={n}: {tc * 1e3:.1f} us")
