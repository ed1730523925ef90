import ctypes as C, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libspec.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so,
                       os.path.join(HERE, "spec_bench.hip")])
lib = C.CDLL(so)
lib.ub_spec.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
src = torch.randint(0, 255, (64 * 65536 + 4096,), dtype=torch.uint8, device="cuda")
sink = torch.empty(256 * 512 * 2, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
iters = 2000
for grid in (256,):
    for mode, name in ((0, "mixed: every wave 3 DMA + 12 MFMA"), (1, "specialised: 4 loader waves x 6 DMA, 4 matrix waves x 24 MFMA"),
                       (2, "DMA only"), (3, "MFMA only"), (4, "specialised, matrix waves on SIMDs {0,2}, loaders on {1,3}"),
                       (5, "specialised, loaders use REGISTER loads"), (6, "mixed, REGISTER loads"),
                       (7, "specialised, loaders at s_setprio 3"), (8, "specialised, prio 3 + s_sleep between MFMAs"),
                       (9, "every wave 12 ds_read_b128 (issued first) + 12 MFMA"), (10, "every wave 12 ds_read_b128 only")):
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); assert lib.ub_spec(src.data_ptr(), sink.data_ptr(), iters, mode, grid, st) == 0; e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        us = min(ts[1:])
        wg_per_cu = grid / 256
        print(f"grid {grid} ({wg_per_cu:.0f} WG/CU)  {name:66s} {us:9.1f} us  = {us * 1e-6 * 2.1e9 / iters / wg_per_cu:7.0f} clk per workgroup-iteration per CU", flush=True)
