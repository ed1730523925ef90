import ctypes as C, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libspec2.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "spec2_bench.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so,
                           os.path.join(HERE, "spec2_bench.hip")])
lib = C.CDLL(so)
lib.ub_spec2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p]
span_rows = 48 * 1024 * 1024 // 128
src = torch.randint(0, 255, (48 * 1024 * 1024 + 4096,), dtype=torch.uint8, device="cuda")
sink = torch.empty(512 * 512 * 2, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
iters = 2000
names = {0: "matrix waves only (4 x 24 MFMA)", 1: "loader waves only (4 x 6 DMA, vmcnt(6))", 2: "both roles, free running",
         3: "both roles, barrier / iteration, counted wait", 4: "mixed: 8 waves x (3 DMA + 12 MFMA interleaved)",
         5: "mixed: 4 waves x (6 DMA + 24 MFMA interleaved)", 6: "both roles free + 16 ds_read_b128 in the matrix waves",
         7: "both roles, barrier + fragment reads", 8: "both roles free, REGISTER loads", 9: "both roles free, 2 loader waves x 12 DMA"}
for gather in (0, 1):
    for grid in (256, 512):
        for mode in range(10):
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); assert lib.ub_spec2(src.data_ptr(), sink.data_ptr(), iters, mode, gather, span_rows, grid, st) == 0; e1.record()
                torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
            us = min(ts[1:])
            wg = grid / 256
            print(f"{'gather' if gather else 'hot   '} grid {grid}  mode {mode} {names[mode]:58s} {us:9.1f} us = "
                  f"{us * 1e-6 * 2.4e9 / iters / wg:7.0f} clk(2.4GHz)/WG-iter/CU  {24 * 1024 * iters * grid / (us * 1e-6) / 1e12 if mode != 0 else 0:6.2f} TB/s", flush=True)
