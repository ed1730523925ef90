"""Gather-path micro-benchmark on real S10 kernel maps (tools/ubench/gather_bench.hip).

    python tools/ubench/run_gather_bench.py [out.json]

Maps: G1 (completed scene, 210 k rows, ~15 pairs / row) and U1 (all 8 children of the stride-2 voxels, 683 k rows,
parent-major order = the generative decoder's stride-1 level before pruning, ~21 pairs / row).  Operand rows of
64 channels (2 chunks of 128 B) and 256 channels on the stride-4 level.  Reports time, gathered GB/s and B/clk/CU."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    so = os.path.join(HERE, "libubench.so")
    src = os.path.join(HERE, "gather_bench.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-o", so, src])
    return so


def main():
    from pasco_amd.graph.synth import make_occupancy
    from pasco_amd.me.backend import hip_backend
    from pasco_amd.me.core import kernel_offsets
    lib = C.CDLL(build())
    lib.ub_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p,
                              C.c_int, C.c_int, C.c_int, C.c_void_p]
    be = hip_backend()
    dev = torch.device("cuda", 0)
    g1 = np.argwhere(make_occupancy(0))
    c1 = torch.from_numpy(np.concatenate([np.zeros((g1.shape[0], 1), np.int64), g1], 1)).int().to(dev).contiguous()
    maps = {}

    def kmap(coords, ts):
        tk, tv, _, _, _ = be.map_insert(coords, dedup=False)
        return be.nbr_build(coords, tk, tv, kernel_offsets(3, ts))

    maps["G1"] = (c1.shape[0], kmap(c1, 1))
    c2 = be.coords_floor(c1, 2)
    _, _, _, uq, nu = be.map_insert(c2)
    c2u = c2[uq.long()].contiguous()
    u1 = be.coords_expand(c2u, 1)
    maps["U1"] = (u1.shape[0], kmap(u1, 1))
    c4 = be.coords_floor(c1, 4)
    _, _, _, uq4, _ = be.map_insert(c4)
    c4u = c4[uq4.long()].contiguous()
    u2 = be.coords_expand(c4u, 2)
    maps["U2"] = (u2.shape[0], kmap(u2, 2))

    zero = torch.zeros(256, dtype=torch.uint8, device=dev)
    sink = torch.empty(8192 * 256 * 16, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    results = []
    variants = [(0, 1, 0), (0, 2, 0), (0, 4, 0), (0, 1, 40), (0, 2, 40), (0, 4, 40), (0, 2, 80), (0, 4, 80),
                (1, 1, 0), (1, 2, 0), (1, 4, 0), (1, 2, 40), (1, 4, 40),
                (2, 1, 0), (2, 2, 0), (2, 3, 0), (2, 4, 0), (2, 2, 80), (2, 4, 80)]
    for name, cin in (("U1", 64), ("G1", 64), ("U2", 128)):
        n, nbr = maps[name]
        nch = cin // 32
        rs = cin * 4
        table = torch.randint(0, 255, (n, rs), dtype=torch.uint8, device=dev)
        pairs = int((nbr >= 0).sum())
        gbytes = pairs * rs
        for mode, depth, pad in variants:
            def run():
                rc = lib.ub_gather(table.data_ptr(), nbr.data_ptr(), n, 27, nch, rs, sink.data_ptr(), zero.data_ptr(),
                                   mode, depth, pad, stream)
                assert rc == 0, (rc, mode, depth, pad)
            run()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            us = float(np.median(ts))
            r = dict(map=name, rows=n, cin=cin, pairs_per_row=round(pairs / n, 2), mode=("line", "frag", "lds")[mode],
                     depth=depth, lds_pad_kb=pad, us=round(us, 1), gathered_GBps=round(gbytes / us / 1e3, 1),
                     B_per_clk_per_CU_at_2p1GHz=round(gbytes / (us * 1e-6) / 256 / 2.1e9, 2))
            print(r, flush=True)
            results.append(r)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
