"""Per-operator microbenchmark on the S10 scene (run on the GPU box):
   python tools/op_bench.py [out.json]
Reports time, algorithmic bytes/flops (SURVEY.md 8(d) formulas) and roofline fractions."""
import json
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasco_amd.me.backend import hip_backend
from pasco_amd.me.core import kernel_offsets
from pasco_amd.graph.synth import make_occupancy

HBM_PEAK = 8.0e12
F32_PEAK = 157.3e12


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/op_bench.json"
    be = hip_backend()
    occ = make_occupancy(0)
    g1 = np.argwhere(occ)
    res = []
    levels = {}
    for s, c in ((1, 64), (2, 128), (4, 256), (8, 256)):
        cs = np.unique(np.floor_divide(g1, s) * s, axis=0)
        coords = torch.from_numpy(np.concatenate([np.zeros((cs.shape[0], 1), np.int64), cs], 1)).int().cuda()
        levels[s] = (coords, c)
    # U4 = full 64x64x8 grid at stride 4
    xs = np.stack(np.meshgrid(np.arange(64), np.arange(64), np.arange(8), indexing="ij"), -1).reshape(-1, 3) * 4
    levels["U4"] = (torch.from_numpy(np.concatenate([np.zeros((xs.shape[0], 1), np.int64), xs], 1)).int().cuda(), 256)
    for name, (coords, c) in levels.items():
        n = coords.shape[0]
        ts = 4 if name == "U4" else name
        cfg_times = {}
        if len(sys.argv) > 2 and sys.argv[2] == "cfgs":
            tk0, tv0, _, _, _ = be.map_insert(coords, dedup=False)
            nbr0 = be.nbr_build(coords, tk0, tv0, kernel_offsets(3, ts))
            x0 = torch.randn(n, c, device="cuda")
            w0 = torch.randn(27, c, c, device="cuda") / np.sqrt(27 * c)
            o0 = torch.empty(n, c, device="cuda")
            for cfg in ("128", "64", "32"):
                os.environ["PASCO_CONV_CFG"] = cfg
                cfg_times[cfg] = round(timeit(lambda: be.conv_fwd(x0, w0, nbr0, n, out=o0), iters=10) * 1e6, 1)
            os.environ.pop("PASCO_CONV_CFG", None)
            sp0 = be.split_weight_f16(w0)
            for cfg in ("128,32", "128,64", "64,32", "64,64", "32,32", "32,64"):
                os.environ["PASCO_CONVH_CFG"] = cfg
                cfg_times["f16x3/" + cfg] = round(timeit(lambda: be.conv_fwd(x0, w0, nbr0, n, out=o0, split=sp0), iters=10) * 1e6, 1)
            os.environ.pop("PASCO_CONVH_CFG", None)
            for cfg in ("128", "64", "32"):
                os.environ["PASCO_CONV_CFG"] = cfg
                cfg_times[cfg] = round(timeit(lambda: be.conv_fwd(x0, w0, nbr0, n, out=o0), iters=10) * 1e6, 1)
            os.environ.pop("PASCO_CONV_CFG", None)
            sp0 = be.split_weight_f16(w0)
            for cfg in ("128,32", "128,64", "64,32", "64,64", "32,32", "32,64"):
                os.environ["PASCO_CONVH_CFG"] = cfg
                cfg_times["f16x3/" + cfg] = round(timeit(lambda: be.conv_fwd(x0, w0, nbr0, n, out=o0, split=sp0), iters=10) * 1e6, 1)
            os.environ.pop("PASCO_CONVH_CFG", None)
        t_ins = timeit(lambda: be.map_insert(coords, dedup=False))
        t_ins_d = timeit(lambda: be.map_insert(coords, dedup=True))
        tk, tv, _, _, _ = be.map_insert(coords, dedup=False)
        offs = kernel_offsets(3, ts)
        t_nbr = timeit(lambda: be.nbr_build(coords, tk, tv, offs))
        nbr = be.nbr_build(coords, tk, tv, offs)
        P = int((nbr >= 0).sum().item())
        x = torch.randn(n, c, device="cuda")
        w = torch.randn(27, c, c, device="cuda") / np.sqrt(27 * c)
        out = torch.empty(n, c, device="cuda")
        t_conv = timeit(lambda: be.conv_fwd(x, w, nbr, n, out=out), iters=10)
        ps = torch.rand(c, device="cuda")
        sp = be.split_weight_f16(w)
        t_conv_h = timeit(lambda: be.conv_fwd(x, w, nbr, n, out=out, split=sp), iters=10)
        t_conv_hf = timeit(lambda: be.conv_fwd(x, w, nbr, n, out=out, split=sp, pro_scale=torch.ones(c, device="cuda"),
                                               pro_shift=torch.zeros(c, device="cuda"), pro_act=1, residual=x, res_act=1), iters=10)
        t_conv_f = timeit(lambda: be.conv_fwd(x, w, nbr, n, out=out, pro_scale=ps, pro_shift=ps, pro_act=1,
                                              epi_scale=ps, epi_shift=ps, epi_act=1, residual=x, res_act=1), iters=10)
        flop = 2.0 * P * c * c
        flop_dense = 2.0 * 27 * n * c * c
        b_alg = 4.0 * P * c + 4.0 * n * c + 8.0 * P + 4.0 * 27 * c * c
        b_min = 4.0 * n * c + 4.0 * n * c + 8.0 * P + 4.0 * 27 * c * c
        w1 = torch.randn(c, c, device="cuda")
        t_k1 = timeit(lambda: be.conv_fwd(x, w1, None, n, out=out))
        r = dict(level=str(name), n=n, c=c, cfg_us=cfg_times, pairs=P, pairs_per_voxel=P / n,
                 t_insert_us=t_ins * 1e6, t_insert_dedup_us=t_ins_d * 1e6, t_nbr_us=t_nbr * 1e6,
                 nbr_GBs=(16.0 * 2 * n + 8.0 * P) / t_nbr / 1e9,
                 t_conv3_us=t_conv * 1e6, t_conv3_f16x3_us=t_conv_h * 1e6, t_conv3_f16x3_fused_us=t_conv_hf * 1e6, t_conv3_fused_us=t_conv_f * 1e6,
                 conv3_TFLOPs=flop / t_conv / 1e12, conv3_TFLOPs_issued=flop_dense / t_conv / 1e12,
                 conv3_frac_f32_peak=flop / t_conv / F32_PEAK,
                 conv3_GBs_alg=b_alg / t_conv / 1e9, conv3_frac_hbm=b_alg / t_conv / HBM_PEAK,
                 conv3_GBs_min=b_min / t_conv / 1e9,
                 t_conv1_us=t_k1 * 1e6, conv1_GBs=(8.0 * n * c + 4 * c * c) / t_k1 / 1e9,
                 conv1_TFLOPs=2.0 * n * c * c / t_k1 / 1e12)
        print(json.dumps(r), flush=True)
        res.append(r)
    # stream copy ceiling
    a = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a), iters=10)
    res.append(dict(stream_copy_GBs=2 * a.numel() * 4 / t / 1e9))
    print(json.dumps(res[-1]))
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
