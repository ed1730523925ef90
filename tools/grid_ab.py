"""The dense bottleneck's implicit GEMMs on k_conv_grid (dense-grid promise, LDS windows) against the gather kernels
(PH_ROUTE_GRID_NEVER: k_conv_dma) - same process, same operands, alternating.  python tools/grid_ab.py [out.txt]
GRID_ABLATE=mask[,mask...]: the development library's phase ablations of k_conv_grid (conv_grid.hip) instead of the A/B."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pasco_amd.me.backend import ROUTE_GRID_NEVER, backend_for  # noqa: E402


def main():
    masks = [int(m, 0) for m in os.environ.get("GRID_ABLATE", "").split(",") if m]
    if masks or os.environ.get("PASCO_GRID_KS"):          # PASCO_GRID_KS=n: the development library with n slices by hand
        from devlib import use_dev_library
        use_dev_library()
    dev = torch.device("cuda:0")
    be = backend_for(dev)
    lines = []
    for dims in ((1, 38, 44, 4),):
        b, x, y, z = dims
        ax = [torch.arange(n, dtype=torch.int32, device=dev) for n in (b, z, x, y)]
        bzxy = torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1).reshape(-1, 4)
        coords = bzxy[:, [0, 2, 3, 1]].contiguous()
        n = coords.shape[0]
        tk, tv, _, _, _ = be.map_insert(coords, dedup=False)
        for ks in ((7, 7, 5), (5, 5, 3), (3, 3, 1)):
            offs = be.grid_offsets(ks)
            nbr = torch.cat([be.nbr_build(coords, tk, tv, offs[i:i + 64]) for i in range(0, len(offs), 64)], dim=0).contiguous()
            kvol = len(offs)
            xin = torch.randn(n, 256, device=dev)
            w = torch.randn(kvol, 256, 256, device=dev) / (kvol * 256) ** 0.5
            split, xs = be.split_weight_rows(w), be.split_rows(xin)
            if masks:
                parts = []
                for m in masks:
                    be.lib.ph_conv_grid_set_ablate(m)
                    for _ in range(3):
                        be.conv_fwd(xin, w, nbr, n, split=split, in_split=xs, grid=(dims, ks), epi_act=1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        be.conv_fwd(xin, w, nbr, n, split=split, in_split=xs, grid=(dims, ks), epi_act=1)
                    e1.record()
                    torch.cuda.synchronize()
                    parts.append(f"[{m:#x}] {e0.elapsed_time(e1) / 20 * 1e3:7.1f}")
                be.lib.ph_conv_grid_set_ablate(0)
                line = f"dims={dims} k={ks} n={n:6d} 256->256 ablate " + "  ".join(parts)
                print(line, flush=True)
                lines.append(line)
                continue
            res = {}
            for name, route in (("grid", 0), ("gather", ROUTE_GRID_NEVER), ("grid", 0), ("gather", ROUTE_GRID_NEVER)):
                with be.routing(route):
                    for _ in range(3):
                        be.conv_fwd(xin, w, nbr, n, split=split, in_split=xs, grid=(dims, ks), epi_act=1)
                    cfg = be.conv_last_config()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        be.conv_fwd(xin, w, nbr, n, split=split, in_split=xs, grid=(dims, ks), epi_act=1)
                    e1.record()
                    torch.cuda.synchronize()
                res.setdefault(name, []).append((e0.elapsed_time(e1) / 20 * 1e3, cfg["kernel"], cfg["ksplit"]))
            line = f"dims={dims} k={ks} n={n:6d} 256->256 " + "  ".join(
                f"{k}: " + " / ".join(f"{t:7.1f} us (kernel {kid}, {s} slices)" for t, kid, s in v) for k, v in res.items())
            print(line, flush=True)
            lines.append(line)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("# launch + split reduction, 20 back-to-back launches per figure; kernel 8 = k_conv_grid, 4 = k_conv_dma\n")
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
