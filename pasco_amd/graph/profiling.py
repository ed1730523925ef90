"""Live per-launch timing of the sparse-conv kernel for bench.py's roofline object.

When enabled, every `conv_fwd` launch is bracketed by two HIP events recorded on the stream the
kernel is launched on (torch's current stream - the library enqueues on exactly that stream), and
its shape + neighbour table are remembered so that algorithmic FLOPs / bytes can be
computed after the timed region without adding work inside it.
"""
from __future__ import annotations

from typing import List

import torch

# 5: a window / gather pair (k_conv_win and k_conv_dma both launched, the device-side predicate lets one of them work)
KERNEL_NAMES = {0: "k_conv_mfma", 1: "k_conv_f16x3", 2: "k_conv_h2", 3: "k_conv_rl", 4: "k_conv_dma",
                5: "k_conv_win", 6: "k_conv_wide", 7: "k_conv_lin", 8: "k_conv_grid"}


def layer_class(kvol: int, cin: int, cout: int) -> str:
    """Layer classes of the per-class roofline table (bench.py): by kernel volume and width."""
    if kvol == 27:
        return f"k3 C={max(cin, cout)}"
    if kvol > 27:
        return "dense bottleneck (implicit GEMM, k > 27)"
    if kvol == 8:
        return "k2 strided / generative transposed"
    if kvol == 1:
        return "k1 tall (linear / 1x1)" if max(cin, cout) >= 256 else "k1 (1x1)"
    return f"k{kvol}"


class ConvProfiler:
    def __init__(self):
        self.records: List[dict] = []
        self.enabled = False

    def wrap(self, backend):
        """Monkey-patch one CBackend instance's conv_fwd with an event-recording shim."""
        if getattr(backend, "_conv_profiled", False):
            return
        inner = backend.conv_fwd
        prof = self

        def conv_fwd(x, weight, nbr, n_out, **kw):
            dev = x.device if x is not None else kw["in_split"].device
            if not prof.enabled or dev.type != "cuda":
                return inner(x, weight, nbr, n_out, **kw)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = inner(x, weight, nbr, n_out, **kw)
            e1.record()
            if weight is None:
                kvol, _, cout = kw["wshape"]
            else:
                kvol, cout = (1 if weight.dim() == 2 else weight.shape[0]), weight.shape[-1]
            n_in, cin = tuple(x.shape) if x is not None else kw["xshape"]
            kid = backend.conv_last_config()["kernel"]          # which kernel the library actually launched
            # the pair count of the launch's neighbour table: a device scalar computed once per table (outside the event
            # pair, no host read) and remembered ON the table - NOT the table itself: a record that kept the table alive kept
            # every kernel map of every profiled step alive, the allocator had to go to the driver for the next step's maps,
            # and those device allocations (0.1 - 5 ms each, box dependent) landed inside the event pairs of whichever
            # launches allocated large outputs: the "slow mode" of the k = 1 launches in rounds 2 - 3 (profiles/README.md)
            pairs = None
            if nbr is not None:
                pairs = getattr(nbr, "_ph_pairs", None)
                if pairs is None:
                    pairs = (nbr >= 0).sum()
                    try:
                        nbr._ph_pairs = pairs
                    except Exception:
                        pass
            name = KERNEL_NAMES.get(kid, f"kernel{kid}")
            if kid == 5 and cout <= 64:          # the 64-wide window launches run the offset-parallel kernel
                name = "k_conv_wop2"
            prof.records.append(dict(e0=e0, e1=e1, pairs=pairs, n_in=n_in, n_out=n_out, cin=cin,
                                     cout=cout, kvol=kvol, kernel=name))
            return out

        inner_split = backend.split_rows

        def split_rows(x, **kw):
            if not prof.enabled or x.device.type != "cuda":
                return inner_split(x, **kw)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = inner_split(x, **kw)
            e1.record()
            prof.records.append(dict(e0=e0, e1=e1, kernel="k_split_rows", n=x.shape[0], c=x.shape[1]))
            return out

        backend.conv_fwd = conv_fwd
        backend.split_rows = split_rows
        backend._conv_profiled = True

    def summary(self, by_class: bool = False):
        """Per kernel (`k_conv_dma` / `k_conv_h2` / `k_conv_f16x3` = split-precision products: LDS-DMA pipeline,
        register-staged, in-kernel split; `k_conv_mfma` = exact fp32 MFMA; `k_split_rows` = the operand split) over every
        recorded launch (k=3 / k=2 strided / generative transposed / k=1 convolutions and the dense bottleneck's
        implicit GEMMs): launches, time, algorithmic flops / bytes (flops = 2 P Cin Cout,
        B_alg = 4 P Cin + 4 N_out Cout + 8 P + 4 K Cin Cout, P = pairs of the neighbour table, P = N for
        identity maps; B_min = the same with 4 N_in Cin in place of 4 P Cin: the compulsory lower bound)."""
        torch.cuda.synchronize()
        pair_cache = {}
        out = {}
        classes = {}
        for r in self.records:
            dt = r["e0"].elapsed_time(r["e1"]) * 1e-3
            if r["kernel"] == "k_split_rows":      # operand preparation of mode 2: 4 B read + 4 B written per element
                d = out.setdefault("k_split_rows", dict(launches=0, time_s=0.0, flops=0.0, bytes_alg=0.0,
                                                        k3_launches=0, k3_time_s=0.0, k3_flops=0.0))
                d["launches"] += 1
                d["time_s"] += dt
                d["bytes_alg"] += 4.0 * r["n"] * r["c"] + 4.0 * r["n"] * ((r["c"] + 31) // 32 * 32)
                continue
            pairs = r["pairs"]
            if pairs is None:
                P = r["n_out"]
                idx_bytes = 0.0
            else:
                key = id(pairs)
                if key not in pair_cache:
                    pair_cache[key] = int(pairs.item())
                P = pair_cache[key]
                idx_bytes = 8.0 * P
            cin, cout, n_out = r["cin"], r["cout"], r["n_out"]
            d = out.setdefault(r["kernel"], dict(launches=0, time_s=0.0, flops=0.0, bytes_alg=0.0, bytes_min=0.0,
                                                 k3_launches=0, k3_time_s=0.0, k3_flops=0.0))
            fl = 2.0 * P * cin * cout
            # compulsory lower bound of SURVEY.md 8(d): every input row read ONCE instead of once per pair
            b_min = 4.0 * min(r["n_in"], P) * cin + 4.0 * n_out * cout + idx_bytes + 4.0 * r["kvol"] * cin * cout
            d["launches"] += 1
            d["time_s"] += dt
            d["flops"] += fl
            d["bytes_alg"] += 4.0 * P * cin + 4.0 * n_out * cout + idx_bytes + 4.0 * r["kvol"] * cin * cout
            d["bytes_min"] += b_min
            if r["kvol"] == 27:
                d["k3_launches"] += 1
                d["k3_time_s"] += dt
                d["k3_flops"] += fl
            c = classes.setdefault((layer_class(r["kvol"], cin, cout), r["kernel"]),
                                   dict(launches=0, time_s=0.0, flops=0.0, bytes_alg=0.0, bytes_min=0.0))
            c["launches"] += 1
            c["time_s"] += dt
            c["flops"] += fl
            c["bytes_alg"] += 4.0 * P * cin + 4.0 * n_out * cout + idx_bytes + 4.0 * r["kvol"] * cin * cout
            c["bytes_min"] += b_min
        if by_class:
            return out, classes
        return out
