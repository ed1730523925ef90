"""Live per-launch timing of the sparse-conv kernel for bench.py's roofline object.

When enabled, every `conv_fwd` launch is bracketed by two HIP events recorded on the stream the
kernel is launched on (torch's current stream - the library enqueues on exactly that stream), and
its shape + neighbour table are remembered so that algorithmic FLOPs / bytes can be
computed after the timed region without adding work inside it.
"""
from __future__ import annotations

from typing import List

import torch


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
            if not prof.enabled or x.device.type != "cuda":
                return inner(x, weight, nbr, n_out, **kw)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = inner(x, weight, nbr, n_out, **kw)
            e1.record()
            kvol = 1 if weight.dim() == 2 else weight.shape[0]
            prof.records.append(dict(e0=e0, e1=e1, nbr=nbr, n_in=x.shape[0], n_out=n_out, cin=x.shape[1],
                                     cout=weight.shape[-1], kvol=kvol,
                                     kernel="k_conv_f16x3" if kw.get("split") is not None else "k_conv_mfma"))
            return out

        backend.conv_fwd = conv_fwd
        backend._conv_profiled = True

    def summary(self):
        """Per kernel (`k_conv_f16x3` = split-precision products, `k_conv_mfma` = exact fp32 MFMA) over every
        recorded launch (k=3 / k=2 strided / generative transposed / k=1 convolutions and the dense bottleneck's
        implicit GEMMs): launches, time, algorithmic flops / bytes (flops = 2 P Cin Cout,
        B_alg = 4 P Cin + 4 N_out Cout + 8 P + 4 K Cin Cout, P = pairs of the neighbour table, P = N for
        identity maps)."""
        torch.cuda.synchronize()
        pair_cache = {}
        out = {}
        for r in self.records:
            dt = r["e0"].elapsed_time(r["e1"]) * 1e-3
            nbr = r["nbr"]
            if nbr is None:
                P = r["n_out"]
                idx_bytes = 0.0
            else:
                key = nbr.data_ptr()
                if key not in pair_cache:
                    pair_cache[key] = int((nbr >= 0).sum().item())
                P = pair_cache[key]
                idx_bytes = 8.0 * P
            cin, cout, n_out = r["cin"], r["cout"], r["n_out"]
            d = out.setdefault(r["kernel"], dict(launches=0, time_s=0.0, flops=0.0, bytes_alg=0.0,
                                                 k3_launches=0, k3_time_s=0.0, k3_flops=0.0))
            fl = 2.0 * P * cin * cout
            d["launches"] += 1
            d["time_s"] += dt
            d["flops"] += fl
            d["bytes_alg"] += 4.0 * P * cin + 4.0 * n_out * cout + idx_bytes + 4.0 * r["kvol"] * cin * cout
            if r["kvol"] == 27:
                d["k3_launches"] += 1
                d["k3_time_s"] += dt
                d["k3_flops"] += fl
        return out
