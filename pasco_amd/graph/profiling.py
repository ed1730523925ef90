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
                                     cout=weight.shape[-1], kvol=kvol))
            return out

        backend.conv_fwd = conv_fwd
        backend._conv_profiled = True

    def summary(self, kvol_filter=27):
        """Aggregate over the recorded launches with kernel volume `kvol_filter` (k=3 convs, the
        dominant kernel): average duration, algorithmic flops / bytes per launch."""
        torch.cuda.synchronize()
        pair_cache = {}
        t = flops = b_alg = b_min = 0.0
        n = 0
        t_all = 0.0
        for r in self.records:
            dt = r["e0"].elapsed_time(r["e1"]) * 1e-3
            t_all += dt
            if r["kvol"] != kvol_filter:
                continue
            nbr = r["nbr"]
            key = nbr.data_ptr()
            if key not in pair_cache:
                pair_cache[key] = int((nbr >= 0).sum().item())
            P = pair_cache[key]
            cin, cout, n_out, n_in = r["cin"], r["cout"], r["n_out"], r["n_in"]
            flops += 2.0 * P * cin * cout
            b_alg += 4.0 * P * cin + 4.0 * n_out * cout + 8.0 * P + 4.0 * r["kvol"] * cin * cout
            b_min += 4.0 * n_in * cin + 4.0 * n_out * cout + 8.0 * P + 4.0 * r["kvol"] * cin * cout
            t += dt
            n += 1
        return dict(launches=n, time_s=t, flops=flops, bytes_alg=b_alg, bytes_min=b_min,
                    all_conv_launches=len(self.records), all_conv_time_s=t_all)
