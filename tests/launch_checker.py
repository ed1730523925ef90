"""Test helper: wraps one backend's `conv_fwd` and checks every launch in place against the CPU oracle (fp32
operands) and an fp64 gather-matmul on sampled output rows - see tests/test_hip_bench_shapes.py."""
import torch

N_SAMPLE = 2048


def unsplit(sp, n_rows, c, exp2=None):
    """[n_rows, cpad/32, 2, 32] f16 operand -> the fp32 values it stands for ((hi + lo) * 2^-exp2), [n_rows, c].
    exp2 = None: an activation operand (SPLIT_ACT_EXP2); weights pass 0."""
    from pasco_amd.me.backend import SPLIT_ACT_EXP2
    e = SPLIT_ACT_EXP2 if exp2 is None else exp2
    v = sp.reshape(n_rows, -1, 2, 32).float()
    return ((v[:, :, 0, :] + v[:, :, 1, :]).reshape(n_rows, -1)[:, :c] * 2.0 ** (-e)).contiguous()


def act64(v, act, slope):
    if act == 1:
        return torch.clamp_min(v, 0.0)
    if act == 2:
        return torch.where(v > 0, v, v * slope)
    return v


class LaunchChecker:
    """Wraps one CBackend's conv_fwd; checks every launch against the oracle and fp64 on sampled rows."""

    def __init__(self, hip, oracle):
        self.hip, self.oracle = hip, oracle
        self.inner = hip.conv_fwd
        self.oracle_fwd = oracle.conv_fwd      # bound before install(): the two may be the same library in the CPU self-test
        self.seen = {}          # instantiation key -> [launches, checked, worst error]
        self.mode1_done = set()
        self.gens = {}
        self.enabled = True

    def install(self):
        self.hip.conv_fwd = self

    def remove(self):
        self.hip.conv_fwd = self.inner

    def __call__(self, x, weight, nbr, n_out, **kw):
        out = self.inner(x, weight, nbr, n_out, **kw)
        if not self.enabled or n_out == 0:
            return out
        cfg = self.hip.conv_last_config()
        key = (cfg["kernel"], cfg["bm"], cfg["bn"], cfg["kc"], cfg["waves"], "ksplit" if cfg["ksplit"] > 1 else "direct",
               "emit" if cfg["emit"] else "plain")
        rec = self.seen.setdefault(key, [0, 0, 0.0, set()])
        rec[0] += 1
        err = self.check(x, weight, nbr, n_out, kw, out, cfg, key)
        rec[1] += 1
        rec[2] = max(rec[2], err)
        return out

    # ------------------------------------------------------------------------------------------------
    def check(self, x, weight, nbr, n_out, kw, out, cfg, key):
        hip, oracle = self.hip, self.oracle
        split = kw.get("split")
        in_split = kw.get("in_split")
        emit = kw.get("emit_split")
        slope = kw.get("slope", 0.01)
        if weight is not None:
            w = weight if weight.dim() == 3 else weight[None]
            kvol, cin, cout = w.shape
        else:                                    # only the operand exists: undo split_weight_rows
            kvol, cin, cout = kw["wshape"]
            w_split, unscale = split
            w = unsplit(w_split, kvol * cout, cin, exp2=0).reshape(kvol, cout, cin).transpose(1, 2).contiguous() * unscale
        dev = w.device
        fp32_x = x is not None and not x.is_meta
        n_in = x.shape[0] if fp32_x else kw["xshape"][0]
        S = min(N_SAMPLE, n_out)
        gen = self.gens.get(dev)
        if gen is None:
            gen = self.gens[dev] = torch.Generator(device=dev).manual_seed(99)
        rows = torch.randperm(n_out, device=dev, generator=gen)[:S] if n_out > S else torch.arange(n_out, device=dev)
        nb = nbr[:, rows].long() if nbr is not None else rows[None].long()
        valid = nb >= 0
        uniq, inv = torch.unique(nb[valid], return_inverse=True)
        nb_sub = torch.full_like(nb, -1)
        nb_sub[valid] = inv
        ps, pb, pact = kw.get("pro_scale"), kw.get("pro_shift"), kw.get("pro_act", 0)
        if fp32_x:
            x_raw = x[uniq].contiguous()
            xp = x_raw
            if ps is not None or pb is not None or pact != 0:       # fp32, separate multiply and add (as ph_split_rows)
                xp = xp * ps if ps is not None else xp
                xp = xp + pb if pb is not None else xp
                xp = torch.relu(xp) if pact == 1 else (torch.where(xp > 0, xp, xp * slope) if pact == 2 else xp)
        else:                                    # rows that exist only in operand form (prologue already applied)
            x_raw = xp = unsplit(in_split[uniq], uniq.shape[0], cin)
        # (b) fp64 gather-matmul
        acc = torch.zeros(S, cout, dtype=torch.float64, device=dev)
        xd = torch.cat([xp.double(), torch.zeros(1, cin, dtype=torch.float64, device=dev)])
        for k in range(kvol):
            acc += xd[nb_sub[k]] @ w[k].double()          # index -1 = the zero row
        # epilogue in fp64; `mag` = mean magnitude of the last pre-activation quantity = the error yardstick
        ref = acc
        if kw.get("bias") is not None:
            ref = ref + kw["bias"].double()
        if kw.get("epi_scale") is not None:
            ref = ref * kw["epi_scale"].double()
        if kw.get("epi_shift") is not None:
            ref = ref + kw["epi_shift"].double()
        mag = float(ref.abs().mean())
        ref = act64(ref, kw.get("epi_act", 0), slope)
        tail = any(kw.get(k) is not None for k in ("residual", "epi2_scale", "epi2_shift", "axis")) or kw.get("res_act", 0) != 0
        if tail:
            if kw.get("epi2_scale") is not None:
                ref = ref * kw["epi2_scale"].double()
            if kw.get("epi2_shift") is not None:
                ref = ref + kw["epi2_shift"].double()
            if kw.get("axis") is not None:          # per-axis table rows (the position encoding of the transformer)
                tab, acoords, lo = kw["axis"]
                ai = (acoords[rows][:, 1:4].long() - lo).clamp(0, tab.shape[1] - 1)
                ref = ref + tab[0][ai[:, 0]].double() + tab[1][ai[:, 1]].double() + tab[2][ai[:, 2]].double()
            if kw.get("residual") is not None:
                ref = ref + kw["residual"][rows].double()
            mag = max(mag, float(ref.abs().mean()))
            ref = act64(ref, kw.get("res_act", 0), slope)
        scale = mag + 1e-12
        got_out, got_split = (out if emit is not None else (out, None))
        worst = 0.0
        if got_out is not None:
            err = float((got_out[rows].double() - ref).abs().max()) / scale
            assert err < 3e-4, f"{key} k{kvol} {cin}->{cout} n={n_out}: fp64 error {err:.2e} of mean |y|"
            worst = err
        # (a) the oracle on the same sub-problem, fp32 operands, exact fp32 arithmetic
        okw = {}
        for name in ("bias", "epi_scale", "epi_shift", "epi2_scale", "epi2_shift"):
            if kw.get(name) is not None:
                okw[name] = kw[name].cpu()
        for name in ("epi_act", "res_act", "slope"):
            if name in kw:
                okw[name] = kw[name]
        if kw.get("residual") is not None:
            okw["residual"] = kw["residual"][rows].cpu().contiguous()
        if kw.get("axis") is not None:
            tab, acoords, lo = kw["axis"]
            okw["axis"] = (tab.cpu(), acoords[rows].cpu().contiguous(), lo)
        if fp32_x:
            okw.update(pro_scale=None if ps is None else ps.cpu(), pro_shift=None if pb is None else pb.cpu(), pro_act=pact)
        exp = self.oracle_fwd(x_raw.cpu(), w.cpu().contiguous(), nb_sub.int().cpu().contiguous(), S, **okw)
        if got_out is not None:
            g = got_out[rows].cpu()
            tol = 3e-4 * scale
            bad = (g - exp).abs() > (tol + 1e-3 * exp.abs())
            assert not bool(bad.any()), f"{key} k{kvol} {cin}->{cout} n={n_out}: oracle mismatch " \
                                        f"{float((g - exp).abs().max()):.3e} (scale {scale:.3e})"
        # (d) the emitted operand
        if got_split is not None:
            osc, osh, oact = emit
            if got_out is not None:
                want = hip.split_rows(got_out, pro_scale=osc, pro_shift=osh, pro_act=oact, slope=slope)
                assert torch.equal(got_split.view(torch.int16), want.view(torch.int16)), f"{key}: emitted operand"
            else:
                r2 = ref
                if osc is not None:
                    r2 = r2 * osc.double()
                if osh is not None:
                    r2 = r2 + osh.double()
                mag2 = float(r2.abs().mean()) + 1e-12
                r2 = act64(r2, oact, slope)
                gs = unsplit(got_split[rows], S, cout).double()
                err = float((gs - r2).abs().max()) / mag2
                assert err < 3e-4, f"{key} k{kvol} {cin}->{cout} n={n_out}: emitted operand error {err:.2e}"
                worst = max(worst, err)
        # (c) bit-for-bit against the in-kernel split (mode 1), once per instantiation and layer shape, in PLAIN form
        # (no prologue / epilogue: their FMA contraction is each kernel's own business; the products and the
        # accumulation order are what must agree): same x, weights and map through both kernels
        shape_key = key + (kvol, cin, cout)
        if (fp32_x and weight is not None and cfg["mma_mode"] == 2 and cfg["ksplit"] == 1 and cfg["kernel"] != 5
                and shape_key not in self.mode1_done):       # kernel 5 = window / gather pair: other summation order
            self.mode1_done.add(shape_key)
            p2 = self.inner(x, weight, nbr, n_out, split=split, rowlist=kw.get("rowlist"))     # same kernel, plain form
            cfg2 = hip.conv_last_config()
            assert (cfg2["kernel"], cfg2["bm"], cfg2["bn"], cfg2["kc"]) == (cfg["kernel"], cfg["bm"], cfg["bn"], cfg["kc"])
            p1 = self.inner(x, weight, nbr, n_out, split=hip.split_weight_f16(weight))
            cfg1 = hip.conv_last_config()
            assert cfg1["mma_mode"] == 1
            if cfg1["ksplit"] == 1:
                assert torch.equal(p1, p2), f"{key} k{kvol} {cin}->{cout} n={n_out}: differs from the in-kernel split " \
                                            f"(max {float((p1 - p2).abs().max()):.3e})"
                self.seen[key][3].add((kvol, cin, cout))
            else:
                # the mode-1 kernel splits this map over the kernel offsets where the checked kernel (k_conv_wide, one
                # workgroup per CU) does not: other fp32 summation order, not bit-comparable - fp32 reorder noise only
                d = float((p1 - p2).abs().max()) / (float(p1.abs().mean()) + 1e-12)
                assert d < 1e-4, f"{key} k{kvol} {cin}->{cout} n={n_out}: in-kernel split (ksplit {cfg1['ksplit']}) differs " \
                                 f"by {d:.2e} of mean |y|"
        return worst
