"""The tile epilogue's compile-time instances (conv_h2_common.h, round 3): activations on / off, operand prologue on / off,
tail (second BN, dense residual, per-axis table residual) on / off, with and without operand emission - on the staged form
(k_conv_dma: per-channel vectors and tail addends through LDS) and the plain form (64-wide tiles), k = 1 and k = 27.
Every combination against the oracle; the emitted operand bit for bit against ph_split_rows of the fp32 result.
Reference layers: conv + BN + ReLU (+ residual) blocks (mink.py:625-638), K / V projections with the position table
(transformer_predictor_v2.py cross-attention), mask heads (decoder_v3.py:267-282)."""
import itertools

import pytest
import torch

from pasco_amd.me.core import kernel_offsets
from tests.test_hip_wide import scene

pytestmark = pytest.mark.gpu

TAILS = ("none", "bn2", "residual", "axis", "residual+axis")


@pytest.mark.parametrize("cout,kind", [(128, "k1"), (128, "k3"), (64, "k3"), (256, "k1")])
def test_epilogue_instances_match_oracle(hip, oracle, cout, kind):
    n, cin = 3001, 64
    coords = scene(n, (24, 24, 10), 7)
    g = torch.Generator().manual_seed(cout + len(kind))
    x = torch.randn(n, cin, generator=g)
    if kind == "k1":
        nbr = None
        w = torch.randn(1, cin, cout, generator=g) / cin ** 0.5
    else:
        offs = kernel_offsets(3, 1)
        tk, tv, _, _, _ = oracle.map_insert(coords.contiguous(), dedup=False)
        nbr = oracle.nbr_build(coords.contiguous(), tk, tv, offs)
        w = torch.randn(len(offs), cin, cout, generator=g) / (len(offs) * cin / 2) ** 0.5
    xc, wc = x.cuda(), w.cuda()
    nb = None if nbr is None else nbr.cuda()
    split, xs = hip.split_weight_rows(wc), hip.split_rows(xc)
    T, lo = 40, -7
    tab = torch.randn(3, T, cout, generator=g)
    acoords = torch.cat([torch.zeros(n, 1, dtype=torch.int32), torch.randint(lo, lo + T, (n, 3), generator=g, dtype=torch.int32)],
                        dim=1).contiguous()
    res = torch.randn(n, cout, generator=g)
    bias = torch.randn(cout, generator=g)
    es, eb = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    e2s, e2b = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    osc, osh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1

    def dev(v):
        if torch.is_tensor(v):
            return v.cuda()
        if isinstance(v, tuple):
            return tuple(dev(t) for t in v)
        return v

    seen = set()
    for acts, osp, tail, emit in itertools.product((False, True), (False, True), TAILS, (False, True)):
        if osp and not emit:
            continue
        kw = dict(bias=bias)
        if acts:
            kw.update(epi_scale=es, epi_shift=eb, epi_act=2, slope=0.1)
        if tail == "bn2":
            kw.update(epi2_scale=e2s, epi2_shift=e2b, res_act=1 if acts else 0)
        if "residual" in tail:
            kw.update(residual=res, res_act=1 if acts else 0)
        if "axis" in tail:
            kw["axis"] = (tab, acoords, lo)
        exp = oracle.conv_fwd(x, w, nbr, n, **kw)
        kwd = {k: dev(v) for k, v in kw.items()}
        if not emit:
            got = hip.conv_fwd(xc, wc, nb, n, split=split, in_split=xs, **kwd)
        else:
            e = (osc.cuda(), osh.cuda(), 1 if acts else 0) if osp else (None, None, 0)
            got, op = hip.conv_fwd(xc, wc, nb, n, split=split, in_split=xs, emit_split=e, **kwd)
            want = hip.split_rows(got, pro_scale=e[0], pro_shift=e[1], pro_act=e[2], slope=0.1 if acts else 0.01)
            assert torch.equal(op.view(torch.int16), want.view(torch.int16)), (acts, osp, tail)
            only = hip.conv_fwd(xc, wc, nb, n, split=split, in_split=xs, emit_split=e, want_out=False, **kwd)
            only = only[1] if isinstance(only, tuple) else only
            assert torch.equal(only.view(torch.int16), op.view(torch.int16)), (acts, osp, tail, "operand only")
        seen.add(hip.conv_last_config()["kernel"])
        err = float((got.cpu() - exp).abs().max()) / float(exp.abs().mean())
        assert err < 1e-4, (acts, osp, tail, emit, err)
    hip.check_status(torch.device("cuda", 0))
    assert seen, "no launch recorded"
