"""k_conv_lin (conv_lin.hip): k = 1 products of 64 / 128 input channels as a row stream - against the oracle, and bit for bit
against k_conv_dma (same products in the same order, same epilogue).  Reference layers: the mask heads' 1x1x1 convolutions
(decoder_v3.py:267-282), the K / V projections and linear layers of the transformer (transformer_predictor_v2.py:167-177,
blocks.py:83-90), the point MLP (cylinder_feat.py)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _switch(hip, on):
    """ph_conv_desc.route of this thread's launches: off = k = 1 products stay on k_conv_dma"""
    from pasco_amd.me.backend import ROUTE_LIN_NEVER
    hip.set_route(0 if on else ROUTE_LIN_NEVER)


def _case(hip, n, cin, cout, gather, tail, emit, seed):
    g = torch.Generator().manual_seed(seed)
    n_in = n + 17 if gather else n
    x = torch.randn(n_in, cin, generator=g)
    w = torch.randn(1, cin, cout, generator=g) / cin ** 0.5
    nbr = None
    if gather:                                         # a k = 1 map with holes and repeats (pruned / expanded rows)
        nbr = torch.randint(0, n_in, (1, n), generator=g, dtype=torch.int32)
        nbr[0, torch.rand(n, generator=g) < 0.1] = -1
    kw = dict(bias=torch.randn(cout, generator=g))
    if "bn" in tail:
        kw.update(epi_scale=torch.rand(cout, generator=g) + 0.5, epi_shift=torch.randn(cout, generator=g) * 0.1, epi_act=2, slope=0.1)
    if "residual" in tail:
        kw.update(residual=torch.randn(n, cout, generator=g), res_act=1 if "bn" in tail else 0)
    if "axis" in tail:
        T, lo = 40, -7
        tab = torch.randn(3, T, cout, generator=g)
        ac = torch.cat([torch.zeros(n, 1, dtype=torch.int32), torch.randint(lo, lo + T, (n, 3), generator=g, dtype=torch.int32)], dim=1)
        kw["axis"] = (tab, ac.contiguous(), lo)
    return x, w, nbr, kw


def _dev(v):
    if torch.is_tensor(v):
        return v.cuda()
    if isinstance(v, tuple):
        return tuple(_dev(t) for t in v)
    return v


CASES = [
    # n, cin, cout, gather, tail, emit
    (1, 64, 128, False, "none", False),
    (31, 64, 100, False, "axis", False),               # the mask heads' width: columns past cout in the last block of 32
    (33, 128, 128, True, "residual", True),
    (3001, 64, 128, False, "bn", True),
    (3001, 64, 256, True, "bn+residual+axis", True),   # two column tiles
    (3001, 128, 384, False, "axis", True),             # three column tiles (K / V projection of a coarse level)
    (3001, 128, 100, True, "bn+axis", False),
    (70001, 64, 128, False, "none", True),             # more blocks than resident waves: every wave walks several
    (70001, 128, 256, True, "residual", False),
    (300001, 64, 128, False, "none", True),            # four to five blocks per wave
    (300001, 128, 256, False, "bn", False),
    (300001, 128, 100, False, "bn", False),            # fewer stores per block in the last column tile
]


@pytest.mark.parametrize("n,cin,cout,gather,tail,emit", CASES)
def test_lin_stream_matches_oracle_and_dma_kernel(hip, oracle, n, cin, cout, gather, tail, emit):
    x, w, nbr, kw = _case(hip, n, cin, cout, gather, tail, emit, seed=n + cin + cout)
    exp = oracle.conv_fwd(x, w, nbr, n, **kw)
    xc, wc = x.cuda(), w.cuda()
    nb = None if nbr is None else nbr.cuda()
    split, xs = hip.split_weight_rows(wc), hip.split_rows(xc)
    kwd = {k: _dev(v) for k, v in kw.items()}
    e = (None, None, 0)

    def run():
        if emit:
            out, op = hip.conv_fwd(xc, wc, nb, n, split=split, in_split=xs, emit_split=e, **kwd)
            return out, op, hip.conv_last_config()["kernel"]
        return hip.conv_fwd(xc, wc, nb, n, split=split, in_split=xs, **kwd), None, hip.conv_last_config()["kernel"]

    try:
        _switch(hip, 1)
        got, op, kid = run()
        _switch(hip, 0)
        ref, op_ref, kid_ref = run()
    finally:
        _switch(hip, 1)
    assert kid == 7 and kid_ref != 7, (kid, kid_ref)
    hip.check_status(torch.device("cuda", 0))
    err = float((got.cpu() - exp).abs().max()) / float(exp.abs().mean())
    assert err < 1e-4, err
    assert torch.equal(got, ref), "row stream vs k_conv_dma: fp32 output"
    if emit:
        assert torch.equal(op.view(torch.int16), op_ref.view(torch.int16)), "row stream vs k_conv_dma: emitted operand"
        want = hip.split_rows(got)
        assert torch.equal(op.view(torch.int16), want.view(torch.int16)), "emitted operand vs ph_split_rows of the fp32 result"


def test_lin_stream_declines_other_shapes(hip):
    g = torch.Generator().manual_seed(5)
    n = 2000
    for cin, cout in ((256, 256), (32, 128), (64, 64)):
        x = torch.randn(n, cin, generator=g).cuda()
        w = (torch.randn(1, cin, cout, generator=g) / cin ** 0.5).cuda()
        hip.conv_fwd(x, w, None, n, split=hip.split_weight_rows(w), in_split=hip.split_rows(x))
        assert hip.conv_last_config()["kernel"] != 7, (cin, cout)
