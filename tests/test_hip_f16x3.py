"""Opt-in split-precision convolution (mma_mode 1: hi*hi + hi*lo + lo*hi on f16 MFMA, fp32 accumulate):
parity with the oracle and an accuracy comparison against an fp64 reference next to the exact-fp32 path."""
import numpy as np
import pytest
import torch

from pasco_amd.me.core import kernel_offsets
from tests.test_hip_ops import scene_coords, unique_map

pytestmark = pytest.mark.gpu


def fp64_reference(x, w, nbr, rows):
    ref = torch.zeros(rows.shape[0], w.shape[2], dtype=torch.float64, device=x.device)
    for k in range(w.shape[0]):
        idx = nbr[k][rows].long()
        ok = idx >= 0
        ref[ok] += x[idx[ok]].double() @ w[k].double()
    return ref


@pytest.mark.parametrize("cin,cout,n", [(64, 64, 30000), (128, 128, 9000), (256, 256, 3000), (64, 20, 5000), (192, 64, 4000)])
def test_split_conv_matches_oracle_and_fp64(hip, oracle, cin, cout, n):
    coords = scene_coords(31, n)
    tk_o, tv_o, c_o, _, _ = unique_map(oracle, coords)
    tk_h, tv_h, c_h, _, _ = unique_map(hip, coords.cuda())
    offs = kernel_offsets(3, 1)
    nbr_o = oracle.nbr_build(c_o, tk_o, tv_o, offs)
    nbr_h = hip.nbr_build(c_h, tk_h, tv_h, offs)
    g = torch.Generator().manual_seed(32)
    m = c_o.shape[0]
    x = torch.randn(m, cin, generator=g) * torch.exp(torch.randn(m, 1, generator=g))      # rows of mixed magnitude
    w = torch.randn(27, cin, cout, generator=g) * torch.exp(2 * torch.randn(27, 1, cout, generator=g)) / 200
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.2
    res = torch.randn(m, cout, generator=g)
    kw = dict(pro_scale=ps, pro_shift=pb, pro_act=1, residual=res, res_act=1)
    exp = oracle.conv_fwd(x, w, nbr_o, m, **kw)
    kw_h = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
    split = hip.split_weight_f16(w.cuda())
    got = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m, split=split, **kw_h).cpu()
    assert torch.allclose(got, exp, rtol=1e-4, atol=1e-4), float((got - exp).abs().max())
    # mode 2 (operands pre-split once): the same hi / lo values reach the same MFMAs -> identical to mode 1
    split2 = hip.split_weight_rows(w.cuda())
    got2 = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m, split=split2, **kw_h).cpu()
    assert torch.allclose(got2, exp, rtol=1e-4, atol=1e-4), float((got2 - exp).abs().max())
    # accuracy against fp64, plain conv (no fusion), next to the exact-fp32 MFMA path
    rows = torch.randint(0, m, (1500,), generator=g).cuda()
    ref = fp64_reference(x.cuda(), w.cuda(), nbr_h, rows)
    f32 = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m)[rows].double()
    spl_all = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m, split=split)
    # without a prologue the two modes feed identical hi / lo values to identical MFMA sequences
    assert torch.equal(hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m, split=split2), spl_all)
    spl = spl_all[rows].double()
    scale = ref.abs().mean()
    e32 = float((f32 - ref).abs().max() / scale)
    esp = float((spl - ref).abs().max() / scale)
    print(f"cin={cin} cout={cout}: max err / mean|y|  fp32 MFMA {e32:.2e}   f16x3 {esp:.2e}")
    # the split path is at (or below) the exact fp32 MFMA path's error against fp64
    assert esp <= 1.5 * e32 + 2e-6, (e32, esp)


def test_split_conv_identity_map_and_strided(hip, oracle):
    g = torch.Generator().manual_seed(33)
    n = 7000
    x = torch.randn(n, 64, generator=g)
    w = torch.randn(64, 128, generator=g) / 8
    b = torch.randn(128, generator=g)
    exp = oracle.conv_fwd(x, w, None, n, bias=b)
    got = hip.conv_fwd(x.cuda(), w.cuda(), None, n, bias=b.cuda(), split=hip.split_weight_f16(w.cuda())).cpu()
    assert torch.allclose(got, exp, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,c", [(1, 8), (333, 24), (5000, 64), (4097, 200), (2000, 384)])
def test_split_rows_bit_exact(hip, oracle, n, c):
    """ph_split_rows against the oracle's integer restatement of the f32 -> f16 hi / lo split."""
    g = torch.Generator().manual_seed(40 + c)
    x = (torch.randn(n, c, generator=g) * torch.exp(3 * torch.randn(n, 1, generator=g))).clamp(-6.0e4, 6.0e4)
    x[0, 0] = 0.0
    if n > 3:
        x[1, 1], x[2, 2], x[3, 3] = 65504.0, -6.1e-5, 5.9e-8          # largest finite, near-subnormal, tiny
    ps, pb = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    for kw in (dict(exp2=0), dict(pro_scale=ps, pro_shift=pb, pro_act=2, slope=0.1, exp2=0),
               dict(pro_scale=ps, pro_shift=pb, pro_act=1)):          # last: an activation operand (x * 2^5)
        if len(kw) > 1:
            x = x.clamp(-3.0e4, 3.0e4) if "exp2" in kw else x.clamp(-1.0e3, 1.0e3)
        exp = oracle.split_rows(x, **kw)
        kw_h = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
        got = hip.split_rows(x.cuda(), **kw_h).cpu()
        assert torch.equal(got.view(torch.int16), exp.view(torch.int16))
    hip.check_status(torch.device("cuda", 0))


def test_split_conv_mode2_fullgrid_splitk(hip, oracle):
    """Few-row / many-offset layer (the dense bottleneck shape class): split-K path with pre-split operands."""
    g = torch.Generator().manual_seed(41)
    xs = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(4), indexing="ij"), -1).reshape(-1, 3)
    coords = torch.from_numpy(np.concatenate([np.zeros((xs.shape[0], 1), np.int64), xs], 1)).int()
    tk_o, tv_o, c_o, _, _ = unique_map(oracle, coords)
    tk_h, tv_h, c_h, _, _ = unique_map(hip, coords.cuda())
    offs = kernel_offsets(3, 1)
    nbr_o = oracle.nbr_build(c_o, tk_o, tv_o, offs)
    nbr_h = hip.nbr_build(c_h, tk_h, tv_h, offs)
    m = c_o.shape[0]
    x = torch.randn(m, 256, generator=g)
    w = torch.randn(27, 256, 256, generator=g) / 60
    b = torch.randn(256, generator=g)
    exp = oracle.conv_fwd(x, w, nbr_o, m, bias=b, epi_act=1)
    got = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m, bias=b.cuda(), epi_act=1, split=hip.split_weight_rows(w.cuda())).cpu()
    assert torch.allclose(got, exp, rtol=1e-4, atol=2e-4), float((got - exp).abs().max())


@pytest.mark.parametrize("cout,n,ksplit_shape", [(64, 20000, False), (128, 9000, False), (256, 1500, True)])
def test_conv_emits_next_operand(hip, oracle, cout, n, ksplit_shape):
    """mode 2 with out_split: the second output is bit-identical to ph_split_rows of the fp32 output with the
    next layer's prologue, for the direct epilogue and for the split-K epilogue; out may be skipped."""
    coords = scene_coords(51, n)
    tk_h, tv_h, c_h, _, _ = unique_map(hip, coords.cuda())
    nbr_h = hip.nbr_build(c_h, tk_h, tv_h, kernel_offsets(3, 1))
    m = c_h.shape[0]
    g = torch.Generator().manual_seed(52)
    cin = 64
    x = torch.randn(m, cin, generator=g).cuda()
    w = (torch.randn(27, cin, cout, generator=g) / 30).cuda()
    res = torch.randn(m, cout, generator=g).cuda()
    sc, sh = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.3).cuda()
    split = hip.split_weight_rows(w)
    for emit in ((None, None, 0), (sc, sh, 1), (sc, None, 2)):
        out, osp = hip.conv_fwd(x, w, nbr_h, m, split=split, residual=res, res_act=1, slope=0.2, emit_split=emit)
        plain = hip.conv_fwd(x, w, nbr_h, m, split=split, residual=res, res_act=1, slope=0.2)
        assert torch.equal(out, plain)
        exp = hip.split_rows(out, pro_scale=emit[0], pro_shift=emit[1], pro_act=emit[2], slope=0.2)
        assert torch.equal(osp.view(torch.int16), exp.view(torch.int16))
        none_out, osp2 = hip.conv_fwd(x, w, nbr_h, m, split=split, residual=res, res_act=1, slope=0.2, emit_split=emit,
                                      want_out=False)
        assert none_out is None and torch.equal(osp2.view(torch.int16), exp.view(torch.int16))
    hip.check_status(x.device)


@pytest.mark.parametrize("cin,cout,n_parents", [(128, 64, 9000), (256, 128, 3000), (256, 256, 2500)])
def test_generative_transposed_conv_on_row_lists(hip, oracle, cin, cout, n_parents):
    """One-pair-per-row maps (every child has its one parent, mink.py:524-527): the launch over row lists grouped by kernel
    offset (ph_conv_desc.rl_*, kernel id 3) gives bit for bit what the walk over all 8 offsets gives, the lists equal the
    oracle's, and the result matches the oracle's convolution."""
    from pasco_amd.me.core import kernel_offsets
    g = torch.Generator().manual_seed(cin + cout)
    par = torch.cat([torch.zeros(n_parents, 1, dtype=torch.int64), torch.randint(0, 40, (n_parents, 3), generator=g) * 2], 1).int()
    par = torch.unique(par, dim=0)
    par = par[torch.randperm(par.shape[0], generator=g)].contiguous()
    kids = hip.coords_expand(par.cuda(), 1)                     # 8 children per parent, stride 2 -> 1
    keep = (torch.rand(kids.shape[0], generator=g) < 0.8).cuda()         # pruned like the decoder's bounds test
    kids = kids[keep].contiguous()
    tk, tv, *_ = hip.map_insert(par.cuda(), dedup=False)
    offs = kernel_offsets(2, 1, 1, True)
    nbr = hip.nbr_build(kids, tk, tv, offs)
    assert bool(((nbr >= 0).sum(0) == 1).all())                 # exactly one pair per child
    n_out = kids.shape[0]
    rl = hip.rowlist_build(nbr)
    pin, pout, counts = oracle.kmap_compact(nbr.cpu())
    rl_o = oracle.rowlist_build(nbr.cpu())
    for key in ("in", "out", "tile_k"):
        assert torch.equal(rl[key].cpu(), rl_o[key])
    x = torch.randn(par.shape[0], cin, generator=g).cuda()
    w = (torch.randn(8, cin, cout, generator=g) / cin ** 0.5).cuda()
    b = torch.randn(cout, generator=g).cuda()
    es, eb = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()
    split = hip.split_weight_rows(w)
    kw = dict(bias=b, epi_scale=es, epi_shift=eb, epi_act=2, slope=0.05, split=split, emit_split=(None, None, 0))
    plain, plain_op = hip.conv_fwd(x, w, nbr, n_out, **kw)
    assert hip.conv_last_config()["kernel"] != 3
    got, got_op = hip.conv_fwd(x, w, nbr, n_out, rowlist=rl, **kw)
    assert hip.conv_last_config()["kernel"] == 3
    assert torch.equal(got, plain) and torch.equal(got_op.view(torch.int16), plain_op.view(torch.int16))
    only, only_op = hip.conv_fwd(x, w, nbr, n_out, rowlist=rl, want_out=False, **kw)
    assert only is None and torch.equal(only_op.view(torch.int16), plain_op.view(torch.int16))
    exp = oracle.conv_fwd(x.cpu(), w.cpu(), nbr.cpu(), n_out, bias=b.cpu(), epi_scale=es.cpu(), epi_shift=eb.cpu(), epi_act=2,
                          slope=0.05)
    assert torch.allclose(got.cpu(), exp, rtol=1e-4, atol=1e-4)
