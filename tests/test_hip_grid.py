"""k_conv_grid (conv_grid.hip: the dense-grid promise of ph_conv_desc.grid_dims / grid_kernel - activations from LDS windows,
neighbours by arithmetic) against the oracle, which reads the kernel map, against fp64, and against the gather kernels
(PH_ROUTE_GRID_NEVER) on the same launch: the bottleneck's three kernel shapes at its real size, batches, ragged row counts,
narrow inputs, the unsplit form with operand emission.  Reference: the dense ASPP block, layers.py:656-726 (torch Conv3d with
'same' padding on the densified stride-8 level)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def grid_map(oracle, hip, dims, ks):
    """sites (b, z, x, y) - y fastest - of the full grid, the box kernel map in the promise's offset order"""
    b, x, y, z = dims
    ax = [torch.arange(n, dtype=torch.int32) for n in (b, z, x, y)]
    bzxy = torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1).reshape(-1, 4)
    coords = bzxy[:, [0, 2, 3, 1]].contiguous()            # columns (b, x, y, z)
    offs = hip.grid_offsets(ks)
    tk, tv, _, _, _ = oracle.map_insert(coords, dedup=False)
    parts = [oracle.nbr_build(coords, tk, tv, offs[i:i + 64]) for i in range(0, len(offs), 64)]
    return coords, torch.cat(parts, dim=0).contiguous()


def test_grid_offsets_order(hip):
    offs = hip.grid_offsets((7, 5, 3))
    kx, ky, kz = 7, 5, 3
    for k, (dx, dy, dz) in enumerate(offs):
        iy, ix, iz = k % ky, (k // ky) % kx, k // (ky * kx)
        assert (dx, dy, dz) == (ix - kx // 2, iy - ky // 2, iz - kz // 2)


@pytest.mark.parametrize("dims,ks,cin,cout", [((1, 38, 44, 4), (7, 7, 5), 256, 256), ((1, 38, 44, 4), (5, 5, 3), 256, 256),
                                              ((1, 38, 44, 4), (3, 3, 1), 256, 256), ((2, 9, 11, 3), (7, 7, 5), 64, 128),
                                              ((1, 17, 13, 5), (3, 5, 3), 96, 128), ((3, 5, 7, 2), (5, 3, 1), 32, 256),
                                              ((1, 40, 40, 4), (5, 7, 5), 128, 128), ((1, 3, 300, 1), (3, 7, 1), 64, 128),
                                              ((1, 2, 2, 2), (3, 3, 3), 32, 128), ((4, 1, 5, 1), (1, 5, 1), 32, 128),
                                              ((1, 64, 2, 8), (7, 3, 5), 64, 256)])     # tiny grids, y shorter than the kernel, 8 planes
def test_grid_conv_matches_oracle_fp64_and_the_gather_kernel(hip, oracle, dims, ks, cin, cout):
    from pasco_amd.me.backend import ROUTE_GRID_NEVER

    coords, nbr = grid_map(oracle, hip, dims, ks)
    n = coords.shape[0]
    kvol = ks[0] * ks[1] * ks[2]
    g = torch.Generator().manual_seed(n + kvol)
    x = torch.randn(n, cin, generator=g)
    x[torch.rand(n, generator=g) < 0.3] = 0                 # empty sites of the densified level
    w = torch.randn(kvol, cin, cout, generator=g) / (kvol * cin) ** 0.5
    es, eb = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, generator=g)
    exp = oracle.conv_fwd(x, w, nbr, n, epi_scale=es, epi_shift=eb, epi_act=1, residual=res)
    # fp64: the same sum by gathers
    x64 = torch.cat([x.double(), torch.zeros(1, cin, dtype=torch.float64)])
    acc = torch.zeros(n, cout, dtype=torch.float64)
    for k in range(kvol):
        idx = nbr[k].long()
        acc += x64[torch.where(idx >= 0, idx, torch.full_like(idx, n))] @ w[k].double()
    exp64 = torch.relu(acc * es.double() + eb.double()) + res.double()

    xc, wc, nb = x.cuda(), w.cuda(), nbr.cuda()
    split, xs = hip.split_weight_rows(wc), hip.split_rows(xc)
    kw = dict(epi_scale=es.cuda(), epi_shift=eb.cuda(), epi_act=1, residual=res.cuda())
    got = hip.conv_fwd(xc, wc, nb, n, split=split, in_split=xs, grid=(dims, ks), **kw)
    cfg = hip.conv_last_config()
    assert cfg["kernel"] == 8 and cfg["bm"] == 256 and cfg["bn"] == 128, cfg
    scale = float(exp64.abs().mean())
    assert float((got.cpu() - exp).abs().max()) / scale < 1e-4
    assert float((got.cpu().double() - exp64).abs().max()) / scale < 2e-5
    with hip.routing(ROUTE_GRID_NEVER):
        ref = hip.conv_fwd(xc, wc, nb, n, split=split, in_split=xs, grid=(dims, ks), **kw)
        assert hip.conv_last_config()["kernel"] != 8
    assert torch.allclose(got, ref, rtol=1e-4, atol=2e-5)
    # the window kernel is no less exact than the gather kernel it replaces
    e_grid = float((got.cpu().double() - exp64).abs().max())
    e_ref = float((ref.cpu().double() - exp64).abs().max())
    assert e_grid < 2.0 * e_ref + 1e-6, (e_grid, e_ref)


def test_grid_conv_unsplit_emits_the_next_operand(hip, oracle):
    """Few units (one group row of a (1, 3, 1) box on 32 channels): no split over the units, the epilogue runs in the kernel and
    emits the split operand of the next convolution."""
    dims, ks, cin, cout = (1, 20, 30, 2), (1, 3, 1), 32, 128
    coords, nbr = grid_map(oracle, hip, dims, ks)
    n = coords.shape[0]
    g = torch.Generator().manual_seed(5)
    x, w = torch.randn(n, cin, generator=g), torch.randn(3, cin, cout, generator=g) / 10
    exp = oracle.conv_fwd(x, w, nbr, n, epi_act=1)
    xc, wc, nb = x.cuda(), w.cuda(), nbr.cuda()
    split, xs = hip.split_weight_rows(wc), hip.split_rows(xc)
    osc, osh = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.1).cuda()
    out, osp = hip.conv_fwd(xc, wc, nb, n, split=split, in_split=xs, grid=(dims, ks), epi_act=1, emit_split=(osc, osh, 1))
    cfg = hip.conv_last_config()
    assert cfg["kernel"] == 8 and cfg["ksplit"] == 1 and cfg["emit"] == 1, cfg
    assert float((out.cpu() - exp).abs().max()) / float(exp.abs().mean()) < 1e-4
    assert torch.equal(osp.view(torch.int16), hip.split_rows(out, pro_scale=osc, pro_shift=osh, pro_act=1).view(torch.int16))


def test_grid_promise_is_checked_by_the_binding(hip):
    x = torch.randn(24, 32, device="cuda")
    w = torch.randn(9, 32, 128, device="cuda")
    nbr = torch.full((9, 24), -1, dtype=torch.int32, device="cuda")
    split = hip.split_weight_rows(w)
    with pytest.raises(ValueError):
        hip.conv_fwd(x, w, nbr, 24, split=split, grid=((1, 2, 3, 5), (3, 3, 1)))      # 30 sites, 24 rows
    with pytest.raises(ValueError):
        hip.conv_fwd(x, w, nbr, 24, split=split, grid=((1, 2, 3, 4), (3, 2, 1)))      # even size / wrong volume
