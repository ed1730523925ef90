"""LDS-window convolution (pasco_amd/csrc/conv_win.hip): the window tables against the oracle's restatement (bit
exact), and the convolution served from windows against the gather kernels, the oracle and fp64 - single-pass tiles,
forced multi-pass tiles on a map without locality, ragged last tiles, emitted operands."""
import ctypes as C

import numpy as np
import pytest
import torch

from pasco_amd.graph.synth import make_occupancy
from pasco_amd.me.core import kernel_offsets
from tests.test_hip_ops import scene_coords, unique_map

pytestmark = pytest.mark.gpu


def _force(hip, mode):
    """ph_conv_desc.route of this thread's launches: 1 = the window kernel whatever the map's locality, -1 = never, 0 = the library decides"""
    from pasco_amd.me.backend import ROUTE_WIN_ALWAYS, ROUTE_WIN_NEVER
    hip.set_route({1: ROUTE_WIN_ALWAYS, -1: ROUTE_WIN_NEVER, 0: 0}[mode])


def s10_map(hip, shuffle=False, n=None, generative=False):
    """3x3x3 kernel map of the S10 scene.  generative=True: the rows of the decoder's stride-1 level - all 8 children
    of the stride-2 voxels, parent-major (octree order: runs of rows are compact bricks); else the lexicographic
    voxel list, optionally shuffled (no locality at all)."""
    g1 = np.argwhere(make_occupancy(0))
    c = torch.from_numpy(np.concatenate([np.zeros((g1.shape[0], 1), np.int64), g1], 1)).int().cuda().contiguous()
    if generative:
        c4 = hip.coords_floor(c, 4)
        _, _, _, uq4, _ = hip.map_insert(c4)
        c2 = hip.coords_expand(c4[uq4.long()].contiguous(), 2)          # children of the stride-4 voxels ...
        c = hip.coords_expand(c2, 1)                                     # ... and theirs: two generative levels
    if shuffle:
        c = c[torch.from_numpy(np.random.default_rng(0).permutation(c.shape[0])).cuda()].contiguous()
    if n is not None:
        c = c[:n].contiguous()
    tk, tv, _, _, _ = hip.map_insert(c, dedup=False)
    return hip.nbr_build(c, tk, tv, kernel_offsets(3, 1))


@pytest.mark.parametrize("n", [1, 127, 128, 3000, 40000])
def test_win_build_matches_oracle(hip, oracle, n):
    coords = scene_coords(61, n)
    tk_o, tv_o, c_o, _, _ = unique_map(oracle, coords)
    tk_h, tv_h, c_h, _, _ = unique_map(hip, coords.cuda())
    offs = kernel_offsets(3, 1)
    nbr_o, nbr_h = oracle.nbr_build(c_o, tk_o, tv_o, offs), hip.nbr_build(c_h, tk_h, tv_h, offs)
    wo, wh = oracle.win_build(nbr_o), hip.win_build(nbr_h)
    assert torch.equal(wh["cnt"].cpu(), wo["cnt"]) and torch.equal(wh["stats"].cpu(), wo["stats"])
    assert torch.equal(wh["slots"].cpu(), wo["slots"])
    cnt = wo["cnt"]
    for t in range(cnt.shape[0]):
        assert torch.equal(wh["rows"][t, : int(cnt[t])].cpu(), wo["rows"][t, : int(cnt[t])])


@pytest.mark.parametrize("n_rows,span", [(1000, 5000), (1000, 300000), (700, 262143), (700, 262145), (300, 4000000)])
def test_win_build_index_spans_on_both_paths(hip, oracle, n_rows, span):
    """ph_win_build on synthetic neighbour tables whose tiles span few or very many input rows: below 2^18 indices per
    tile the bitmap / popcount path builds the window, beyond it the hash set + sort - same tables either way (oracle)."""
    g = torch.Generator().manual_seed(span % 1000 + n_rows)
    nbr = torch.randint(0, span, (27, n_rows), generator=g).int()
    nbr[torch.rand(27, n_rows, generator=g) < 0.3] = -1
    nbr[:, 5] = -1                                   # a row without neighbours
    nbr[0, 0], nbr[1, 0] = 0, span - 1               # the full span inside the first tile
    wo, wh = oracle.win_build(nbr), hip.win_build(nbr.cuda())
    assert torch.equal(wh["cnt"].cpu(), wo["cnt"]) and torch.equal(wh["stats"].cpu(), wo["stats"])
    assert torch.equal(wh["slots"].cpu(), wo["slots"])
    for t in range(wo["cnt"].shape[0]):
        assert torch.equal(wh["rows"][t, : int(wo["cnt"][t])].cpu(), wo["rows"][t, : int(wo["cnt"][t])])


def _conv_case(hip, nbr, cin, cout, g, emit=False):
    n = nbr.shape[1]
    x = torch.randn(n, cin, device="cuda", generator=g) * torch.exp(torch.randn(n, 1, device="cuda", generator=g))
    w = torch.randn(27, cin, cout, device="cuda", generator=g) / (27 * cin) ** 0.5
    b = torch.randn(cout, device="cuda", generator=g)
    res = torch.randn(n, cout, device="cuda", generator=g)
    sc, sh = torch.rand(cout, device="cuda", generator=g) + 0.5, torch.randn(cout, device="cuda", generator=g) * 0.2
    kw = dict(bias=b, epi_scale=sc, epi_shift=sh, epi_act=1, residual=res, res_act=1)
    if emit:
        kw["emit_split"] = (sc, sh, 1)
    return x, w, kw


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128), (64, 128), (256, 256), (128, 64), (32, 64)])
def test_window_conv_equals_gather_conv(hip, cin, cout):
    """The same launch with and without window tables on the S10 map (windows chosen by the device-side predicate)."""
    nbr = s10_map(hip, n=90112 if cin == 256 else 300032, generative=True)
    n = nbr.shape[1]
    win = hip.win_build(nbr)
    passes = win["stats"].tolist()
    tiles = (n + 127) // 128
    assert passes[0] * 4 <= tiles * 5 and passes[1] * 4 <= tiles * 5, "a generative-order map should be window-friendly"
    g = torch.Generator(device="cuda").manual_seed(7)
    x, w, kw = _conv_case(hip, nbr, cin, cout, g, emit=True)
    split = hip.split_weight_rows(w)
    ref, ref_s = hip.conv_fwd(x, w, nbr, n, split=split, **kw)
    cfg_ref = hip.conv_last_config()
    _force(hip, 1)          # 128-wide tiles use windows only on request (measured slower): force the window side
    try:
        got, got_s = hip.conv_fwd(x, w, nbr, n, split=split, win=win, **kw)
    finally:
        _force(hip, 0)
    cfg = hip.conv_last_config()
    assert cfg_ref["kernel"] in (2, 4, 6) and cfg["kernel"] == 5      # 6: k_conv_wide serves the 256-channel map at this size
    # same products, different fp32 summation order (chunk-major vs offset-major): rounding-level agreement, measured
    # against the largest magnitude the sums reach
    big = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 4e-6 * big, (float((got - ref).abs().max()), big)
    # the emitted operand is the split of the fp32 result either way
    want = hip.split_rows(got, pro_scale=kw["emit_split"][0], pro_shift=kw["emit_split"][1], pro_act=1)
    assert torch.equal(got_s.view(torch.int16), want.view(torch.int16))
    # fp64 on sampled rows
    rows = torch.randint(0, n, (1500,), device="cuda", generator=g)
    acc = torch.zeros(1500, cout, dtype=torch.float64, device="cuda")
    for k in range(27):
        idx = nbr[k][rows].long()
        ok = idx >= 0
        acc[ok] += x[idx[ok]].double() @ w[k].double()
    r = torch.relu((acc + kw["bias"].double()) * kw["epi_scale"].double() + kw["epi_shift"].double())
    r = torch.relu(r + kw["residual"][rows].double())
    assert float((got[rows].double() - r).abs().max()) <= 1e-4 * float(acc.abs().mean())
    hip.check_status(x.device)


@pytest.mark.parametrize("cin,cout,n", [(64, 64, None), (128, 128, None), (64, 64, 129), (256, 256, None), (128, 256, 300),
                                       (64, 64, 300), (128, 64, None), (32, 64, 1000), (64, 64, 1)])
def test_window_conv_multi_pass_and_ragged(hip, oracle, cin, cout, n):
    """A shuffled map has no locality (windows of > 1000 rows): forced onto the window kernel it runs 3 - 5 passes
    per tile; the result must still be the convolution (oracle on fp32 operands).  n = 129: a ragged second tile."""
    nbr = s10_map(hip, shuffle=n is None, n=n)
    n = nbr.shape[1]
    win = hip.win_build(nbr)
    if n >= 9000:
        assert int(win["cnt"].max()) > 1024, "the shuffled map should need several passes"
    g = torch.Generator(device="cuda").manual_seed(8)
    x, w, kw = _conv_case(hip, nbr, cin, cout, g)
    split = hip.split_weight_rows(w)
    _force(hip, 1)
    try:
        got = hip.conv_fwd(x, w, nbr, n, split=split, win=win, **kw)
    finally:
        _force(hip, 0)
    ref = hip.conv_fwd(x, w, nbr, n, split=split, **kw)
    scale = float(ref.abs().mean())
    assert float((got - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
    rows = torch.randperm(n, device="cuda", generator=g)[:3000].sort()[0]      # the oracle on a row sample
    okw = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
    okw["residual"] = kw["residual"][rows].cpu()
    exp = oracle.conv_fwd(x.cpu(), w.cpu(), nbr[:, rows].contiguous().cpu(), rows.shape[0], **okw)
    assert torch.allclose(got[rows].cpu(), exp, rtol=1e-4, atol=1e-4 * scale)
    # the predicate would have sent this map to the gather kernel
    if n >= 9000:
        tiles = (n + 127) // 128
        assert win["stats"].tolist()[0] * 4 > tiles * 5
