"""Randomised GPU-vs-oracle sweep over operator shapes (seeded): convolutions with random channel
counts / kernel volumes / fusion flags on both product paths, coordinate maps with random duplicate
rates, pruning / union / pooling chains."""
import numpy as np
import pytest
import torch

from pasco_amd.me.core import kernel_offsets
from tests.test_hip_ops import scene_coords, unique_map

pytestmark = pytest.mark.gpu


def test_conv_fuzz(hip, oracle):
    rng = np.random.default_rng(123)
    g = torch.Generator().manual_seed(123)
    for it in range(28):
        ks = int(rng.choice([1, 2, 3]))
        cin = int(rng.choice([8, 16, 24, 40, 64, 67, 96, 128, 131, 192, 256]))
        cout = int(rng.choice([4, 20, 32, 64, 100, 128, 256]))
        n = int(rng.choice([1, 31, 33, 64, 65, 500, 4097, 12000]))
        coords = scene_coords(1000 + it, n, extent=(24, 20, 10))
        tk_o, tv_o, c_o, _, _ = unique_map(oracle, coords)
        tk_h, tv_h, c_h, _, _ = unique_map(hip, coords.cuda())
        m = c_o.shape[0]
        if ks == 1:
            nbr_o = nbr_h = None
            w = torch.randn(cin, cout, generator=g) / np.sqrt(cin)
        else:
            offs = kernel_offsets(ks, 1)
            nbr_o = oracle.nbr_build(c_o, tk_o, tv_o, offs)
            nbr_h = hip.nbr_build(c_h, tk_h, tv_h, offs)
            assert torch.equal(nbr_h.cpu(), nbr_o)
            w = torch.randn(len(offs), cin, cout, generator=g) / np.sqrt(cin * len(offs) / 2)
        x = torch.randn(m, cin, generator=g)
        kw = {}
        if rng.random() < 0.6:
            kw.update(pro_scale=torch.rand(cin, generator=g) + 0.5, pro_shift=torch.randn(cin, generator=g) * 0.1,
                      pro_act=int(rng.integers(0, 3)))
        if rng.random() < 0.6:
            kw.update(bias=torch.randn(cout, generator=g))
        if rng.random() < 0.6:
            kw.update(epi_scale=torch.rand(cout, generator=g) + 0.5, epi_shift=torch.randn(cout, generator=g) * 0.1,
                      epi_act=int(rng.integers(0, 3)))
        if rng.random() < 0.4:
            kw.update(epi2_scale=torch.rand(cout, generator=g) + 0.5, epi2_shift=torch.randn(cout, generator=g) * 0.1,
                      res_act=int(rng.integers(0, 3)))
        if rng.random() < 0.5:
            kw.update(residual=torch.randn(m, cout, generator=g), res_act=int(rng.integers(0, 3)))
        exp = oracle.conv_fwd(x, w, nbr_o, m, **kw)
        kw_h = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
        got = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m, **kw_h).cpu()
        assert torch.allclose(got, exp, rtol=1e-3, atol=2e-4), (it, ks, cin, cout, n, float((got - exp).abs().max()))
        if hip.split_supported(cin, cout):
            got2 = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m, split=hip.split_weight_f16(w.cuda()), **kw_h).cpu()
            assert torch.allclose(got2, exp, rtol=1e-3, atol=2e-4), ("split", it, ks, cin, cout, n,
                                                                      float((got2 - exp).abs().max()))
            got3 = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m, split=hip.split_weight_rows(w.cuda()), **kw_h).cpu()
            assert torch.allclose(got3, exp, rtol=1e-3, atol=2e-4), ("pre-split", it, ks, cin, cout, n,
                                                                      float((got3 - exp).abs().max()))
    hip.check_status(torch.device("cuda", 0))


def test_split_conv_range_flag(hip):
    """An activation beyond the f16 range must be reported, not silently wrapped to inf."""
    n, c = 256, 64
    x = torch.randn(n, c).cuda()
    w = (torch.randn(c, c) / 8).cuda()
    split = hip.split_weight_f16(w)
    hip.conv_fwd(x, w, None, n, split=split)
    hip.check_status(x.device)                       # in range: silent
    x[7, 3] = 1.0e5
    hip.conv_fwd(x, w, None, n, split=split)
    with pytest.raises(RuntimeError, match="f16 range"):
        hip.check_status(x.device)
    hip.check_status(x.device)                       # flag cleared
    hip.conv_fwd(x, w, None, n, split=hip.split_weight_rows(w))      # mode 2: raised by the operand split
    with pytest.raises(RuntimeError, match="f16 range"):
        hip.check_status(x.device)


def test_coordinate_chain_fuzz(hip, oracle):
    rng = np.random.default_rng(7)
    for it in range(10):
        n = int(rng.choice([1, 70, 1000, 20000]))
        dup = float(rng.choice([0.0, 0.2, 0.9]))
        coords = scene_coords(2000 + it, n, extent=(30, 30, 12), lo=(-13, -7, -3), batch=int(rng.integers(1, 3)), dup=dup)
        outs = []
        for be, dev in ((oracle, "cpu"), (hip, "cuda")):
            tk, tv, c, r2u, uq = unique_map(be, coords.to(dev))
            s = int(rng.choice([2, 4])) if be is oracle else s
            fl = be.coords_floor(c, s)
            tk2, tv2, c2, r2u2, _ = unique_map(be, fl)
            nbr = be.nbr_build(c2, tk, tv, kernel_offsets(s, 1))
            x = torch.randn(c.shape[0], 5, generator=torch.Generator().manual_seed(it)).to(dev)
            pooled = be.maxpool_fwd(x, nbr)
            mask = (torch.arange(c.shape[0]) % 3 != 0).to(dev)
            keep = be.mask_compact(mask)
            cat = torch.cat([be.gather_rows(c, keep), c]).contiguous()
            _, _, r2u3, uq3, nu3 = be.map_insert(cat)
            outs.append([t.cpu() for t in (c, r2u, c2, r2u2, nbr, pooled, keep, r2u3, uq3)])
        for a, b in zip(*outs):
            assert torch.equal(a, b), it
