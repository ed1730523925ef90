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
    # accuracy against fp64, plain conv (no fusion), next to the exact-fp32 MFMA path
    rows = torch.randint(0, m, (1500,), generator=g).cuda()
    ref = fp64_reference(x.cuda(), w.cuda(), nbr_h, rows)
    f32 = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m)[rows].double()
    spl = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m, split=split)[rows].double()
    scale = ref.abs().mean()
    e32 = float((f32 - ref).abs().max() / scale)
    esp = float((spl - ref).abs().max() / scale)
    print(f"cin={cin} cout={cout}: max err / mean|y|  fp32 MFMA {e32:.2e}   f16x3 {esp:.2e}")
    assert esp < 8 * e32 + 2e-6, (e32, esp)


def test_split_conv_identity_map_and_strided(hip, oracle):
    g = torch.Generator().manual_seed(33)
    n = 7000
    x = torch.randn(n, 64, generator=g)
    w = torch.randn(64, 128, generator=g) / 8
    b = torch.randn(128, generator=g)
    exp = oracle.conv_fwd(x, w, None, n, bias=b)
    got = hip.conv_fwd(x.cuda(), w.cuda(), None, n, bias=b.cuda(), split=hip.split_weight_f16(w.cuda())).cpu()
    assert torch.allclose(got, exp, rtol=1e-4, atol=1e-4)
