"""Input stage: the point MLP + voxel max of `CylinderFeat` and the MIMO merge.

Pinned against vectors the REFERENCE's own `CylinderFeat.forward` + `Augmenter.merge` produced in the build container
(tests/golden/make_golden.py::golden_input_stage -> input_stage.npz), then the fused sort-free pass
(`CBackend.pooled_merge`: ph_points_* / ph_cells_max) against the torch formulation and, on the GPU, against the oracle
bit for bit."""
import numpy as np
import pytest
import torch

import pasco_amd.me as ME
from pasco_amd.graph.unet import CylinderFeat, PascoNet, merge_subnet_inputs
from pasco_amd.me.backend import StatusError
from tests.test_golden import load, sub_sd


def fixture_net():
    d = load("input_stage.npz")
    fea_dim, f, m = [int(v) for v in d["cfg"]]
    feat = CylinderFeat(fea_dim=fea_dim, out_pt_fea_dim=f).eval()
    missing, unexpected = feat.load_state_dict(sub_sd(d, "sd."), strict=True)
    pt_fea = [d[f"pt_fea_{i}"] for i in range(m)]
    xy_ind = [d[f"xy_ind_{i}"] for i in range(m)]
    return d, feat, pt_fea, xy_ind, m, f


def test_cylinder_feat_matches_the_reference():
    """PPmodel (BatchNorm1d / Linear / ReLU chain) + sorted unique + max, torch modules on CPU: the restatement itself."""
    d, feat, pt_fea, xy_ind, m, f = fixture_net()
    with torch.no_grad():
        unq, pooled = feat(pt_fea, xy_ind)
    assert torch.equal(unq, d["unq"].long())
    assert torch.allclose(pooled, d["pooled"], rtol=1e-5, atol=1e-6), float((pooled - d["pooled"]).abs().max())


def run_fused(device, d, feat, pt_fea, xy_ind, m, f):
    net = PascoNet(n_classes=20, n_infers=m, in_channels=pt_fea[0].shape[1], f=f, num_queries=4, heavy_decoder=False).eval()
    net.feat = feat
    net = net.to(device)
    with torch.no_grad():
        return net, net.prepare_input([t.to(device) for t in pt_fea], [t.to(device) for t in xy_ind])


def test_fused_input_stage_matches_the_reference_cpu(oracle_registered):
    """`PascoNet.prepare_input` on its fused route (oracle arithmetic): coordinates and row order of the merged tensor exactly
    the reference's, features to fp32 rounding of the MLP."""
    d, feat, pt_fea, xy_ind, m, f = fixture_net()
    calls = []
    inner = oracle_registered.pooled_merge
    oracle_registered.pooled_merge = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    try:
        net, x = run_fused("cpu", d, feat, pt_fea, xy_ind, m, f)
    finally:
        del oracle_registered.pooled_merge
    assert calls, "the fused route was not taken"
    assert torch.equal(x.C, d["merged_C"].int())
    assert torch.allclose(x.F, d["merged_F"], rtol=1e-5, atol=1e-6), float((x.F - d["merged_F"]).abs().max())
    oracle_registered.check_status(torch.device("cpu"))           # no all-zero row in the fixture
    with torch.no_grad():                                         # the general route gives the same tensor
        y = net.prepare_input(pt_fea, xy_ind, fused_stage=False)
    assert torch.equal(y.C, x.C) and torch.allclose(y.F, x.F, rtol=1e-6, atol=1e-7)


def random_points(seed, m, nvox, c, lo=(-20, -7, -3), extent=(60, 50, 12)):
    g = torch.Generator().manual_seed(seed)
    xyz, starts = [], [0]
    for i in range(m):
        vox = torch.stack([torch.randint(lo[a], lo[a] + extent[a], (nvox,), generator=g) for a in range(3)], dim=1)
        reps = 1 + torch.poisson(torch.ones(nvox), generator=g).long()
        ind = vox.repeat_interleave(reps, dim=0)
        xyz.append(ind[torch.randperm(ind.shape[0], generator=g)])
        starts.append(starts[-1] + xyz[-1].shape[0])
    xyz = torch.cat(xyz).contiguous()
    h = torch.randn(xyz.shape[0], c, generator=g)
    return h, xyz, starts


def torch_formulation(h, xyz, starts):
    """scatter-max per (subnet, voxel) + the sparse merge of pasco_amd.graph.unet (itself pinned by the U-Net fixtures)."""
    from pasco_amd.graph.unet import unique_rows_sorted
    b = torch.zeros(xyz.shape[0], dtype=torch.int64)
    for k in range(1, len(starts) - 1):
        b[starts[k]:] = k
    unq, inv = unique_rows_sorted(torch.cat([b[:, None], xyz], dim=1))
    pooled = torch.full((unq.shape[0], h.shape[1]), float("-inf"))
    pooled.scatter_reduce_(0, inv[:, None].expand_as(h), h, reduce="amax", include_self=True)
    return merge_subnet_inputs(ME.SparseTensor(pooled, unq.int()), len(starts) - 1)


@pytest.mark.parametrize("m,nvox,c", [(1, 500, 8), (3, 2000, 16), (8, 300, 4)])
def test_pooled_merge_equals_scatter_max_plus_merge_cpu(oracle_registered, m, nvox, c):
    h, xyz, starts = random_points(m * 7 + c, m, nvox, c)
    coords, feats = oracle_registered.pooled_merge(h, xyz, starts)
    ref = torch_formulation(h, xyz, starts)
    assert torch.equal(coords, ref.C) and torch.equal(feats, ref.F)
    # explicit bounds (a looser box than the points span) give the same rows
    c2, f2 = oracle_registered.pooled_merge(h, xyz, starts, bounds=((-64, -64, -16), (128, 128, 32)))
    assert torch.equal(c2, coords) and torch.equal(f2, feats)


def test_all_zero_row_is_flagged_for_the_general_route(oracle_registered):
    """A merged row whose channels are all exactly 0 (ME.to_sparse drops it): the fused pass cannot change the row count
    without a host read - it raises status bit 3 and the general route drops the row."""
    dev = torch.device("cpu")
    oracle_registered.status_word(dev).zero_()
    h, xyz, starts = random_points(3, 2, 200, 8)
    lone = torch.tensor([[500, 500, 500]])                              # a voxel only this point lives in, features 0
    xyz = torch.cat([xyz[:starts[1]], lone, xyz[starts[1]:]]).contiguous()
    h = torch.cat([h[:starts[1]], torch.zeros(1, 8), h[starts[1]:]]).contiguous()
    starts = [0, starts[1] + 1, starts[2] + 1]
    got = oracle_registered.pooled_merge(h, xyz, starts)
    ref = torch_formulation(h, xyz, starts)
    if got is not None:                                                # (the lone voxel blows the box up: may be declined)
        with pytest.raises(StatusError) as ei:
            oracle_registered.check_status(dev)
        assert ei.value.bits == 8 and got[0].shape[0] == ref.C.shape[0] + 1


@pytest.mark.gpu
def test_fused_input_stage_matches_the_reference_gpu(hip):
    d, feat, pt_fea, xy_ind, m, f = fixture_net()
    net, x = run_fused("cuda", d, feat, pt_fea, xy_ind, m, f)
    assert torch.equal(x.C.cpu(), d["merged_C"].int())
    assert torch.allclose(x.F.cpu(), d["merged_F"], rtol=1e-4, atol=1e-5), float((x.F.cpu() - d["merged_F"]).abs().max())
    hip.check_status(torch.device("cuda", 0))


@pytest.mark.gpu
@pytest.mark.parametrize("m,nvox,c,extent", [(1, 500, 8, (60, 50, 12)), (3, 63000, 64, (256, 256, 32)), (8, 20000, 64, (200, 200, 30))])
def test_pooled_merge_hip_equals_oracle(hip, oracle, m, nvox, c, extent):
    """Same points, same features: coordinates, row order and every max bit for bit (max is exact; the order of the atomics
    does not matter) - up to the benchmark's size (3 x 63 k voxels x ~2 points, 64 channels)."""
    h, xyz, starts = random_points(m + c, m, nvox, c, lo=(-8, -16, -2), extent=extent)
    exp_c, exp_f = oracle.pooled_merge(h, xyz, starts)
    got_c, got_f = hip.pooled_merge(h.cuda(), xyz.cuda(), starts)
    assert torch.equal(got_c.cpu(), exp_c) and torch.equal(got_f.cpu(), exp_f)
    hip.check_status(torch.device("cuda", 0))
