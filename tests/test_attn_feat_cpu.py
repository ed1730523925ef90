"""Attention straight on a level's feature operand (ph_attn_cross_feat / CrossAttentionLayer.attend_feat), CPU tier: the
algebra (composed query / output maps, position columns, angle model of the sine table) against the reference's own form of
the layer - key = value = bb_feat + pos through nn.MultiheadAttention (transformer/blocks.py:73-92) - on the oracle."""
import os

import pytest
import torch
import torch.nn as nn

from pasco_amd.graph import fused
from pasco_amd.graph.transformer import CrossAttentionLayer, PositionEmbeddingSineSparse, sine_position_encoding
from pasco_amd.me import backend


@pytest.fixture()
def split_checker(oracle):
    backend.register_checker_backend(oracle)
    oracle.status_word(torch.device("cpu")).zero_()
    oracle.checker_split = True
    try:
        yield oracle
    finally:
        oracle.checker_split = False
        backend.register_checker_backend(None)


def test_angle_model_reproduces_the_table(split_checker):
    """tab[t] = tab[far] + [t == 0] (tab[0] - tab[far]) + eps_t G to the rounding of the table's own entries."""
    pe = PositionEmbeddingSineSparse(128, normalize=True)
    dev = torch.device("cpu")
    tab = pe.table(dev).double()
    eps, G, i0, ifar = pe.angle_model(dev)
    eps = eps.double() * 2.0 ** -pe.EPS_EXP2
    model = tab[ifar][None] + eps[:, None] * G[None]
    model[i0] = tab[i0]
    err = (model - tab).abs().max().item()
    assert err < 4e-7, err                     # fp32 sin / cos of an fp32 angle: a few 1e-7 of rounding in the table itself
    nz = (eps != 0).nonzero().reshape(-1) + pe.TABLE_LO
    assert nz.abs().max().item() < 256         # only small |t| differ from the far angle
    # and the table is what the reference's formula gives
    ref = sine_position_encoding(torch.arange(-40, 300)[:, None].repeat(1, 3), 128)[:, :128]
    assert torch.allclose(tab[-40 - pe.TABLE_LO:300 - pe.TABLE_LO].float(), ref, atol=2e-6)


@pytest.mark.parametrize("C,N,masked", [(64, 500, True), (32, 77, False), (128, 300, True)])
def test_attend_feat_equals_the_reference_layer(split_checker, C, N, masked):
    torch.manual_seed(C + N)
    D, H, Q, B = 384, 8, 20, 2
    ca = CrossAttentionLayer(D, H).eval()
    lin = nn.Linear(C, D)
    for p in list(ca.parameters()) + list(lin.parameters()):      # biases and norms away from their zero / one defaults
        if p.dim() == 1:
            nn.init.normal_(p, 0.0, 0.3)
    pe = PositionEmbeddingSineSparse(D // 3, normalize=True)
    x = torch.randn(B, N, C)
    x[1, N - 9:] = 0                                               # zero-padded rows: keys like any other (padding_mask=None)
    coords = torch.randint(-3, 70, (B, N, 4), dtype=torch.int32)
    coords[..., 0] = torch.arange(B)[:, None]
    coords[0, :40, 1:] = torch.randint(0, 2, (40, 3), dtype=torch.int32)    # zeros on single axes
    coords[1, N - 9:, 1:] = 0
    q_embed, query_pos = torch.randn(B, Q, D), torch.randn(B, Q, D)
    allow = None
    bits = any_ = None
    if masked:
        allow = torch.rand(B, N, Q) > 0.6
        allow[:, :, 3] = False                                     # a query with nothing allowed attends everywhere
        bits, any_ = split_checker.attn_mask_pack(allow.reshape(B * N, Q).float().contiguous(), B, N)
    with torch.no_grad():
        # reference form, fp64
        pos = sine_position_encoding(coords.reshape(-1, 4)[:, 1:], D // 3).reshape(B, N, D).double()
        src = lin.double()(x.double())
        mha = ca.multihead_attn.double()
        qn = ca.norm.double()(q_embed.double())
        am = None
        if masked:
            am = ~allow.permute(0, 2, 1)
            am = am & ~am.all(dim=-1, keepdim=True)
            am = am.repeat_interleave(H, dim=0)
        exp = qn + mha(query=qn + query_pos.double(), key=src + pos, value=src + pos, attn_mask=am)[0]
        ca.float(); lin.float()
        # the layer on the feature operand
        from pasco_amd.me.backend import SPLIT_ACT_EXP2
        comp = ca.composed_feat(lin, pe, SPLIT_ACT_EXP2)
        x_split = split_checker.split_rows(x.reshape(B * N, C).contiguous())
        aug = split_checker.pos_aug(coords.reshape(B * N, 4).contiguous(), pe.angle_model(torch.device("cpu"))[0], pe.TABLE_LO)
        got = ca.attend_feat(q_embed, comp, x_split, aug, N, query_pos, (bits, any_))
    err = (got.double() - exp).abs().max().item()
    assert err < 5e-5 * exp.abs().max().item(), (err, exp.abs().max().item())
    split_checker.check_status(torch.device("cpu"))


def test_pos_aug_flags_a_coordinate_outside_the_table(split_checker):
    pe = PositionEmbeddingSineSparse(128, normalize=True)
    eps = pe.angle_model(torch.device("cpu"))[0]
    c = torch.tensor([[0, 1, 2, pe.TABLE_HI + 5]], dtype=torch.int32)
    split_checker.pos_aug(c, eps, pe.TABLE_LO)
    with pytest.raises(backend.StatusError):
        split_checker.check_status(torch.device("cpu"))


def test_graph_takes_the_feature_operand_path(split_checker, monkeypatch):
    """PascoNet on the oracle with the split plumbing on: every cross-attention level runs on its feature operand, and the
    result equals the K / V-operand path (PASCO_ATTN_FEAT=0)."""
    from pasco_amd.graph import PascoNet
    from pasco_amd.graph.synth import TeacherKeep, make_scene
    old = fused.MIN_ROWS_LINEAR
    fused.MIN_ROWS_LINEAR = 1
    try:
        torch.manual_seed(3)
        net = PascoNet(n_classes=20, n_infers=2, in_channels=16, f=32, num_queries=12, heavy_decoder=False).eval()
        scene = make_scene(4, n_infers=2, in_channels=16, grid=(32, 32, 8), occupancy=0.15)

        def run():
            tk = TeacherKeep(scene, torch.device("cpu"))
            with torch.no_grad():
                x = net.prepare_input(scene.in_feats, scene.in_coords)
                return net(x, scene.global_min_Cs, scene.global_max_Cs, scene.min_Cs, scene.max_Cs, keep_override=tk)

        calls = {"n": 0}
        inner = split_checker.attn_cross_feat

        def spy(*a, **kw):
            calls["n"] += 1
            return inner(*a, **kw)

        split_checker.attn_cross_feat = spy
        try:
            got = run()
        finally:
            del split_checker.attn_cross_feat
        assert calls["n"] == 3
        monkeypatch.setenv("PASCO_ATTN_FEAT", "0")
        ref = run()
        for a, b in zip(got["panop_predictions"], ref["panop_predictions"]):
            assert torch.equal(a["voxel_logits"].C, b["voxel_logits"].C)
            assert torch.allclose(a["voxel_logits"].F, b["voxel_logits"].F, rtol=2e-4, atol=2e-4)
            assert torch.allclose(a["query_logits"], b["query_logits"], rtol=2e-4, atol=2e-4)
    finally:
        fused.MIN_ROWS_LINEAR = old
