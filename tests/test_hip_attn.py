"""GPU parity of the fused masked cross-attention kernel (ph_attn_cross_fwd) against a plain torch
fp32 reference of the same op and against the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def torch_ref(q, k, v, allow):
    """q [B,H,Q,Dh] pre-scaled; k,v [B,N,H*Dh]; allow bool [B,N,Q] or None."""
    B, H, Q, Dh = q.shape
    N = k.shape[1]
    kk = k.view(B, N, H, Dh).permute(0, 2, 1, 3).double()
    vv = v.view(B, N, H, Dh).permute(0, 2, 1, 3).double()
    s = q.double() @ kk.transpose(-1, -2)                     # [B,H,Q,N]
    if allow is not None:
        al = allow.permute(0, 2, 1)                           # [B,Q,N]
        al = al | ~al.any(dim=-1, keepdim=True)               # nothing allowed -> everything allowed
        s = s.masked_fill(~al[:, None], float("-inf"))
    o = torch.softmax(s, dim=-1) @ vv
    return o.permute(0, 2, 1, 3).reshape(B, Q, H * Dh).float()


@pytest.mark.parametrize("B,H,Q,N", [(1, 8, 100, 1), (2, 8, 100, 15), (2, 8, 100, 16), (3, 8, 100, 17),
                                      (3, 8, 100, 4097), (2, 8, 128, 3000), (1, 2, 5, 70000), (3, 8, 100, 60000)])
@pytest.mark.parametrize("masked", [False, True])
def test_attn_cross_matches_torch(hip, B, H, Q, N, masked):
    g = torch.Generator().manual_seed(N + Q)
    Dh = 48
    q = (torch.randn(B, H, Q, Dh, generator=g) * Dh ** -0.5).cuda()
    k = torch.randn(B, N, H * Dh, generator=g).cuda()
    v = torch.randn(B, N, H * Dh, generator=g).cuda()
    bits = any_ = allow = None
    if masked:
        allow = torch.rand(B, N, Q, generator=g) > 0.7
        allow[:, :, 3] = False                                  # a query with nothing allowed
        if N > 20:
            allow[0, : N // 2, Q - 1] = False
        allow = allow.cuda()
        bits, any_ = hip.attn_mask_pack(allow.reshape(B * N, Q).float().contiguous(), B, N)
    got = hip.attn_cross_fwd(q, k, v, bits, any_)
    exp = torch_ref(q, k, v, allow)
    assert torch.isfinite(got).all()
    assert torch.allclose(got, exp, rtol=1e-4, atol=2e-5), float((got - exp).abs().max())


def test_attn_mask_pack_and_oracle(hip, oracle):
    g = torch.Generator().manual_seed(0)
    B, N, Q, H, Dh = 2, 777, 100, 8, 48
    vals = (torch.rand(B * N, Q, generator=g) > 0.5).float()
    vals[:, 7] = 0
    b_o, a_o = oracle.attn_mask_pack(vals, B, N)
    b_h, a_h = hip.attn_mask_pack(vals.cuda(), B, N)
    assert torch.equal(b_h.cpu(), b_o) and torch.equal(a_h.cpu(), a_o)
    assert torch.equal(hip.bits_or_reduce(b_h).cpu(), a_o) and torch.equal(oracle.bits_or_reduce(b_o), a_o)
    logits = torch.randn(B * N, Q, generator=g)
    p_o, _ = oracle.attn_mask_pack(logits, B, N, positive_only=True, want_any=False)
    p_h, _ = hip.attn_mask_pack(logits.cuda(), B, N, positive_only=True, want_any=False)
    assert torch.equal(p_h.cpu(), p_o)
    nbr = torch.randint(-1, B * N, (8, 500), generator=g).int()
    assert torch.equal(hip.bits_orpool(p_h.reshape(-1, 4), nbr.cuda()).cpu(), oracle.bits_orpool(p_o.reshape(-1, 4), nbr))
    q = torch.randn(B, H, Q, Dh, generator=g) * Dh ** -0.5
    k = torch.randn(B, N, H * Dh, generator=g)
    v = torch.randn(B, N, H * Dh, generator=g)
    exp = oracle.attn_cross_fwd(q, k, v, b_o, a_o)
    got = hip.attn_cross_fwd(q.cuda(), k.cuda(), v.cuda(), b_h, a_h).cpu()
    assert torch.allclose(got, exp, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("Q", [3, 10, 100, 128])
def test_attn_workspace_is_large_enough(hip, Q):
    """The partial-result records are sized by the kernel instantiation (112 / 128 queries), not by Q:
    run with an exactly-sized scratch followed by a canary region and check the canary survives."""
    from pasco_amd.me.backend import _ptr
    B, H, Dh, N = 2, 8, 48, 5000
    g = torch.Generator().manual_seed(77)
    q = torch.randn(B, H, Q, Dh, generator=g).cuda()
    k = torch.randn(B, N, H * Dh, generator=g).cuda()
    v = torch.randn(B, N, H * Dh, generator=g).cuda()
    need = int(hip.fn["attn_workspace_bytes"](N, B, H, Q, Dh))
    guard = 1 << 20
    ws = torch.full((need + guard,), 0x5A, dtype=torch.uint8, device="cuda")
    out = torch.empty(B, Q, H * Dh, device="cuda")
    rc = hip.fn["attn_cross_fwd"](_ptr(q), _ptr(k), _ptr(v), None, None, _ptr(out), N, B, H, Q, Dh, _ptr(ws), need,
                                  hip.stream(q.device))
    assert rc == 0
    torch.cuda.synchronize()
    assert bool((ws[need:] == 0x5A).all()), "attention kernel wrote past its declared workspace"
    ref = torch.nn.functional.scaled_dot_product_attention(
        q, k.view(B, N, H, Dh).transpose(1, 2), v.view(B, N, H, Dh).transpose(1, 2), scale=1.0)
    assert torch.allclose(out, ref.transpose(1, 2).reshape(B, Q, H * Dh), rtol=1e-4, atol=1e-4)


def _unsplit(op, c, exp2):
    """split operand [rows, c/32, 2, 32] f16 -> the fp32 values it stands for."""
    x = op.float()
    return ((x[:, :, 0] + x[:, :, 1]).reshape(op.shape[0], c) * float(2.0 ** -exp2))


@pytest.mark.parametrize("B,H,Q,N", [(1, 8, 100, 1), (2, 8, 100, 31), (2, 8, 100, 32), (3, 8, 100, 33),
                                      (3, 8, 100, 4097), (2, 8, 128, 3000), (1, 2, 5, 70000), (3, 8, 100, 60000),
                                      (2, 4, 64, 2500)])
@pytest.mark.parametrize("masked", [False, True])
def test_attn_cross_split_matches_torch(hip, B, H, Q, N, masked):
    """ph_attn_cross_split: K and V arrive as the split f16 operands a projection emits; the kernel's three-product f16
    MFMAs must reproduce fp64 attention on the values those operands stand for to fp32 accuracy."""
    g = torch.Generator().manual_seed(N + Q + 1)
    Dh = 48
    q = (torch.randn(B, H, Q, Dh, generator=g) * Dh ** -0.5).cuda()
    k = (torch.randn(B, N, H * Dh, generator=g) * 1.7).cuda()
    v = torch.randn(B, N, H * Dh, generator=g).cuda()
    ks = hip.split_rows(k.reshape(B * N, -1).contiguous())
    vs = hip.split_rows(v.reshape(B * N, -1).contiguous())
    from pasco_amd.me.backend import SPLIT_ACT_EXP2
    k2 = _unsplit(ks, H * Dh, SPLIT_ACT_EXP2).view(B, N, -1)
    v2 = _unsplit(vs, H * Dh, SPLIT_ACT_EXP2).view(B, N, -1)
    bits = any_ = allow = None
    if masked:
        allow = torch.rand(B, N, Q, generator=g) > 0.7
        allow[:, :, 3] = False                                  # a query with nothing allowed
        if N > 40:
            allow[0, : N // 2, Q - 1] = False                   # whole tiles masked for one query
        allow = allow.cuda()
        bits, any_ = hip.attn_mask_pack(allow.reshape(B * N, Q).float().contiguous(), B, N)
    got = hip.attn_cross_split(q, ks, vs, N, bits, any_)
    exp = torch_ref(q, k2, v2, allow)
    assert torch.isfinite(got).all()
    assert torch.allclose(got, exp, rtol=1e-4, atol=2e-5), float((got - exp).abs().max())
    # and it agrees with the fp32 kernel on the same values
    ref32 = hip.attn_cross_fwd(q, k2.contiguous(), v2.contiguous(), bits, any_)
    assert torch.allclose(got, ref32, rtol=1e-4, atol=2e-5)


def test_attn_cross_split_vs_oracle_and_peaked_scores(hip, oracle):
    """Oracle parity on the same operands, with score magnitudes up to ~60 (peaked softmax: the running maximum moves)."""
    g = torch.Generator().manual_seed(5)
    B, N, Q, H, Dh = 2, 1777, 100, 8, 48
    q = torch.randn(B, H, Q, Dh, generator=g) * 1.5
    k = torch.randn(B, N, H * Dh, generator=g)
    v = torch.randn(B, N, H * Dh, generator=g) * 3
    allow = torch.rand(B, N, Q, generator=g) > 0.5
    bits_o, any_o = oracle.attn_mask_pack(allow.reshape(B * N, Q).float(), B, N)
    ks_o, vs_o = oracle.split_rows(k.reshape(B * N, -1)), oracle.split_rows(v.reshape(B * N, -1))
    exp = oracle.attn_cross_split(q, ks_o, vs_o, N, bits_o, any_o)
    ks, vs = hip.split_rows(k.reshape(B * N, -1).cuda()), hip.split_rows(v.reshape(B * N, -1).cuda())
    assert torch.equal(ks.cpu(), ks_o)
    got = hip.attn_cross_split(q.cuda(), ks, vs, N, bits_o.cuda(), any_o.cuda()).cpu()
    assert torch.allclose(got, exp, rtol=2e-4, atol=1e-4), float((got - exp).abs().max())


@pytest.mark.parametrize("s", [1, 2, 4])
def test_bits_block_or_matches_oracle_and_pooling(hip, oracle, s):
    """ph_bits_block_or: OR of the fine voxels' bit rows inside every level voxel's s^3 block = bits_orpool over the stride
    map followed by a lookup of the level voxel (the path it replaces), and the oracle."""
    g = torch.Generator().manual_seed(31 + s)
    B, n1 = 2, 6000
    fine = torch.cat([torch.randint(0, B, (n1, 1), generator=g), torch.randint(0, 40, (n1, 3), generator=g)], 1).int()
    fine = torch.unique(fine, dim=0)
    fine = fine[torch.randperm(fine.shape[0], generator=g)].contiguous()
    bits1 = torch.randint(-2 ** 31, 2 ** 31 - 1, (fine.shape[0], 4), generator=g, dtype=torch.int64).int()
    # level voxels: per batch the distinct block origins (in shuffled order) padded with (0, 0, 0) rows
    per_b = []
    for b in range(B):
        org = torch.unique((fine[fine[:, 0] == b][:, 1:] // s) * s, dim=0)
        per_b.append(org[torch.randperm(org.shape[0], generator=g)])
    N = max(o.shape[0] for o in per_b) + 3
    level = torch.zeros(B, N, 4, dtype=torch.int32)
    for b, o in enumerate(per_b):
        level[b, : o.shape[0], 1:] = o
        level[b, :, 0] = b
    lo = torch.zeros(B, 3, dtype=torch.int32)
    hi = torch.full((B, 3), 39, dtype=torch.int32)
    tko, tvo, *_ = oracle.map_insert(fine, dedup=False)
    exp, rng_o = oracle.bits_block_or(level.reshape(-1, 4), N, s, tko, tvo, bits1, lo, hi, want_range=True)
    tk, tv, *_ = hip.map_insert(fine.cuda(), dedup=False)
    got, rng = hip.bits_block_or(level.reshape(-1, 4).cuda(), N, s, tk, tv, bits1.cuda(), lo.cuda(), hi.cuda(), want_range=True)
    assert torch.equal(got.cpu(), exp) and int(rng.item()) == 0 and int(rng_o.item()) == 0
    # brute force: OR over the block's fine voxels
    ref = torch.zeros_like(exp)
    for i, c in enumerate(level.reshape(-1, 4).tolist()):
        b = i // N
        inside = (fine[:, 0] == b) & ((fine[:, 1:] >= torch.tensor(c[1:])) & (fine[:, 1:] < torch.tensor(c[1:]) + s)).all(1)
        if bool(inside.any()):
            acc = bits1[inside][0].clone()
            for row in bits1[inside][1:]:
                acc |= row
            ref[i] = acc
        if i > 400:
            break
    assert torch.equal(exp[:402], ref[:402])
    # a coordinate outside the box raises the flag
    level[1, 0, 2] = -1
    _, rng = hip.bits_block_or(level.reshape(-1, 4).cuda(), N, s, tk, tv, bits1.cuda(), lo.cuda(), hi.cuda(), want_range=True)
    assert int(rng.item()) == 1


# ---- attention on the level's feature operand (ph_attn_cross_feat, ph_pos_aug) ---------------------------------------
def feat_ref(q2, x_split, aug, B, N, allow):
    """fp64 restatement: rows r = [x | aug] (x = hi + lo of the operand, unscaled), Y = softmax(q2 r^T + mask) r."""
    from pasco_amd.me.backend import SPLIT_ACT_EXP2
    xs = x_split.double()
    x = (xs[:, :, 0] + xs[:, :, 1]).reshape(B, N, -1) * 2.0 ** -SPLIT_ACT_EXP2
    # the position columns carry the operand's 2^exp2 like the feature columns (ph_pos_aug, round 5)
    r = torch.cat([x, aug.double().reshape(B, N, 16) * 2.0 ** -SPLIT_ACT_EXP2], dim=-1)             # [B, N, E]
    s = torch.einsum("bhqe,bne->bhqn", q2.double(), r)
    if allow is not None:
        al = allow.permute(0, 2, 1)
        al = al | ~al.any(dim=-1, keepdim=True)
        s = s.masked_fill(~al[:, None], float("-inf"))
    y = torch.einsum("bhqn,bne->bhqe", torch.softmax(s, dim=-1), r)        # [B, H, Q, E]
    H, Q, E = y.shape[1:]
    return y.permute(0, 2, 1, 3).reshape(B, Q, H * E).float()


def feat_inputs(B, H, Q, N, seed, hip):
    from pasco_amd.graph.transformer import PositionEmbeddingSineSparse
    g = torch.Generator().manual_seed(seed)
    C = 64
    x = torch.randn(B * N, C, generator=g) * torch.rand(B * N, 1, generator=g) * 3
    coords = torch.randint(-2, 260, (B * N, 4), generator=g, dtype=torch.int32)
    coords[: max(1, N // 7), 1:] = torch.randint(0, 3, (max(1, N // 7), 3), generator=g, dtype=torch.int32)
    q2 = torch.randn(B, H, Q, C + 16, generator=g) * 48 ** -0.5
    q2[..., C + 6:] = 0                                              # unused position columns
    q2[..., C + 3:C + 6] *= 2.0 ** -14
    pe = PositionEmbeddingSineSparse(128, normalize=True)
    eps = pe.angle_model(torch.device("cuda"))[0]
    return x.cuda(), coords.cuda(), q2.cuda(), eps, pe.TABLE_LO


@pytest.mark.parametrize("B,H,Q,N", [(1, 8, 100, 1), (2, 8, 100, 31), (2, 8, 100, 32), (3, 8, 100, 33), (3, 8, 100, 4097),
                                      (2, 8, 128, 3000), (1, 2, 5, 70000), (3, 8, 100, 60000)])
@pytest.mark.parametrize("masked", [False, True])
def test_attn_cross_feat_matches_fp64(hip, B, H, Q, N, masked):
    x, coords, q2, eps, lo = feat_inputs(B, H, Q, N, N + Q, hip)
    x_split = hip.split_rows(x)
    aug = hip.pos_aug(coords, eps, lo)
    bits = any_ = allow = None
    if masked:
        g = torch.Generator().manual_seed(N)
        allow = torch.rand(B, N, Q, generator=g) > 0.7
        allow[:, :, 3] = False
        if N > 20:
            allow[0, : N // 2, Q - 1] = False
        allow = allow.cuda()
        bits, any_ = hip.attn_mask_pack(allow.reshape(B * N, Q).float().contiguous(), B, N)
    got = hip.attn_cross_feat(q2, x_split, aug, N, bits, any_)
    exp = feat_ref(q2, x_split, aug, B, N, allow)
    hip.check_status(x.device)
    assert torch.isfinite(got).all()
    assert torch.allclose(got, exp, rtol=1e-4, atol=2e-5), float((got - exp).abs().max())


def test_attn_feat_position_coefficients_share_the_feature_range(hip):
    """the Q2 position columns used to be split as q * 2^8 * 2^exp2, so a coefficient above ~8 raised the f16
    range flag (status bit 0) and cost a whole-step redo on the exact path - a cliff trained weights could hit silently.
    The key columns now carry the 2^exp2 (ph_pos_aug), the coefficients meet the flag only above 256 like the feature
    columns: coefficients of ~100 give fp64-accurate results and no flag; above 256 the flag still fires."""
    B, H, Q, N = 2, 8, 100, 3000
    x, coords, q2, eps, lo = feat_inputs(B, H, Q, N, 21, hip)
    q2 = q2.clone()
    q2[..., 64:70] *= 100.0 / float(q2[..., 64:70].abs().max())
    xs, aug = hip.split_rows(x), hip.pos_aug(coords, eps, lo)
    got = hip.attn_cross_feat(q2, xs, aug, N)
    hip.check_status(x.device)                        # no flag
    exp = feat_ref(q2, xs, aug, B, N, None)
    assert torch.allclose(got, exp, rtol=1e-4, atol=1e-4), float((got - exp).abs().max())
    q2[0, 0, 0, 64] = 300.0
    hip.attn_cross_feat(q2, xs, aug, N)
    from pasco_amd.me.backend import F16RangeError
    with pytest.raises(F16RangeError):
        hip.check_status(x.device)


def test_pos_aug_and_feat_attention_match_the_oracle(hip, oracle):
    B, H, Q, N = 2, 8, 100, 777
    x, coords, q2, eps, lo = feat_inputs(B, H, Q, N, 5, hip)
    aug_h = hip.pos_aug(coords, eps, lo)
    aug_o = oracle.pos_aug(coords.cpu(), eps.cpu(), lo)
    assert torch.equal(aug_h.cpu().view(torch.int16), aug_o.view(torch.int16))
    xs_h = hip.split_rows(x)
    xs_o = oracle.split_rows(x.cpu())
    assert torch.equal(xs_h.cpu().view(torch.int16), xs_o.view(torch.int16))
    g = torch.Generator().manual_seed(1)
    allow = (torch.rand(B * N, Q, generator=g) > 0.5).float()
    b_o, a_o = oracle.attn_mask_pack(allow, B, N)
    b_h, a_h = hip.attn_mask_pack(allow.cuda(), B, N)
    exp = oracle.attn_cross_feat(q2.cpu(), xs_o, aug_o, N, b_o, a_o)
    got = hip.attn_cross_feat(q2, xs_h, aug_h, N, b_h, a_h).cpu()
    assert torch.allclose(got, exp, rtol=1e-3, atol=1e-4), float((got - exp).abs().max())
    # a coordinate outside the table raises the stream's status bit 2
    bad = coords.clone()
    bad[3, 2] = 100000
    hip.pos_aug(bad, eps, lo)
    from pasco_amd.me.backend import StatusError
    with pytest.raises(StatusError):
        hip.check_status(x.device)


def test_attn_cross_feat_workspace_canary(hip):
    from pasco_amd.me.backend import _ptr, SPLIT_ACT_EXP2
    B, H, Q, N = 2, 8, 100, 5000
    x, coords, q2, eps, lo = feat_inputs(B, H, Q, N, 9, hip)
    xs, aug = hip.split_rows(x), hip.pos_aug(coords, eps, lo)
    need = int(hip.fn["attn_workspace_bytes"](N, B, H, Q, 80))
    ws = torch.full((need + 4096,), 0x5A, dtype=torch.uint8, device="cuda")
    out = torch.empty((B, Q, H * 80), device="cuda")
    rc = hip.fn["attn_cross_feat"](_ptr(q2), _ptr(xs), _ptr(aug), 64, SPLIT_ACT_EXP2, None, None, _ptr(out), N, B, H, Q,
                                   _ptr(ws), need, None, hip.stream(x.device))
    assert rc == 0
    torch.cuda.synchronize()
    assert bool((ws[need:] == 0x5A).all())
    assert torch.allclose(out, feat_ref(q2, xs, aug, B, N, None), rtol=1e-4, atol=2e-5)
