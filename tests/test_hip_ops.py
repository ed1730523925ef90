"""GPU parity: every C-ABI entry point of libpascohip.so against the CPU oracle on the same seeded
inputs.  Integer / index results (unique rows, neighbour tables, COO kernel maps, pruned rows,
to_sparse coordinates) must be bit-exact; fp32 features within 1e-3 relative (north_star).
"""
import numpy as np
import pytest
import torch

from pasco_amd.me.core import kernel_offsets

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-3, 1e-4


def scene_coords(seed, n, extent=(40, 36, 12), lo=(-8, -16, 0), batch=1, dup=0.0, step=1):
    g = torch.Generator().manual_seed(seed)
    cs = []
    for b in range(batch):
        c = torch.stack([torch.randint(0, e, (n,), generator=g) for e in extent], dim=1) * step + torch.tensor(lo)
        cs.append(torch.cat([torch.full((n, 1), b), c], dim=1))
    c = torch.cat(cs).int()
    if dup > 0:
        k = int(c.shape[0] * dup)
        idx = torch.randint(0, c.shape[0], (k,), generator=g)
        c = torch.cat([c, c[idx]])[torch.randperm(c.shape[0] + k, generator=g)]
    return c.contiguous()


def unique_map(be, coords):
    tk, tv, r2u, uq, nu = be.map_insert(coords)
    c = be.gather_rows(coords, uq) if nu != coords.shape[0] else coords
    return tk, tv, c, r2u, uq


@pytest.mark.parametrize("n,dup", [(0, 0.0), (1, 0.0), (777, 0.3), (50000, 0.1), (300000, 0.0)])
def test_map_insert_exact(hip, oracle, n, dup):
    coords = scene_coords(1, n, extent=(200, 200, 30), dup=dup) if n else torch.zeros((0, 4), dtype=torch.int32)
    _, _, r2u_o, uq_o, nu_o = oracle.map_insert(coords)
    tk, tv, r2u_h, uq_h, nu_h = hip.map_insert(coords.cuda())
    assert nu_h == nu_o
    assert torch.equal(uq_h.cpu(), uq_o)
    assert torch.equal(r2u_h.cpu(), r2u_o)
    if n:
        # find every coordinate again + some misses
        q = torch.cat([coords, coords + torch.tensor([0, 1000, 0, 0], dtype=torch.int32)]).cuda()
        rows = hip.map_find(q, tk, tv).cpu()
        assert torch.equal(rows[: coords.shape[0]], r2u_o)
        assert bool((rows[coords.shape[0]:] == -1).all())


@pytest.mark.parametrize("ts", [2, 4, 8])
def test_coords_floor_expand_exact(hip, oracle, ts):
    coords = scene_coords(2, 5000, lo=(-33, -17, -5))
    assert torch.equal(hip.coords_floor(coords.cuda(), ts).cpu(), oracle.coords_floor(coords, ts))
    assert torch.equal(hip.coords_expand(coords.cuda(), ts).cpu(), oracle.coords_expand(coords, ts))


@pytest.mark.parametrize("ks,n", [(3, 20000), (2, 20000), (4, 3000), (3, 1), (3, 130)])
def test_nbr_and_coo_exact(hip, oracle, ks, n):
    coords = scene_coords(3, n)
    tk_o, tv_o, c_o, _, _ = unique_map(oracle, coords)
    tk_h, tv_h, c_h, _, _ = unique_map(hip, coords.cuda())
    assert torch.equal(c_h.cpu(), c_o)
    offs = kernel_offsets(ks, 1)
    nbr_o = oracle.nbr_build(c_o, tk_o, tv_o, offs)
    nbr_h = hip.nbr_build(c_h, tk_h, tv_h, offs)
    assert torch.equal(nbr_h.cpu(), nbr_o)
    pi_o, po_o, cnt_o = oracle.kmap_compact(nbr_o)
    pi_h, po_h, cnt_h = hip.kmap_compact(nbr_h)
    assert torch.equal(cnt_h.cpu(), cnt_o)
    for k, c in enumerate(cnt_o.tolist()):
        assert torch.equal(pi_h[k, :c].cpu(), pi_o[k, :c])
        assert torch.equal(po_h[k, :c].cpu(), po_o[k, :c])


CONV_SHAPES = [  # (kernel, cin, cout, n)
    (3, 64, 64, 30000), (3, 128, 128, 9000), (3, 256, 256, 3000), (3, 16, 20, 500), (3, 8, 8, 1),
    (3, 67, 33, 1000), (3, 64, 100, 2000), (3, 32, 160, 2000),
]


@pytest.mark.parametrize("ks,cin,cout,n", CONV_SHAPES)
def test_conv3_parity(hip, oracle, ks, cin, cout, n):
    coords = scene_coords(4, n)
    tk_o, tv_o, c_o, _, _ = unique_map(oracle, coords)
    tk_h, tv_h, c_h, _, _ = unique_map(hip, coords.cuda())
    offs = kernel_offsets(ks, 1)
    nbr_o = oracle.nbr_build(c_o, tk_o, tv_o, offs)
    nbr_h = hip.nbr_build(c_h, tk_h, tv_h, offs)
    g = torch.Generator().manual_seed(5)
    m = c_o.shape[0]
    x = torch.randn(m, cin, generator=g)
    w = torch.randn(len(offs), cin, cout, generator=g) / np.sqrt(cin * 8)
    exp = oracle.conv_fwd(x, w, nbr_o, m)
    got = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m).cpu()
    assert torch.allclose(got, exp, rtol=RTOL, atol=ATOL), float((got - exp).abs().max())


def test_conv_asymmetric_identity_weight(hip, oracle):
    """A = identity-like check with an asymmetric kernel: catches transposed C/D layouts."""
    coords = scene_coords(6, 4000)
    tk_h, tv_h, c_h, _, _ = unique_map(hip, coords.cuda())
    m = c_h.shape[0]
    cin = cout = 64
    x = torch.arange(m * cin, dtype=torch.float32).reshape(m, cin) % 97
    w = torch.zeros(27, cin, cout)
    w[13] = torch.diag(torch.arange(1, cin + 1, dtype=torch.float32))  # centre offset only
    w[13, 0, 5] = 3.0  # asymmetric entry
    nbr = hip.nbr_build(c_h, tk_h, tv_h, kernel_offsets(3, 1))
    got = hip.conv_fwd(x.cuda(), w.cuda(), nbr, m).cpu()
    exp = x @ w[13]
    assert torch.allclose(got, exp, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("cin,cout", [(64, 128), (259, 256), (64, 20), (192, 64), (131, 128)])
def test_conv1_and_strided_parity(hip, oracle, cin, cout):
    g = torch.Generator().manual_seed(7)
    n = 5000
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cin, cout, generator=g) / np.sqrt(cin)
    b = torch.randn(cout, generator=g)
    exp = oracle.conv_fwd(x, w, None, n, bias=b)
    got = hip.conv_fwd(x.cuda(), w.cuda(), None, n, bias=b.cuda()).cpu()
    assert torch.allclose(got, exp, rtol=RTOL, atol=ATOL)
    assert torch.allclose(got, x @ w + b, rtol=RTOL, atol=ATOL)


def test_strided_and_transposed_parity(hip, oracle):
    coords = scene_coords(8, 20000)
    g = torch.Generator().manual_seed(9)
    res = {}
    for name, be, dev in (("o", oracle, "cpu"), ("h", hip, "cuda")):
        tk, tv, c, _, _ = unique_map(be, coords.to(dev))
        fl = be.coords_floor(c, 2)
        tk2, tv2, c2, r2u, _ = unique_map(be, fl)
        nbr_dn = be.nbr_build(c2, tk, tv, kernel_offsets(2, 1))
        kids = be.coords_expand(c2, 1)
        tk3, tv3, c3, _, _ = unique_map(be, kids)
        nbr_up = be.nbr_build(c3, tk2, tv2, kernel_offsets(2, 1, transposed=True))
        res[name] = (c, c2, nbr_dn, c3, nbr_up)
    for a, b in zip(res["o"], res["h"]):
        assert torch.equal(a, b.cpu())
    c, c2, nbr_dn, c3, nbr_up = res["o"]
    x = torch.randn(c.shape[0], 64, generator=g)
    wd = torch.randn(8, 64, 128, generator=g) / 16
    wu = torch.randn(8, 128, 64, generator=g) / 16
    dn_o = oracle.conv_fwd(x, wd, nbr_dn, c2.shape[0])
    dn_h = hip.conv_fwd(x.cuda(), wd.cuda(), res["h"][2], c2.shape[0])
    assert torch.allclose(dn_h.cpu(), dn_o, rtol=RTOL, atol=ATOL)
    up_o = oracle.conv_fwd(dn_o, wu, nbr_up, c3.shape[0])
    up_h = hip.conv_fwd(dn_h, wu.cuda(), res["h"][4], c3.shape[0])
    assert torch.allclose(up_h.cpu(), up_o, rtol=RTOL, atol=ATOL)


def test_fused_prologue_epilogue_parity(hip, oracle):
    coords = scene_coords(10, 12000)
    tk_o, tv_o, c_o, _, _ = unique_map(oracle, coords)
    tk_h, tv_h, c_h, _, _ = unique_map(hip, coords.cuda())
    offs = kernel_offsets(3, 1)
    nbr_o = oracle.nbr_build(c_o, tk_o, tv_o, offs)
    nbr_h = hip.nbr_build(c_h, tk_h, tv_h, offs)
    g = torch.Generator().manual_seed(11)
    m, cin, cout = c_o.shape[0], 64, 64
    x = torch.randn(m, cin, generator=g)
    w = torch.randn(27, cin, cout, generator=g) / 40
    ps, pb = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.2
    es, eb = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    bias, res = torch.randn(cout, generator=g), torch.randn(m, cout, generator=g)
    for pro_act, epi_act, res_act in ((1, 1, 1), (0, 2, 0), (1, 0, 1), (2, 2, 2)):
        kw = dict(bias=bias, pro_scale=ps, pro_shift=pb, pro_act=pro_act, epi_scale=es, epi_shift=eb,
                  epi_act=epi_act, slope=0.01, residual=res, res_act=res_act)
        exp = oracle.conv_fwd(x, w, nbr_o, m, **kw)
        kw_h = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
        got = hip.conv_fwd(x.cuda(), w.cuda(), nbr_h, m, **kw_h).cpu()
        assert torch.allclose(got, exp, rtol=RTOL, atol=ATOL), (pro_act, epi_act, res_act)


@pytest.mark.parametrize("s,c", [(2, 100), (4, 100), (2, 7)])
def test_maxpool_parity(hip, oracle, s, c):
    coords = scene_coords(12, 30000)
    outs = []
    for be, dev in ((oracle, "cpu"), (hip, "cuda")):
        tk, tv, cc, _, _ = unique_map(be, coords.to(dev))
        _, _, c2, _, _ = unique_map(be, be.coords_floor(cc, s))
        nbr = be.nbr_build(c2, tk, tv, kernel_offsets(s, 1))
        x = torch.randn(cc.shape[0], c, generator=torch.Generator().manual_seed(13)).to(dev)
        outs.append((c2.cpu(), be.maxpool_fwd(x, nbr).cpu()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


def test_rows_and_dense_exact(hip, oracle):
    g = torch.Generator().manual_seed(14)
    n, c = 40000, 64
    x = torch.randn(n, c, generator=g)
    mask = torch.rand(n, generator=g) > 0.6
    keep_o = oracle.mask_compact(mask)
    keep_h = hip.mask_compact(mask.cuda())
    assert torch.equal(keep_h.cpu(), keep_o)
    assert torch.equal(hip.gather_rows(x.cuda(), keep_h).cpu(), x[mask])
    for cc in (3, 4, 20):
        y = torch.randn(n, cc, generator=g)
        assert torch.equal(hip.gather_rows(y.cuda(), keep_h).cpu(), y[mask])
    # empty / full masks
    assert hip.mask_compact(torch.zeros(n, dtype=torch.bool).cuda()).numel() == 0
    assert hip.mask_compact(torch.ones(5, dtype=torch.bool).cuda()).tolist() == [0, 1, 2, 3, 4]
    # scatter-add with unique targets
    rows = torch.randperm(n, generator=g)[: n // 2].int()
    dst = torch.randn(n, c, generator=g)
    exp = oracle.scatter_add_rows(x[: n // 2].contiguous(), rows, dst.clone())
    got = hip.scatter_add_rows(x[: n // 2].contiguous().cuda(), rows.cuda(), dst.clone().cuda()).cpu()
    assert torch.equal(got, exp)
    # dense round trip
    coords = scene_coords(15, 20000, extent=(64, 48, 16), lo=(-16, -8, 0), batch=2)
    _, _, cu, _, _ = unique_map(oracle, coords)
    f = torch.randn(cu.shape[0], 24, generator=g)
    f[::5] = 0
    dims = (2, 64, 48, 16)
    d_o = oracle.to_dense(f, cu, (-16, -8, 0), 1, dims)
    d_h = hip.to_dense(f.cuda(), cu.cuda(), (-16, -8, 0), 1, dims)
    assert torch.equal(d_h.cpu(), d_o)
    co, fo = oracle.to_sparse(d_o)
    ch, fh = hip.to_sparse(d_h)
    assert torch.equal(ch.cpu(), co) and torch.equal(fh.cpu(), fo)
    assert co.shape[0] == int((f != 0).any(dim=1).sum())



def test_sine_pe_matches_oracle_and_torch(hip, oracle):
    from pasco_amd.graph.transformer import PositionEmbeddingSineSparse, sine_position_encoding
    g = torch.Generator().manual_seed(90)
    c4 = torch.randint(-40, 300, (7001, 4), generator=g, dtype=torch.int32)
    c4[:50, 1:] = 0
    c4[50:60, 1] = 5000          # outside the lookup table of the module: evaluated in the kernel
    c4[60:70, 3] = -2000
    pe = PositionEmbeddingSineSparse(128, normalize=True)
    exp = oracle.sine_pe(c4, pe.dim_t(torch.device("cpu")), pe.scale, coff=1)
    got = pe(c4.cuda(), coff=1)
    assert got.shape == (7001, 384)
    assert torch.allclose(got.cpu(), exp, atol=2e-6, rtol=0)
    ref = sine_position_encoding(c4.cuda()[:, 1:], 128)
    assert torch.allclose(got, ref, atol=2e-6, rtol=0)
    # lookup and evaluation are the same numbers
    direct = hip.sine_pe(c4.cuda(), pe.dim_t(torch.device("cuda", 0)), pe.scale, coff=1)
    assert torch.equal(got, direct)


def test_gram_batched_matches_matmul(hip):
    from pasco_amd.graph.ensemble import _gram
    g = torch.Generator().manual_seed(91)
    a, b = torch.rand(100003, 100, generator=g).cuda(), torch.rand(100003, 100, generator=g).cuda()
    ref = (a.double().t() @ b.double()).float()
    assert torch.allclose(_gram(a, b), ref, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("c,m", [(20, 3), (19, 3), (20, 1), (20, 8)])
def test_sem_ensemble_matches_oracle_and_torch(hip, oracle, c, m):
    """ph_sem_ensemble: softmax + resampling + class-0 fill + mean + confidences in one pass, against the oracle and the
    torch formulation of Ensembler.ensemble_sem_compl (ensembler.py:159-187)."""
    g = torch.Generator().manual_seed(70 + c + m)
    n_sites = 50001
    logits, rows = [], []
    for i in range(m):
        n_i = 3000 + 517 * i
        logits.append(torch.randn(n_i, c, generator=g) * 3)
        r = torch.randint(-n_i // 2, n_i, (n_sites,), generator=g).clamp(min=-1).int()
        rows.append(r)
    o_out, o_conf = oracle.sem_ensemble(logits, rows)
    h_out, h_conf = hip.sem_ensemble([t.cuda() for t in logits], [t.cuda() for t in rows])
    ref = []
    for i in range(m):
        p = torch.softmax(logits[i], dim=-1)
        d = torch.zeros(n_sites, c)
        ok = rows[i] >= 0
        d[ok] = p[rows[i][ok].long()]
        d[~ok, 0] = 1.0
        ref.append(d)
    ref.append(torch.stack(ref).mean(0))
    for i in range(m + 1):
        assert torch.allclose(h_out[i].cpu(), o_out[i], rtol=1e-5, atol=1e-6)
        assert torch.allclose(o_out[i], ref[i], rtol=1e-5, atol=1e-6)
        assert torch.allclose(h_conf[i].cpu(), ref[i].max(dim=1)[0], rtol=1e-5, atol=1e-6)
        assert torch.allclose(o_conf[i], ref[i].max(dim=1)[0], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("q", [100, 128, 5, 64])
def test_ens_row_kernels_match_oracle_and_torch(hip, oracle, q):
    """ph_ens_resample / ph_ens_merge / ph_ens_finish against the oracle and the torch formulation they replace
    (Ensembler.ensemble_panop; reference ensembler.py:44-62, 86-98, 100-118)."""
    g = torch.Generator().manual_seed(900 + q)
    n, n_sites, c = 7001, 60000, 20
    logits = torch.randn(n, q, generator=g) * 4
    logits[5] = -200.0                                              # sigmoid underflows to exactly 0: flag must be 0
    rows = torch.randint(-n, n, (n_sites,), generator=g).clamp(min=-1).int()
    sel = torch.randperm(n_sites, generator=g)[:20011].sort()[0].int()
    rows[sel[7].item()] = 5
    # resample
    o_out, o_flag = oracle.ens_resample(logits, rows, sel)
    h_out, h_flag = hip.ens_resample(logits.cuda(), rows.cuda(), sel.cuda())
    r = rows[sel.long()]
    ref = torch.where((r >= 0)[:, None], torch.sigmoid(logits[r.clamp(min=0).long()]), torch.zeros(1))
    assert torch.allclose(o_out, ref, rtol=1e-6, atol=1e-7) and torch.allclose(h_out.cpu(), o_out, rtol=2e-6, atol=1e-7)
    assert torch.equal(o_flag.bool(), (o_out != 0).any(1)) and torch.equal(h_flag.cpu().bool(), (h_out.cpu() != 0).any(1))
    assert not bool(o_flag[7]) and not bool(h_flag[7])
    # merge: bit-exact (three rounded fp32 operations)
    anchor = torch.rand(sel.shape[0], q, generator=g)
    m = torch.rand(sel.shape[0], q, generator=g)
    perm = torch.randperm(q, generator=g).int()
    for i in (1, 2, 5):
        ref_m = (anchor * i + m[:, perm.long()]) / (i + 1)
        o = oracle.ens_merge(anchor.clone(), m, perm, i)
        h = hip.ens_merge(anchor.clone().cuda(), m.cuda(), perm.cuda(), i).cpu()
        assert torch.equal(o, ref_m) and torch.equal(h, ref_m)
    # finish
    sem = torch.softmax(torch.randn(n_sites, c, generator=g), dim=-1)
    sem[sel[3].item()] = 0.05                                       # a tie: the first maximum (class 0) wins
    keep = torch.randperm(q, generator=g)[: max(1, q // 3)].sort()[0].int()
    nzc = (sem[sel.long()].argmax(dim=1) != 0).float()
    assert nzc[3] == 0
    ref_f = anchor[:, keep.long()] * nzc[:, None]
    o_out, o_flag = oracle.ens_finish(anchor, keep, sem, sel)
    h_out, h_flag = hip.ens_finish(anchor.cuda(), keep.cuda(), sem.cuda(), sel.cuda())
    assert torch.equal(o_out, ref_f) and torch.equal(h_out.cpu(), ref_f)
    assert torch.equal(o_flag.bool(), (ref_f != 0).any(1)) and torch.equal(h_flag.cpu().bool(), (ref_f != 0).any(1))
    e_out, e_flag = hip.ens_finish(anchor.cuda(), keep[:0].cuda(), sem.cuda(), sel.cuda())     # no query kept
    assert e_out.shape == (sel.shape[0], 0) and not bool(e_flag.any())


def test_coordinate_range_guard_matches_oracle(hip, oracle):
    """Out-of-range coordinates: flagged on insert, -1 on lookup, no neighbour across the edge (same as the oracle)."""
    dev = torch.device("cuda", 0)
    hip.status_word(dev).zero_()                                # sticky flag: other tests may have raised it on purpose
    g = torch.Generator().manual_seed(12)
    c = torch.randint(-131072, 131072, (5000, 4), generator=g).int()
    c[:, 0] = torch.randint(0, 1024, (5000,), generator=g).int()
    tk, tv, *_ = hip.map_insert(c.cuda())
    hip.check_status(dev)
    q = c.clone()
    q[::3, 1] += 1 << 18                                        # would alias row i with a plain mask
    q[1::3, 0] += 1024
    exp = oracle.map_find(q, *oracle.map_insert(c)[:2])
    got = hip.map_find(q.cuda(), tk, tv).cpu()
    assert torch.equal(got, exp) and bool((got[::3] == -1).all()) and bool((got[1::3] == -1).all())
    edge = torch.tensor([[0, 131071, 0, 0], [0, -131072, 0, 0]], dtype=torch.int32)
    offs = [(1, 0, 0), (-1, 0, 0), (0, 0, 0)]
    tk2, tv2, *_ = hip.map_insert(edge.cuda())
    assert hip.nbr_build(edge.cuda(), tk2, tv2, offs).cpu().tolist() == [[-1, -1], [-1, -1], [0, 1]]
    hip.map_insert(torch.tensor([[0, 0, 0, -131073]], dtype=torch.int32).cuda())
    with pytest.raises(RuntimeError, match="packable range"):
        hip.check_status(dev)
    hip.check_status(dev)


@pytest.mark.parametrize("ks,dil,n", [(3, 1, 30000), (3, 2, 5000), (5, 1, 300), ((3, 1, 3), 1, 2000)])
def test_nbr_build_same_map_equals_plain_build(hip, oracle, ks, dil, n):
    """ph_nbr_build_same (half the probes + mirrored writes) gives the table of ph_nbr_build and of the oracle."""
    from pasco_amd.me.core import kernel_offsets
    g = torch.Generator().manual_seed(77 + n)
    c = torch.cat([torch.randint(0, 3, (n, 1), generator=g), torch.randint(-20, 20, (n, 3), generator=g)], 1).int()
    c = torch.unique(c, dim=0)
    c = c[torch.randperm(c.shape[0], generator=g)].contiguous()
    offs = kernel_offsets(ks, 1, dil, False)
    if len(offs) > 64:
        pytest.skip("kernel volume beyond one nbr_build call")
    tk, tv, *_ = hip.map_insert(c.cuda(), dedup=False)
    plain = hip.nbr_build(c.cuda(), tk, tv, offs)
    same = hip.nbr_build(c.cuda(), tk, tv, offs, same_map=True)
    assert torch.equal(plain, same)
    tko, tvo, *_ = oracle.map_insert(c, dedup=False)
    assert torch.equal(same.cpu(), oracle.nbr_build(c, tko, tvo, offs, same_map=True))
    assert bool((same[len(offs) // 2].cpu() == torch.arange(c.shape[0])).all())


def test_project_canonical_matches_oracle_bit_for_bit(hip, oracle):
    """ph_project_canonical: the canonical grid through a subnet's transform (transform_utils.py:60-74), same integers as
    the C restatement (which equals the torch formula: tests/test_oracle_properties.py) on the full 256 x 256 x 32 grid."""
    import numpy as np
    from pasco_amd.graph.ensemble import CANONICAL_SIZE, MIN_BOUND, RESOLUTION
    g = torch.Generator().manual_seed(8)
    for k in range(4):
        th = float(torch.rand(1, generator=g)) * 6.28
        T = torch.eye(4)
        T[0, 0], T[0, 1], T[1, 0], T[1, 1] = np.cos(th), -np.sin(th), np.sin(th), np.cos(th)
        if k % 2:
            T[0] = -T[0]
        T[:3, 3] = torch.randn(3, generator=g) * 2
        exp = oracle.project_canonical(T, CANONICAL_SIZE, RESOLUTION, MIN_BOUND)
        got = hip.project_canonical(T.cuda(), CANONICAL_SIZE, RESOLUTION, MIN_BOUND).cpu()
        assert torch.equal(got, exp)


def test_keep_mask_matches_oracle(hip, oracle):
    """ph_keep_mask (decoder keep masks in one pass) bit for bit against the oracle: both source kinds, OR over several
    sources, box test, the "nothing kept -> first rows" fallback decided on the device, ragged sizes."""
    g = torch.Generator().manual_seed(11)
    for n in (1, 63, 64, 1000, 1001, 70001):
        coords = torch.randint(-5, 60, (n, 4), generator=g, dtype=torch.int32)
        lo, hi = torch.tensor([0, 3, -2], dtype=torch.int32), torch.tensor([40, 50, 30], dtype=torch.int32)
        rows = [torch.randint(-3, 2, (n,), generator=g, dtype=torch.int32) for _ in range(3)]
        masks = [(r >= 0).contiguous() for r in rows]
        none = torch.full((n,), -1, dtype=torch.int32)
        cases = [dict(srcs=rows), dict(srcs=masks), dict(srcs=[rows[0]], coords=coords, lo=lo, hi=hi, fallback_rows=1000),
                 dict(srcs=[none], coords=coords, lo=lo, hi=hi, fallback_rows=1000), dict(srcs=[none], coords=coords, lo=lo, hi=hi),
                 dict(srcs=[masks[1]], coords=coords, lo=lo, hi=hi, fallback_rows=5)]
        for kw in cases:
            exp = oracle.keep_mask(**kw)
            dkw = {k: ([t.cuda() for t in v] if k == "srcs" else (v.cuda() if torch.is_tensor(v) else v)) for k, v in kw.items()}
            got = hip.keep_mask(**dkw)
            assert got.dtype == torch.bool and torch.equal(got.cpu(), exp), (n, sorted(kw))
