"""Property tests (hypothesis) of the coordinate-map semantics on the CPU oracle: negative
coordinates, duplicates, empty inputs, strides (SURVEY.md section 4 plan item 4)."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from pasco_amd.me.core import kernel_offsets

coords_st = st.lists(st.tuples(st.integers(0, 2), st.integers(-20, 20), st.integers(-20, 20), st.integers(-9, 9)),
                     min_size=0, max_size=120)


def as_tensor(rows):
    return torch.tensor(rows, dtype=torch.int32).reshape(-1, 4).contiguous()


@settings(max_examples=60, deadline=None)
@given(coords_st)
def test_insert_first_occurrence(oracle, rows):
    c = as_tensor(rows)
    tk, tv, r2u, uq, nu = oracle.map_insert(c)
    seen, order = {}, []
    for i, r in enumerate(rows):
        if r not in seen:
            seen[r] = len(order)
            order.append(i)
    assert nu == len(order) and uq.tolist() == order
    assert r2u.tolist() == [seen[r] for r in rows]
    if rows:
        assert oracle.map_find(c, tk, tv).tolist() == r2u.tolist()


@settings(max_examples=40, deadline=None)
@given(coords_st, st.sampled_from([2, 4, 8]))
def test_floor_is_python_floor(oracle, rows, ts):
    c = as_tensor(rows)
    out = oracle.coords_floor(c, ts).tolist()
    assert out == [[b, (x // ts) * ts, (y // ts) * ts, (z // ts) * ts] for b, x, y, z in rows]


@settings(max_examples=30, deadline=None)
@given(coords_st)
def test_kernel_map_is_symmetric(oracle, rows):
    c = as_tensor(rows)
    tk, tv, _, uq, nu = oracle.map_insert(c)
    u = c[uq.long()].contiguous()
    if nu == 0:
        return
    nbr = oracle.nbr_build(u, tk, tv, kernel_offsets(3, 1))
    for k in range(27):
        for o, i in enumerate(nbr[k].tolist()):
            if i >= 0:
                assert nbr[26 - k][i].item() == o
    assert nbr[13].tolist() == list(range(nu))


@settings(max_examples=30, deadline=None)
@given(coords_st)
def test_expand_children_have_their_parent(oracle, rows):
    c = as_tensor([(b, 2 * x, 2 * y, 2 * z) for b, x, y, z in rows])
    tk, tv, _, uq, nu = oracle.map_insert(c)
    par = c[uq.long()].contiguous()
    kids = oracle.coords_expand(par, 1)
    tk2, tv2, r2u, uq2, nk = oracle.map_insert(kids)
    assert nk == 8 * nu                                   # aligned parents generate disjoint children
    if nu:
        nbr = oracle.nbr_build(kids, tk, tv, kernel_offsets(2, 1, transposed=True))
        assert int((nbr >= 0).sum()) == 8 * nu            # exactly one (parent, offset) per child
        assert nbr.reshape(8, nu, 8)[0, :, 0].tolist() == list(range(nu))


def test_oracle_split_rows_matches_numpy_float16(oracle):
    """pho_split_rows (integer restatement of IEEE binary16 rounding) against numpy's float16 conversion,
    including subnormals, ties and the overflow threshold; layout [n][cpad/32][hi x32 | lo x32]."""
    rng = np.random.default_rng(5)
    n, c = 257, 40
    x = (rng.standard_normal((n, c)) * np.exp(rng.standard_normal((n, 1)) * 6)).astype(np.float32)
    x = np.clip(x, -65000, 65000)
    specials = np.array([0.0, -0.0, 65504.0, 65519.9, 6.1035156e-05, 6.0975552e-05, 5.9604645e-08, 2.9802322e-08,
                         2.9802326e-08, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, -(1.0 + 2.0 ** -11), 1e-10], np.float32)
    x[0, :specials.size] = specials
    got = oracle.split_rows(torch.from_numpy(x), exp2=0).numpy()           # [n, 2, 2, 32] f16
    assert got.shape == (n, 2, 2, 32)
    # an activation operand (exp2 = 5) is the split of x * 32: small values keep their lo half
    xs = np.clip(x, -2000, 2000)
    got5 = oracle.split_rows(torch.from_numpy(xs)).numpy()
    h5 = (xs * 32).astype(np.float16)
    l5 = (xs * 32 - h5.astype(np.float32)).astype(np.float16)
    assert np.array_equal(got5[:, :, 0, :].reshape(n, 64)[:, :c].view(np.int16), h5.view(np.int16))
    assert np.array_equal(got5[:, :, 1, :].reshape(n, 64)[:, :c].view(np.int16), l5.view(np.int16))
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    pad = np.zeros((n, 64), np.float16)
    exp_hi, exp_lo = pad.copy(), pad.copy()
    exp_hi[:, :c], exp_lo[:, :c] = hi, lo
    assert np.array_equal(got[:, :, 0, :].reshape(n, 64).view(np.int16), exp_hi.view(np.int16))
    assert np.array_equal(got[:, :, 1, :].reshape(n, 64).view(np.int16), exp_lo.view(np.int16))
    # hi + lo reproduces x to 2^-21 relative, or to half an f16 subnormal step (2^-25) where lo underflows
    # (what the three-product scheme relies on)
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    err = np.abs(rec - x.astype(np.float64))
    assert np.all(err <= np.maximum(np.abs(x) * 2.0 ** -21, 2.0 ** -25))


def test_prune_same_mask_is_one_map_event(oracle_registered):
    """Two prunes of one map with the SAME mask tensor share rows and key; a changed mask (new tensor or in-place
    edit) is a new event."""
    import pasco_amd.me as ME
    g = torch.Generator().manual_seed(3)
    c = torch.unique(torch.randint(0, 12, (300, 4), generator=g, dtype=torch.int32), dim=0)
    c[:, 0] = 0
    c = torch.unique(c, dim=0).contiguous()
    x = ME.SparseTensor(torch.randn(c.shape[0], 5, generator=g), c)
    y = ME.SparseTensor(torch.randn(c.shape[0], 3, generator=g), coordinate_map_key=x.coordinate_map_key,
                        coordinate_manager=x.coordinate_manager)
    keep = torch.rand(c.shape[0], generator=g) > 0.4
    prune = ME.MinkowskiPruning()
    px, py = prune(x, keep), prune(y, keep)
    assert px.coordinate_map_key == py.coordinate_map_key
    assert torch.equal(px.C, c[keep]) and torch.equal(px.F, x.F[keep]) and torch.equal(py.F, y.F[keep])
    keep[0] = not bool(keep[0])                                   # in-place edit: version changes
    pz = prune(x, keep)
    assert torch.equal(pz.C, c[keep]) and torch.equal(pz.F, x.F[keep])
    assert pz.coordinate_map_key != px.coordinate_map_key


def test_coordinates_outside_the_key_range_are_flagged_not_aliased(oracle):
    """The 64-bit key holds batch 0..1023 and coordinates -2^17 .. 2^17-1: an insert beyond that raises the status flag
    (check_status), lookups beyond it answer -1 instead of the row of the voxel they would alias."""
    dev = torch.device("cpu")
    oracle.status_word(dev).zero_()                            # sticky flag: other tests may have raised it on purpose
    ok = torch.tensor([[0, 1, 2, 3], [3, -131072, 131071, 0], [1023, 5, 5, 5]], dtype=torch.int32)
    tk, tv, *_ = oracle.map_insert(ok)
    oracle.check_status(dev)                                   # nothing raised
    alias = torch.tensor([[0, 1 + (1 << 18), 2, 3], [1024, 1, 2, 3], [0, 1, 2, 3]], dtype=torch.int32)
    assert oracle.map_find(alias, tk, tv).tolist() == [-1, -1, 0]
    oracle.map_insert(torch.tensor([[0, 131072, 0, 0]], dtype=torch.int32))
    with pytest.raises(RuntimeError, match="packable range"):
        oracle.check_status(dev)
    oracle.check_status(dev)                                   # the flag was cleared by the read
    # neighbours across the edge of the range do not exist
    edge = torch.tensor([[0, 131071, 0, 0], [0, -131072, 0, 0]], dtype=torch.int32)
    tk, tv, *_ = oracle.map_insert(edge)
    offs = [(1, 0, 0), (-1, 0, 0), (0, 0, 0)]
    nbr = oracle.nbr_build(edge, tk, tv, offs)
    assert nbr.tolist() == [[-1, -1], [-1, -1], [0, 1]]


def test_project_canonical_kernel_restatement_equals_the_torch_formula(oracle):
    """pho_project_canonical (the restatement the HIP kernel is checked against) = graph/ensemble.py::project_canonical, the
    reference's `transform` (transform_utils.py:60-74), bit for bit on rotated / translated / flipped transforms."""
    from pasco_amd.graph.ensemble import MIN_BOUND, RESOLUTION, canonical_sites, project_canonical
    size = (40, 36, 12)
    sites = canonical_sites(size, torch.device("cpu"))
    g = torch.Generator().manual_seed(3)
    for k in range(6):
        th = float(torch.rand(1, generator=g)) * 6.28
        T = torch.eye(4)
        T[0, 0], T[0, 1], T[1, 0], T[1, 1] = np.cos(th), -np.sin(th), np.sin(th), np.cos(th)
        if k % 2:
            T[1] = -T[1]
        T[:3, 3] = torch.randn(3, generator=g) * 3
        exp = project_canonical(sites, T)
        got = oracle.project_canonical(T, size, RESOLUTION, MIN_BOUND)
        assert torch.equal(got[:, 1:], exp) and bool((got[:, 0] == 0).all())


def test_stride_chain_equals_step_by_step_strides(oracle_registered):
    """Encoder maps built together from the input rows (one host read) == stride(stride(stride(x, 2), 2), 2): same
    coordinates in the same row order at every level, and the cached keys serve the step-by-step calls."""
    import pasco_amd.me as ME
    g = torch.Generator().manual_seed(21)
    c = torch.cat([torch.zeros(3000, 1, dtype=torch.long), torch.randint(-40, 90, (3000, 3), generator=g)], dim=1).int()
    a = ME.SparseTensor(torch.zeros(3000, 1), c)
    b = ME.SparseTensor(torch.zeros(3000, 1), c)
    ma, mb = a.coordinate_manager, b.coordinate_manager
    chain = ma.stride_chain(a.coordinate_map_key, 3)
    k = b.coordinate_map_key
    for l in range(3):
        k = mb.stride(k, 2)
        assert chain[l].tensor_stride == k.tensor_stride
        assert torch.equal(ma.get_coordinates(chain[l]), mb.get_coordinates(k)), f"level {l}"
    k1 = ma.stride(a.coordinate_map_key, 2)
    assert k1 == chain[0] and ma.stride(ma.stride(k1, 2), 2) == chain[2]
    # kernel maps of the strided convolutions come out the same
    n1 = ma.kernel_map(chain[0], chain[1], 2)
    n2 = mb.kernel_map(mb.stride(b.coordinate_map_key, 2), mb.stride(mb.stride(b.coordinate_map_key, 2), 2), 2)
    assert torch.equal(n1, n2)


def test_rowlist_promise_is_checked_on_the_device(oracle):
    """Row lists serve maps with exactly one pair per output row; a 3x3x3 map handed over as such raises status bit 5 at the
    next check instead of leaving rows unwritten."""
    from pasco_amd.me.backend import StatusError
    from pasco_amd.me.core import kernel_offsets
    dev = torch.device("cpu")
    oracle.status_word(dev).zero_()
    c = torch.cat([torch.zeros(200, 1, dtype=torch.long), torch.randint(0, 6, (200, 3), generator=torch.Generator().manual_seed(4))], 1).int()
    tk, tv, _, uq, nu = oracle.map_insert(c.contiguous())
    cu = c[uq.long()].contiguous()
    nbr = oracle.nbr_build(cu, tk, tv, kernel_offsets(3, 1))          # many pairs per row
    oracle.rowlist_build(nbr)
    with pytest.raises(StatusError) as ei:
        oracle.check_status(dev)
    assert ei.value.bits == 32 and "one-pair" in str(ei.value)
    ident = torch.arange(cu.shape[0], dtype=torch.int32).reshape(1, -1).contiguous()      # a one-pair map: silent
    oracle.rowlist_build(ident)
    oracle.check_status(dev)


def test_keep_mask_matches_the_torch_formulation(oracle):
    """`keep_mask` (one pass) against the element-wise torch formulation of decoder_v3.py:148-158, 411-420: OR over the
    sources, the "nothing kept -> first rows" fallback, the inclusive box test; both source kinds."""
    g = torch.Generator().manual_seed(3)
    n = 5000
    coords = torch.randint(-5, 60, (n, 4), generator=g, dtype=torch.int32)
    lo, hi = torch.tensor([0, 3, -2], dtype=torch.int32), torch.tensor([40, 50, 30], dtype=torch.int32)
    inside = ((coords[:, 1:] >= lo) & (coords[:, 1:] <= hi)).all(dim=1)
    rows = [torch.randint(-3, 2, (n,), generator=g, dtype=torch.int32) for _ in range(3)]
    masks = [r >= 0 for r in rows]
    want_or = masks[0] | masks[1] | masks[2]
    assert torch.equal(oracle.keep_mask(rows), want_or)
    assert torch.equal(oracle.keep_mask([m.contiguous() for m in masks]), want_or)
    assert torch.equal(oracle.keep_mask([rows[0]], coords, lo, hi, fallback_rows=1000), masks[0] & inside)
    none = torch.full((n,), -1, dtype=torch.int32)
    first = torch.arange(n) < 1000
    assert torch.equal(oracle.keep_mask([none], coords, lo, hi, fallback_rows=1000), first & inside)
    assert torch.equal(oracle.keep_mask([none], coords, lo, hi), torch.zeros(n, dtype=torch.bool))
    assert oracle.keep_mask([torch.zeros(0, dtype=torch.int32)]).shape == (0,)
