"""Property tests (hypothesis) of the coordinate-map semantics on the CPU oracle: negative
coordinates, duplicates, empty inputs, strides (SURVEY.md section 4 plan item 4)."""
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
