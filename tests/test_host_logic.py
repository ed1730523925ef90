"""Host-side restatements used on the hot path: each must equal the torch expression it replaces."""
import torch

from pasco_amd.graph.ensemble import _gram
from pasco_amd.graph.unet import unique_rows_sorted


def test_unique_rows_sorted_equals_torch_unique_dim():
    g = torch.Generator().manual_seed(0)
    for k in (3, 4):
        for n in (1, 7, 5000):
            r = torch.randint(-60, 300, (n, k), generator=g)
            if k == 4:
                r[:, 0] = torch.randint(0, 3, (n,), generator=g)
            r[n // 2:] = r[: n - n // 2].clone()  # duplicates
            u0, i0 = torch.unique(r, return_inverse=True, dim=0)
            u1, i1 = unique_rows_sorted(r)
            assert torch.equal(u0, u1) and torch.equal(i0, i1)
            assert u1.dtype == r.dtype


def test_gram_matches_matmul_cpu():
    g = torch.Generator().manual_seed(1)
    a, b = torch.rand(1000, 7, generator=g), torch.rand(1000, 7, generator=g)
    assert torch.allclose(_gram(a, b), a.t() @ b, rtol=1e-6, atol=1e-6)
