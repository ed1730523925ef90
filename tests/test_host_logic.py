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


def test_weight_fragments_are_the_documented_permutation():
    """`ph_conv_desc.w_frag` (include/pasco_hip.h): lane l31 + 32 h of fragment (k, c, j, hi / lo) holds channels 16 c + 8 h .. + 7 of
    column min(32 j + l31, cout - 1) of `w_split` - checked entry by entry on a kernel with a ragged column count, and cached on
    the operand tensor."""
    import random
    import torch
    from pasco_amd.me.backend import CBackend
    K, cout, cpad = 27, 40, 64
    w = torch.randn(K * cout, cpad // 32, 2, 32).half()
    f = CBackend.weight_fragments(w, K, cout, cpad)
    assert f.shape == (K, cpad // 16, 2, 2, 64, 8) and f.is_contiguous()
    assert CBackend.weight_fragments(w, K, cout, cpad) is f                  # cached on the operand
    ws = w.view(K, cout, cpad // 32, 2, 32)
    rnd = random.Random(0)
    for _ in range(3000):
        k, c, j, part = rnd.randrange(K), rnd.randrange(cpad // 16), rnd.randrange(2), rnd.randrange(2)
        l31, h, q = rnd.randrange(32), rnd.randrange(2), rnd.randrange(8)
        n = min(32 * j + l31, cout - 1)
        assert f[k, c, j, part, l31 + 32 * h, q] == ws[k, n, c >> 1, part, (c & 1) * 16 + 8 * h + q]


def test_conv_route_is_per_thread_and_per_call(oracle):
    """`CBackend.routing(bits)` / `set_route`: ph_conv_desc.route of THIS thread's launches only (no process-global state)."""
    import threading
    seen = {}

    def other():
        seen["other"] = getattr(oracle._tls, "route", 0)
    with oracle.routing(0x5):
        assert oracle._tls.route == 0x5
        t = threading.Thread(target=other)
        t.start()
        t.join()
    assert seen["other"] == 0 and getattr(oracle._tls, "route", 0) == 0
    assert oracle.set_route(0x2) == 0 and oracle.set_route(0) == 0x2


def test_eval_modules_warn_once_when_autograd_is_on(oracle_registered):
    import warnings
    import torch
    import pasco_amd.me as ME
    from pasco_amd.me import modules as M
    M._GRAD_MODE_WARNED = False
    bn = ME.MinkowskiBatchNorm(8).eval()
    x = ME.SparseTensor(torch.randn(20, 8), torch.cat([torch.zeros(20, 1, dtype=torch.int32), torch.arange(60, dtype=torch.int32).view(20, 3)], 1))
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        with torch.enable_grad():
            y = bn(x)
            bn(x)
    assert y._pending is None                                   # the module-by-module route
    assert sum("autograd enabled" in str(r.message) for r in rec) == 1
    with torch.no_grad():
        assert bn(x)._pending is not None                        # the deferred route
