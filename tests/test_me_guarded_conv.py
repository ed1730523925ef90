"""The plain `pasco_amd.me` convolution modules (the literal drop-in route: reference code on `import pasco_amd.me as ME`)
run their products on the split-precision kernels GUARDED by a device-side predicate: the operand split reports an f16 range
overflow into a word of the call, the exact fp32 kernel is launched behind the split one with that word as its predicate
(`ph_conv_desc.exact_if`) - exact results whenever the range is left, no host read, nothing for the caller to check."""
import pytest
import torch

import pasco_amd.me as ME
from pasco_amd.me import modules as M


def _scene(n=4000, extent=(24, 24, 10), c=64, seed=0, big=None, gain=1.0):
    g = torch.Generator().manual_seed(seed)
    X, Y, Z = extent
    site = torch.randperm(X * Y * Z, generator=g)[:n]
    coords = torch.stack([torch.zeros_like(site), site // (Y * Z), (site // Z) % Y, site % Z], 1).int()
    feats = torch.randn(n, c, generator=g) * gain
    if big is not None:
        feats[17, 3] = big                       # one activation beyond the f16 range of the scaled operand (|x| > 2047)
    return coords, feats


def _run(device, big, mode, n=4000, extent=(24, 24, 10), gain=1.0):
    torch.manual_seed(3)
    conv = ME.MinkowskiConvolution(64, 64, kernel_size=3, bias=True, dimension=3).to(device).eval()
    coords, feats = _scene(n=n, extent=extent, big=big, gain=gain)
    x = ME.SparseTensor(feats.to(device), coords.to(device))
    M.set_me_conv(mode)
    try:
        with torch.no_grad():
            return conv(x).F.cpu()
    finally:
        M.set_me_conv("guarded")


def _check(device, be, **kw):
    exact = _run(device, None, "exact", **kw)
    guarded = _run(device, None, "guarded", **kw)
    scale = float(exact.abs().mean())
    assert float((guarded - exact).abs().max()) <= 2e-5 * scale            # split precision: fp32-class, not bit-equal
    assert not torch.equal(guarded, exact) or device == "cpu"              # i.e. the split kernel did the work
    # an activation outside the range: the guard fires and the exact kernel's result is what comes back, bit for bit
    exact_big = _run(device, 5000.0, "exact", **kw)
    guarded_big = _run(device, 5000.0, "guarded", **kw)
    assert torch.isfinite(guarded_big).all() and torch.equal(guarded_big, exact_big)
    # a tensor of TINY activations only (the fixed 2^5 operand scale leaves |x| ~ 1e-6 with 2^-11-class hi / lo pairs,
    # 5e-4 relative error at 1e-6 and 6e-2 at 1e-8): no value reaches the full-precision range, the magnitude bit of the guard
    # word stays clear and the exact kernel's result comes back, bit for bit
    for gain in (1e-6, 1e-8):
        exact_tiny = _run(device, None, "exact", gain=gain, **kw)
        guarded_tiny = _run(device, None, "guarded", gain=gain, **kw)
        assert torch.equal(guarded_tiny, exact_tiny), gain
    # small but not tiny (|x| ~ 1e-2: values above 2^-9 exist): the split kernel keeps the work, at fp32-class accuracy
    exact_small = _run(device, None, "exact", gain=1e-2, **kw)
    guarded_small = _run(device, None, "guarded", gain=1e-2, **kw)
    bias_free = float((exact_small - exact_small.mean(0)).abs().mean())
    assert float((guarded_small - exact_small).abs().max()) <= 1e-4 * bias_free
    be.check_status(torch.device(device))                                  # and the stream's own status pair stayed clean


def test_guarded_module_conv_on_the_checker(oracle_registered):
    oracle_registered.checker_split = True
    try:
        _check("cpu", oracle_registered)
    finally:
        oracle_registered.checker_split = False


def test_exact_if_guards_an_exact_launch(oracle):
    coords, feats = _scene(n=500, c=16)
    tk, tv, _, uq, nu = oracle.map_insert(coords.contiguous())
    from pasco_amd.me.core import kernel_offsets
    nbr = oracle.nbr_build(coords, tk, tv, kernel_offsets(3, 1))
    w = torch.randn(27, 16, 8, generator=torch.Generator().manual_seed(1))
    out = torch.full((500, 8), 7.0)
    from pasco_amd.me.backend import STATUS_MAGNITUDE
    flag = torch.full((1,), STATUS_MAGNITUDE, dtype=torch.int32)      # what a healthy operand split leaves: no overflow, magnitude present
    oracle.conv_fwd(feats, w, nbr, 500, out=out, exact_if=flag)
    assert bool((out == 7.0).all()), "no overflow, magnitude present: the guarded launch must not touch the output"
    for word in (STATUS_MAGNITUDE | 1, 0):                             # an overflow / a tensor of tiny values only: the exact kernel works
        out.fill_(7.0)
        flag.fill_(word)
        oracle.conv_fwd(feats, w, nbr, 500, out=out, exact_if=flag)
        assert torch.equal(out, oracle.conv_fwd(feats, w, nbr, 500)), word
    with pytest.raises(ValueError):
        oracle.conv_fwd(feats, w, nbr, 500, exact_if=flag, split=(w, 1.0))


@pytest.mark.gpu
def test_guarded_module_conv_on_the_gpu(hip):
    _check("cuda", hip)
    _check("cuda", hip, n=60000, extent=(96, 96, 16))       # >= 16 384 rows: window tables, the k_conv_wop2 / gather pair


@pytest.mark.gpu
def test_exact_if_guards_an_exact_launch_gpu(hip):
    coords, feats = _scene(n=20000, c=64, extent=(48, 48, 16))
    x = ME.SparseTensor(feats.cuda(), coords.cuda())
    mgr = x.coordinate_manager
    nbr = mgr.kernel_map(x.coordinate_map_key, x.coordinate_map_key, 3)
    w = torch.randn(27, 64, 64, generator=torch.Generator().manual_seed(1)).cuda() * 0.05
    out = torch.full((20000, 64), 7.0, device="cuda")
    from pasco_amd.me.backend import STATUS_MAGNITUDE
    flag = torch.full((1,), STATUS_MAGNITUDE, dtype=torch.int32, device="cuda")
    hip.conv_fwd(x.F, w, nbr, 20000, out=out, exact_if=flag)
    assert bool((out == 7.0).all())
    for word in (STATUS_MAGNITUDE | 1, 0):
        out.fill_(7.0)
        flag.fill_(word)
        hip.conv_fwd(x.F, w, nbr, 20000, out=out, exact_if=flag)
        assert torch.equal(out, hip.conv_fwd(x.F, w, nbr, 20000)), word
    # ph_split_rows raises the magnitude bit exactly when a value reaches the full-precision range of the scaled operand
    for gain, want in ((1.0, STATUS_MAGNITUDE), (1e-5, 0)):
        word = torch.zeros(1, dtype=torch.int32, device="cuda")
        hip.split_rows((feats * gain).cuda(), status=word)
        assert int(word) == want, (gain, int(word))


# ---- deferred BatchNorm / activation on the plain modules -----------------------------------------------------------------
def _chain(device, defer, act="relu"):
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(5)
    bn = ME.MinkowskiBatchNorm(16).eval()
    bn.bn.running_mean.copy_(torch.randn(16, generator=g) * 0.3)
    bn.bn.running_var.copy_(torch.rand(16, generator=g) + 0.5)
    bn.bn.weight.data.copy_(torch.rand(16, generator=g) + 0.5)
    bn.bn.bias.data.copy_(torch.randn(16, generator=g) * 0.1)
    relu = ME.MinkowskiReLU() if act == "relu" else ME.MinkowskiLeakyReLU(0.2)
    conv = ME.MinkowskiConvolution(16, 8, kernel_size=3, bias=True, dimension=3).eval()
    bn, conv = bn.to(device), conv.to(device)
    coords, feats = _scene(n=700, c=16, extent=(12, 12, 8), seed=2)
    x = ME.SparseTensor(feats.to(device), coords.to(device))
    old = M._ME_DEFER
    M._ME_DEFER = defer
    try:
        with torch.no_grad():
            h = relu(bn(x))
            pending = h._pending
            y = conv(h)
            # anything else that reads the values sees the computed tensor
            hF = h.F.clone()
            s = (h + h).F
            p = ME.MinkowskiPruning()(h, feats[:, 0].to(device) > 0).F
        return pending, hF.cpu(), y.F.cpu(), s.cpu(), p.cpu()
    finally:
        M._ME_DEFER = old


@pytest.mark.parametrize("act", ["relu", "leaky"])
def test_deferred_batchnorm_and_activation_change_nothing_observable(oracle_registered, act):
    pend1, h1, y1, s1, p1 = _chain("cpu", True, act)
    pend0, h0, y0, s0, p0 = _chain("cpu", False, act)
    assert pend0 is None and pend1 is not None and pend1[2] == (1 if act == "relu" else 2)      # recorded, not computed
    for a, b in ((h1, h0), (y1, y0), (s1, s0), (p1, p0)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), float((a - b).abs().max())


def test_training_mode_is_never_deferred(oracle_registered):
    bn = ME.MinkowskiBatchNorm(8).train()
    coords, feats = _scene(n=100, c=8, extent=(8, 8, 4))
    x = ME.SparseTensor(feats, coords)
    with torch.no_grad():
        assert bn(x)._pending is None
    with torch.enable_grad():
        assert ME.MinkowskiReLU()(x)._pending is None


@pytest.mark.gpu
def test_deferred_chain_on_the_gpu(hip):
    _, h1, y1, s1, p1 = _chain("cuda", True)
    _, h0, y0, s0, p0 = _chain("cuda", False)
    for a, b in ((h1, h0), (y1, y0), (s1, s0), (p1, p0)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), float((a - b).abs().max())
