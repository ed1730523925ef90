"""Host-side plumbing of the split-operand path (mma_mode 2: ph_split_rows operands, operands emitted by the
producing launch, tensors that exist only as an operand, linear layers on the convolution kernel) exercised on
the CPU tier: the checker library is allowed to take those descriptors (it reads the operands back as hi + lo,
following the device's data flow) and the graph must give the same results as its plain fp32 path."""
import pytest
import torch

from pasco_amd.graph import PascoNet, fused
from pasco_amd.graph.synth import TeacherKeep, make_scene
from pasco_amd.me import backend


@pytest.fixture()
def split_checker(oracle):
    backend.register_checker_backend(oracle)
    oracle.status_word(torch.device("cpu")).zero_()      # other tests raise the range flag on purpose
    old = fused.MIN_ROWS_LINEAR
    try:
        yield oracle
    finally:
        oracle.checker_split = False
        fused.MIN_ROWS_LINEAR = old
        backend.register_checker_backend(None)


def run(net, scene):
    tk = TeacherKeep(scene, torch.device("cpu"))
    with torch.no_grad():
        x = net.prepare_input(scene.in_feats, scene.in_coords)
        ret = net(x, scene.global_min_Cs, scene.global_max_Cs, scene.min_Cs, scene.max_Cs, keep_override=tk)
        return ret, net.ensemble(ret, scene.Ts)


@pytest.mark.parametrize("heavy,n_infers,queries", [(False, 2, 12), (True, 2, 10)])
def test_split_operand_graph_equals_fp32_graph(split_checker, heavy, n_infers, queries):
    torch.manual_seed(3)
    net = PascoNet(n_classes=20, n_infers=n_infers, in_channels=16, f=32, num_queries=queries,
                   heavy_decoder=heavy).eval()
    scene = make_scene(4, n_infers=n_infers, in_channels=16, grid=(32, 32, 8), occupancy=0.15)
    ref, ref_ens = run(net, scene)                       # plain fp32 descriptors
    split_checker.checker_split = True
    fused.MIN_ROWS_LINEAR = 1                            # small scene: still route the linears through the kernel
    calls = {"split_only": 0, "emit": 0, "no_x": 0}
    inner = split_checker.conv_fwd

    def spy(x, weight, nbr, n_out, **kw):
        calls["emit"] += kw.get("emit_split") is not None
        calls["split_only"] += kw.get("want_out", True) is False
        calls["no_x"] += x is None
        return inner(x, weight, nbr, n_out, **kw)

    split_checker.conv_fwd = spy
    try:
        got, got_ens = run(net, scene)
    finally:
        del split_checker.conv_fwd
    assert calls["emit"] > 10 and calls["split_only"] > 5 and calls["no_x"] > 5, calls   # the path was really taken
    for s in ref["sem_logits_at_scales"]:
        for a, b in zip(got["sem_logits_at_scales"][s], ref["sem_logits_at_scales"][s]):
            assert torch.equal(a.C, b.C)
            assert torch.allclose(a.F, b.F, rtol=2e-4, atol=2e-4), float((a.F - b.F).abs().max())
    for a, b in zip(got["panop_predictions"], ref["panop_predictions"]):
        assert torch.equal(a["voxel_logits"].C, b["voxel_logits"].C)
        assert torch.allclose(a["voxel_logits"].F, b["voxel_logits"].F, rtol=1e-3, atol=1e-3)
        assert torch.allclose(a["query_logits"], b["query_logits"], rtol=1e-3, atol=1e-3)
    for a, b in zip(got_ens[1], ref_ens[1]):             # ensembled semantic probabilities
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


def test_launch_checker_formulas_on_the_oracle(split_checker):
    """tests/launch_checker.py (the in-place fp64 / oracle check the GPU suite wraps around every benchmark
    launch) run against the oracle itself on a small split-operand graph: its restatement of prologue, epilogue,
    residual tail, operand emission and operand-only inputs must agree with the library for every launch."""
    from tests.launch_checker import LaunchChecker
    torch.manual_seed(5)
    net = PascoNet(n_classes=20, n_infers=2, in_channels=16, f=32, num_queries=10, heavy_decoder=False).eval()
    scene = make_scene(6, n_infers=2, in_channels=16, grid=(32, 32, 8), occupancy=0.15)
    split_checker.checker_split = True
    fused.MIN_ROWS_LINEAR = 1
    chk = LaunchChecker(split_checker, split_checker)
    chk.install()
    try:
        run(net, scene)
    finally:
        del split_checker.conv_fwd
    launches = sum(v[0] for v in chk.seen.values())
    assert launches > 60 and all(v[0] == v[1] for v in chk.seen.values())


def test_gathered_split_equals_split_of_gathered_rows(split_checker):
    """`fused.gathered_split`: the operand of a tensor is made once and its ROWS gathered - bit for bit the operand of the
    gathered fp32 rows (splitting is row-wise), with -1 rows giving zero rows, odd channel counts padded to a group."""
    import pasco_amd.me as ME
    split_checker.checker_split = True
    g = torch.Generator().manual_seed(5)
    for c in (32, 40, 64):
        n = 301
        coords = torch.cat([torch.zeros(n, 1, dtype=torch.int32), torch.arange(n, dtype=torch.int32)[:, None].repeat(1, 3)], dim=1)
        mgr = ME.CoordinateManager(D=3, device=torch.device("cpu"))
        key = mgr.insert_unique(coords.contiguous(), 1)
        x = ME.SparseTensor(torch.randn(n, c, generator=g), coordinate_map_key=key, coordinate_manager=mgr)
        rows = torch.randperm(n, generator=g)[:97].int().contiguous()
        got = fused.gathered_split(x, rows, key)
        assert got is not None and got.channels == c
        want = split_checker.split_rows(split_checker.gather_rows(x.F.contiguous(), rows))
        assert torch.equal(got.split.view(torch.int16), want.view(torch.int16))
        again = fused.gathered_split(x, rows, key)          # the level's operand is cached on the tensor
        assert len(x.__dict__["_ph_in_split"]) == 1 and torch.equal(again.split.view(torch.int16), want.view(torch.int16))
    with fused.precision_override("f32"):
        assert fused.gathered_split(x, rows, key) is None
