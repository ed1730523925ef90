"""Panoptic post-processing on the device (include/pasco_hip.h panop_*, csrc/panop.hip; reference helper.py:91-303) against
the torch formulation of the same function (`_panoptic_inference_torch`, itself pinned to the REFERENCE's output by
tests/golden/ensemble.npz in test_golden_ensemble.py): segment tables and every integer output identical, confidence maps to
1e-6 (the sums over the kept queries run in another order).  CPU legs: the oracle's C restatement; `-m gpu`: libpascohip.so."""
import pytest
import torch

import pasco_amd.me as ME
from pasco_amd.graph import panoptic as P

THINGS = [1, 2, 3, 4, 5, 6, 7, 8]


def make_case(seed, n, q, n_classes=20, quantize=False, keep_none=False, extent=(40, 36, 12)):
    g = torch.Generator().manual_seed(seed)
    X, Y, Z = extent
    n = min(n, X * Y * Z)
    site = torch.randperm(X * Y * Z, generator=g)[:n]
    coords = torch.stack([torch.zeros_like(site), site // (Y * Z), (site // Z) % Y, site % Z], 1).int()
    masks = torch.rand(n, q, generator=g) ** 3                        # most entries small, some above the 0.3 threshold
    own = torch.randint(0, q, (n,), generator=g)
    masks[torch.arange(n), own] = 0.5 + 0.5 * torch.rand(n, generator=g)
    if quantize:                                                      # exact ties between queries
        masks = (masks * 4).round() / 4
    ql = torch.randn(1, q, n_classes + 1, generator=g)
    cls = torch.randint(0, n_classes + 1, (q,), generator=g)          # incl. class 0 and the dustbin
    cls[: q // 3] = torch.tensor([3, 12, 12, 5, 15, 15, 15, 9] * 16)[: q // 3]   # things, repeated stuff classes
    ql[0, torch.arange(q), cls] += 6.0
    if quantize:
        ql[0, 1] = ql[0, 0]                                           # two queries with identical probabilities
    qp = torch.softmax(ql, -1)
    if keep_none:
        qp = torch.full_like(qp, 1.0 / (n_classes + 1))
    return coords, masks.contiguous(), qp, extent


def run_both(device, coords, masks, qp, extent):
    v = ME.SparseTensor(masks.to(device), coords.to(device))
    kw = dict(overlap_threshold=0.4, object_mask_threshold=0.7, thing_ids=THINGS, scene_size=extent,
              min_C=torch.zeros(3, dtype=torch.int32), input_query_logit=False, input_voxel_logit=False)
    got = P.panoptic_inference(v, qp.to(device), **kw)
    assert isinstance(got, P.PanopticResult), "the device path did not serve the call"
    exp = P._panoptic_inference_torch(v, qp.to(device), **kw)
    return got, exp


def compare(got, exp):
    # the dense maps are built on first access, but the result behaves like the reference's dict before that too
    assert "semantic_seg_denses" in got and "vox_all_mask_probs_denses" in got and "nonsense" not in got
    assert set(exp.keys()) <= set(got.keys()) and got.get("nonsense", 5) == 5
    info = lambda r: [(s["id"], s["isthing"], s["category_id"], s["query_id"]) for s in r["segments_infos"][0]]
    assert info(got) == info(exp)
    for a, b in zip(got["segments_infos"][0], exp["segments_infos"][0]):
        assert abs(a["confidence"] - b["confidence"]) < 1e-7 and torch.equal(a["all_class_probs"].cpu(), b["all_class_probs"].cpu())
    assert torch.equal(got["panoptic_seg_sparses"][0].cpu(), exp["panoptic_seg_sparses"][0].cpu())
    for k in ("panoptic_seg_denses", "semantic_seg_denses"):
        assert torch.equal(got[k].cpu(), exp[k].cpu()), k
    assert torch.equal(got["ins_uncertainty_denses"].cpu(), exp["ins_uncertainty_denses"].cpu())
    for k in ("vox_confidence_denses", "vox_uncertainty_denses"):
        a, b = got[k].cpu(), exp[k].cpu()
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7, equal_nan=True), (k, float((a - b).abs().max()))
    a, b = got["vox_all_mask_probs_denses"][0].cpu(), exp["vox_all_mask_probs_denses"][0].cpu()
    assert a.shape == b.shape and torch.equal(a, b)
    assert got.get("semantic_seg_denses") is got["semantic_seg_denses"] and len(got) == len(got.keys()) == len(list(got.items()))


CASES = [dict(seed=1, n=5000, q=100), dict(seed=2, n=3000, q=100, quantize=True), dict(seed=3, n=700, q=6),
         dict(seed=4, n=2000, q=128), dict(seed=5, n=900, q=64, keep_none=True), dict(seed=6, n=1, q=100),
         dict(seed=7, n=4097, q=65, quantize=True), dict(seed=8, n=300, q=1)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"n{c['n']}_q{c['q']}" + ("_ties" if c.get("quantize") else ""))
def test_panoptic_rows_oracle_vs_torch(oracle_registered, case):
    got, exp = run_both("cpu", *make_case(**case))
    compare(got, exp)
    if case.get("keep_none"):
        assert got["segments_infos"][0] == [] and int(got["panoptic_seg_sparses"][0].abs().sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES + [dict(seed=9, n=210000, q=100, extent=(256, 256, 32))],
                         ids=lambda c: f"n{c['n']}_q{c['q']}" + ("_ties" if c.get("quantize") else ""))
def test_panoptic_rows_hip_vs_torch(hip, case):
    got, exp = run_both("cuda", *make_case(**case))
    compare(got, exp)


def test_many_outputs_share_one_copy(oracle_registered):
    """`panoptic_inference_many`: every output of a step in one call, the same results as one call each."""
    cases = [make_case(seed=20 + i, n=1500, q=100) for i in range(3)]
    kw = dict(overlap_threshold=0.4, object_mask_threshold=0.7, thing_ids=THINGS, scene_size=cases[0][3],
              min_C=torch.zeros(3, dtype=torch.int32), input_query_logit=False, input_voxel_logit=False)
    pairs = [(ME.SparseTensor(m, c), qp) for c, m, qp, _ in cases]
    many = P.panoptic_inference_many(pairs, **kw)
    for (v, qp), res in zip(pairs, many):
        compare(res, P._panoptic_inference_torch(v, qp, **kw))


def test_output_without_queries_is_served(oracle_registered):
    """An ensemble whose queries were all filtered out by the matching (ensembler.py:100-110) has [N, 0] masks: the call
    must still answer (no segments, zero labels), next to a normal output in the same step."""
    c, m, qp, ext = make_case(seed=40, n=500, q=10)
    kw = dict(overlap_threshold=0.4, object_mask_threshold=0.7, thing_ids=THINGS, scene_size=ext,
              min_C=torch.zeros(3, dtype=torch.int32), input_query_logit=False, input_voxel_logit=False)
    empty = (ME.SparseTensor(m[:, :0].contiguous(), c), qp[:, :0])
    res = P.panoptic_inference_many([(ME.SparseTensor(m, c), qp), empty], **kw)
    assert isinstance(res[0], P.PanopticResult) and res[1]["segments_infos"] == [[]]
    assert int(res[1]["panoptic_seg_sparses"][0].abs().sum()) == 0 and res[1]["panoptic_seg_denses"].shape[1:] == ext


def test_without_a_backend_the_torch_form_serves():
    c, m, qp, ext = make_case(seed=30, n=400, q=10)
    with pytest.raises(RuntimeError):
        from pasco_amd.me.backend import backend_for
        backend_for(torch.device("cpu"))
    # a SparseTensor cannot even be made on the CPU without a backend: the fallback is exercised through its own entry
    assert P._device_backend(m) is None
