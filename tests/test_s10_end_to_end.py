"""ONE whole step at the benchmark's size, HIP against the CPU oracle, end to end.

S10 seed 0 (256 x 256 x 32, ~210 k occupied voxels), MIMO M = 3, light decoder, teacher-forced pruning - the scene
`bench.py` times - through `PascoNet.step_inference` (= the reference's `Net.step_inference`,
net_panoptic_sparse.py:539-608: point MLP + merge, U-Net, mask transformer, `Ensembler.ensemble_sem_compl` /
`ensemble_panop` (ensembler.py:20-187), `panoptic_inference` (helper.py:91-303)), once on the MI355X through libpascohip
and once on the host through the oracle library + torch-CPU (the ~20 s `bench.py`'s `cpu_baseline` spends per scene).

The contract, stated once (DESIGN.md section 2 quotes this file):
  * coordinates of every sparse output: bit-exact, same row order;
  * every floating-point logit tensor (`sem_logits_at_scales`, `voxel_logits`, `query_logits`):
    |got - exp| <= 1e-3 (|exp| + 0.25 mean |exp|) (a logit that cancels to ~0 has no meaningful relative error of its
    own) - the same floor two HIP paths are held to against each other (tests/test_hip_bench_shapes.py).  Measured (round
    5): 7.9e-5 of mean |y| at the worst element, 1.85e-5 with the oracle's attention-mask decisions forced into the HIP
    run (asserted <= 1e-4); see FLOOR below for how the 7.2e-4 of rounds 3 - 4 turned out to be the checker's own;
  * semantic ensemble and the subnets' mask probabilities of the chain: |difference| <= 2e-3 (they live in [0, 1]);
  * the ensembling + panoptic stage (Hungarian matching of queries, merged masks, segments) is compared on IDENTICAL inputs
    - the device's stage fed the oracle's subnet predictions: rows bit-exact, probabilities to 1e-5, same segments, panoptic
    ids on <= 1e-4 of the rows different (an arg-max over queries whose two best masks tie to within rounding).  The chain's
    own ensemble output is reported only: with random-init weights the matching is decided by differences far below the
    logits' agreement.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


FLOOR = 0.25         # |got - exp| <= 1e-3 (|exp| + FLOOR * mean |exp|): the floor two HIP paths are held to against each other
# History of this number.  Rounds 3 - 4 measured 7.2e-4 of mean |y| at the worst voxel logit, held FLOOR at 1.0 and blamed the
# attention masks being thresholds of the previous prediction.  Round 5 tested that: with the oracle run's 189 M mask decisions
# FORCED into the HIP run (tests/mask_freeze.py) the error did not move, and a third arithmetic (every product on the exact fp32
# MFMA) sat 5.5e-5 from the split path but the same 7.2e-4 from the oracle - the outlier was the CHECKER: its attention summed the
# softmax denominator and the weighted values over up to 631 k keys sequentially in fp32.  With those two sums in double
# (oracle/pasco_oracle.c pho_attn_cross_*) the HIP path is 7.9e-5 of mean |y| from the oracle (3.2e-4 under this floor), and
# with the mask decisions frozen 1.85e-5: what was left WAS the thresholds (~60 of 189 M decisions land on the other side of 0).
FROZEN_MAX_OVER_MEAN = 1e-4   # with the oracle's mask decisions forced: max |error| / mean |y| (measured 1.85e-5)
PROB_ATOL = 2e-4     # probabilities ([0, 1]) of the chain: measured 2.5e-5 at most (logits agree to 8e-5 of their mean magnitude)


def _rel(a, b, floor=FLOOR):
    scale = float(b.abs().mean())
    err = float((a - b).abs().max())
    rel = float(((a - b).abs() / (b.abs() + floor * scale)).max())
    return err / max(scale, 1e-30), rel


def test_s10_step_hip_vs_oracle_end_to_end(hip, oracle):
    import bench
    from pasco_amd.graph.synth import TeacherKeep, make_scene
    from pasco_amd.me import backend

    scene_cpu = make_scene(seed=0, n_infers=3, in_channels=283)

    def run(device, unet_only=False):
        net = bench.build_net(3, 283, device)
        sc = scene_cpu.to(device)
        tk = TeacherKeep(sc, device)
        with torch.no_grad():
            x = net.prepare_input(sc.in_feats, sc.in_coords)
            ret = net(x, sc.global_min_Cs, sc.global_max_Cs, sc.min_Cs, sc.max_Cs, keep_override=tk)
            if unet_only:
                return ret
            conf, sem_probs, panop = net.ensemble(ret, sc.Ts)
            from pasco_amd.graph.panoptic import panoptic_inference
            pis = [panoptic_inference(p["voxel_probs"], p["query_probs"], overlap_threshold=net.overlap_threshold,
                                      object_mask_threshold=net.object_mask_threshold, thing_ids=net.thing_ids,
                                      scene_size=net.ensembler.scene_size,
                                      min_C=torch.zeros(3, dtype=torch.int32, device=x.device), input_query_logit=False,
                                      input_voxel_logit=False) for p in panop]
        return ret, conf, sem_probs, panop, pis, net, sc

    from pasco_amd.graph.ensemble import Ensembler
    got = run(torch.device("cuda", 0))
    torch.cuda.synchronize()
    match_orig = Ensembler.match_queries
    recorded = []                      # the oracle's query matchings: (a_idx, b_idx, matched IoUs) per auxiliary subnet

    def recording(anchor, aux, thr):
        out = match_orig(anchor, aux, thr)
        recorded.append(tuple(t.clone() for t in out))
        return out

    from tests import mask_freeze
    mask_decisions = []                # the oracle run's attention-mask decisions (mask logit > 0), per decoder layer
    backend.register_checker_backend(oracle)
    Ensembler.match_queries = staticmethod(recording)
    try:
        with mask_freeze.recording(mask_decisions):
            exp = run(torch.device("cpu"))
    finally:
        Ensembler.match_queries = staticmethod(match_orig)
        backend.register_checker_backend(None)

    worst = {"max/mean": 0.0, "elementwise": 0.0}

    failures = []

    def close(a, b, what):
        a = a.cpu()
        m, r = _rel(a, b)
        worst["max/mean"], worst["elementwise"] = max(worst["max/mean"], m), max(worst["elementwise"], r)
        if m > 1e-3 or r > 1e-3:           # say where: one column (a query / class) or scattered elements?
            scale = float(b.abs().mean())
            rel = (a - b).abs() / (b.abs() + FLOOR * scale)
            flat = rel.reshape(-1, rel.shape[-1])
            col = flat.max(dim=0)[0]
            top = torch.topk(col, min(3, col.numel()))
            frac = float((flat > 1e-3).float().mean())
            idx = int(flat.argmax())
            r_, c_ = idx // flat.shape[1], idx % flat.shape[1]
            av, bv = a.reshape(-1, rel.shape[-1])[r_, c_], b.reshape(-1, rel.shape[-1])[r_, c_]
            failures.append(f"{what}: max error {m:.3e} of mean |y|, element-wise relative {r:.3e} (floor {FLOOR} mean |y| = "
                            f"{FLOOR * scale:.3e}); {frac:.2e} of the elements beyond 1e-3; worst columns "
                            f"{[(int(i), round(float(v), 5)) for v, i in zip(top.values, top.indices)]}; worst element "
                            f"[{r_}, {c_}]: {float(av):.6f} vs {float(bv):.6f}")

    g_ret, e_ret = got[0], exp[0]
    n_rows = 0
    for s in e_ret["sem_logits_at_scales"]:
        for i, (a, b) in enumerate(zip(g_ret["sem_logits_at_scales"][s], e_ret["sem_logits_at_scales"][s])):
            assert torch.equal(a.C.cpu(), b.C), f"coordinates of the semantic logits at scale {s}, subnet {i}"
            close(a.F, b.F, f"sem logits scale {s} subnet {i}")
            n_rows += b.F.shape[0]
    for i, (a, b) in enumerate(zip(g_ret["panop_predictions"], e_ret["panop_predictions"])):
        assert torch.equal(a["voxel_logits"].C.cpu(), b["voxel_logits"].C), f"coordinates of the voxel logits, subnet {i}"
        close(a["voxel_logits"].F, b["voxel_logits"].F, f"voxel logits subnet {i}")
        close(a["query_logits"], b["query_logits"], f"query logits subnet {i}")
    print(f"[s10 e2e] logits: worst max-error / mean |y| {worst['max/mean']:.2e}, worst element-wise relative "
          f"{worst['elementwise']:.2e} over {n_rows} semantic rows + the voxel / query logits")
    for f in failures:
        print("[s10 e2e] BEYOND 1e-3:", f)
    # ---- the same step with the oracle run's attention-mask decisions forced into the HIP run -----------------------------
    # The masks of decoder layer l are thresholds (mask logit > 0, transformer_predictor_v2.py:224) of layer l - 1's
    # prediction: a near-zero logit that the two arithmetics round to different sides changes one key of one query's
    # attention.  With the decisions frozen, what remains is arithmetic alone (measured 1.85e-5 against 7.9e-5 free-running).
    stats = {}
    with mask_freeze.forcing(mask_decisions, stats):
        frozen = run(torch.device("cuda", 0), unet_only=True)
    torch.cuda.synchronize()
    fw = {"max/mean": 0.0, "elementwise": 0.0}
    pairs = [(a.F, b.F, f"sem logits scale {s} subnet {i}") for s in e_ret["sem_logits_at_scales"]
             for i, (a, b) in enumerate(zip(frozen["sem_logits_at_scales"][s], e_ret["sem_logits_at_scales"][s]))]
    for i, (a, b) in enumerate(zip(frozen["panop_predictions"], e_ret["panop_predictions"])):
        pairs += [(a["voxel_logits"].F, b["voxel_logits"].F, f"voxel logits subnet {i}"),
                  (a["query_logits"], b["query_logits"], f"query logits subnet {i}")]
    for a, b, what in pairs:
        m, r = _rel(a.cpu(), b, FLOOR)
        fw["max/mean"], fw["elementwise"] = max(fw["max/mean"], m), max(fw["elementwise"], r)
        if r > 1e-3 or m > FROZEN_MAX_OVER_MEAN:
            failures.append(f"frozen masks: {what}: max error {m:.3e} of mean |y| (bound {FROZEN_MAX_OVER_MEAN}), element-wise "
                            f"relative {r:.3e} (floor {FLOOR} mean |y|)")
    print(f"[s10 e2e] attention-mask decisions: {stats['decisions']} in {stats['calls']} layers, {stats['differ']} decided "
          f"differently by the HIP run on its own, {stats['differ_above_noise']} of them with a logit above 1e-3 of the mean")
    print(f"[s10 e2e] logits with the oracle's mask decisions forced: worst max-error / mean |y| {fw['max/mean']:.2e}, "
          f"element-wise relative (floor {FLOOR}) {fw['elementwise']:.2e}  [free-running: {worst['max/mean']:.2e}]")
    if stats["differ_above_noise"]:
        failures.append(f"{stats['differ_above_noise']} attention-mask decisions differ with a logit above the noise")
    # ---- a third arithmetic: every product on the exact fp32 MFMA, against the oracle and against the split path -----------
    from pasco_amd.graph import fused
    fused.set_conv_precision("f32")
    try:
        exact = run(torch.device("cuda", 0), unet_only=True)
    finally:
        fused.set_conv_precision("f16x3")
    torch.cuda.synchronize()
    tri = {"exact fp32 MFMA vs oracle": 0.0, "split precision vs oracle": worst["max/mean"], "split vs exact fp32 MFMA": 0.0}
    for i, (a, b, c) in enumerate(zip(exact["panop_predictions"], e_ret["panop_predictions"], g_ret["panop_predictions"])):
        assert torch.equal(a["voxel_logits"].C.cpu(), b["voxel_logits"].C)
        m, r = _rel(a["voxel_logits"].F.cpu(), b["voxel_logits"].F)
        tri["exact fp32 MFMA vs oracle"] = max(tri["exact fp32 MFMA vs oracle"], m)
        if m > 1e-3 or r > 1e-3:
            failures.append(f"exact fp32 MFMA vs oracle: voxel logits subnet {i}: {m:.3e} / {r:.3e}")
        m2, _ = _rel(c["voxel_logits"].F.cpu(), a["voxel_logits"].F.cpu())
        tri["split vs exact fp32 MFMA"] = max(tri["split vs exact fp32 MFMA"], m2)
    print("[s10 e2e] three fp32 arithmetics of the same graph, voxel logits, max error / mean |y|: " +
          ", ".join(f"{k} {v:.2e}" for k, v in tri.items()))

    # ensembled semantic probabilities + confidences (dense [C, X, Y, Z] / [X, Y, Z] per subnet and for the ensemble)
    d_sem = max(float((a.cpu() - b).abs().max()) for a, b in zip(got[2], exp[2]))
    d_conf = max(float((a.cpu() - b).abs().max()) for a, b in zip(got[1], exp[1]))
    print(f"[s10 e2e] ensembled semantic probabilities: max |difference| {d_sem:.2e}, confidences {d_conf:.2e}")
    if d_sem > PROB_ATOL or d_conf > PROB_ATOL:
        failures.append(f"ensembled semantic probabilities {d_sem:.3e} / confidences {d_conf:.3e}")
    # ---- the ensembling + panoptic stage on IDENTICAL inputs --------------------------------------------------------------
    # The panoptic ensemble matches the subnets' queries by a Hungarian assignment on soft IoUs (ensembler.py:64-110).  With
    # random-init weights many queries' masks are near-copies of each other, so the assignment is decided by differences far
    # below the 7e-4 the subnets' logits agree to: the chain's ENSEMBLE output (index M) is reported, not asserted.  What is
    # asserted: the device's ensembler + panoptic inference (k_sem_ensemble, k_ens_resample / merge / finish at ~2.1 M
    # canonical sites, graph/panoptic.py) fed the ORACLE's subnet predictions against the oracle's own ensembling of them.
    # The cost matrix is thresholded (IoU <= 0.2 -> 0, ensembler.py:64-110), so most of its entries tie and the optimal
    # assignment is not unique: the device's own assignment must reach the SAME optimum (sum of matched IoUs), and the merged
    # masks are compared under the oracle's assignment (replayed into the device's stage).
    import pasco_amd.me as ME
    n_sub = len(exp[0]["panop_predictions"])
    for i, (a, b) in enumerate(zip(got[3], exp[3])):
        if torch.equal(a["voxel_probs"].C.cpu(), b["voxel_probs"].C):
            dv = float((a["voxel_probs"].F.cpu() - b["voxel_probs"].F).abs().max())
        else:
            dv = float("nan")
        print(f"[s10 e2e] chain, output {i}{' (ensemble: matching-dependent, reported only)' if i == n_sub else ''}: mask "
              f"probabilities max |difference| {dv:.2e}")
        if i < n_sub and not dv <= PROB_ATOL:
            failures.append(f"chain: mask probabilities of subnet {i}: {dv:.3e}")
    dev = torch.device("cuda", 0)
    e_ret = exp[0]
    ret_dev = {"sem_logits_at_scales": {1: [ME.SparseTensor(t.F.to(dev), t.C.to(dev)) for t in e_ret["sem_logits_at_scales"][1]]},
               "panop_predictions": [{"voxel_logits": ME.SparseTensor(p["voxel_logits"].F.to(dev), p["voxel_logits"].C.to(dev)),
                                      "query_logits": p["query_logits"].to(dev)} for p in e_ret["panop_predictions"]]}
    net_dev, sc_dev = got[5], got[6]
    from pasco_amd.graph.panoptic import panoptic_inference
    replay = list(recorded)
    objective = []                     # (device's own optimum, oracle's) of every matching: sum of the matched IoUs

    def replaying(anchor, aux, thr):
        own = match_orig(anchor, aux, thr)
        a_idx, b_idx, iou = replay.pop(0)
        objective.append((float(own[2].sum()), float(iou.sum())))
        return a_idx.to(anchor.device), b_idx.to(anchor.device), iou

    Ensembler.match_queries = staticmethod(replaying)
    try:
        with torch.no_grad():
            conf_d, sem_d, panop_d = net_dev.ensemble(ret_dev, sc_dev.Ts)
    finally:
        Ensembler.match_queries = staticmethod(match_orig)
    for k, (own, ora) in enumerate(objective):
        print(f"[s10 e2e] stage: matching {k}: sum of matched IoUs {own:.6f} (device's assignment) vs {ora:.6f} (oracle's)")
        if abs(own - ora) > 1e-4 * max(1.0, abs(ora)):
            failures.append(f"stage: matching {k}: the device's optimum {own:.6f} differs from the oracle's {ora:.6f}")
    with torch.no_grad():
        pis_d = [panoptic_inference(p["voxel_probs"], p["query_probs"], overlap_threshold=net_dev.overlap_threshold,
                                    object_mask_threshold=net_dev.object_mask_threshold, thing_ids=net_dev.thing_ids,
                                    scene_size=net_dev.ensembler.scene_size, min_C=torch.zeros(3, dtype=torch.int32, device=dev),
                                    input_query_logit=False, input_voxel_logit=False) for p in panop_d]
    d_sem2 = max(float((a.cpu() - b).abs().max()) for a, b in zip(sem_d, exp[2]))
    print(f"[s10 e2e] stage: ensembled semantic probabilities max |difference| {d_sem2:.2e}")
    if d_sem2 > 1e-5:
        failures.append(f"stage: ensembled semantic probabilities {d_sem2:.3e}")
    for i, (a, b) in enumerate(zip(panop_d, exp[3])):
        if not torch.equal(a["voxel_probs"].C.cpu(), b["voxel_probs"].C):
            failures.append(f"stage: rows of the ensembled masks {i} differ")
            continue
        dv = float((a["voxel_probs"].F.cpu() - b["voxel_probs"].F).abs().max())
        ds = float((a["sem_probs"].F.cpu() - b["sem_probs"].F).abs().max())
        dq = float((a["query_probs"].cpu() - b["query_probs"]).abs().max())
        print(f"[s10 e2e] stage, output {i}: {a['voxel_probs'].C.shape[0]} rows identical; mask probabilities max |difference| "
              f"{dv:.2e}, sem {ds:.2e}, query {dq:.2e}")
        if max(dv, ds, dq) > 1e-5:
            failures.append(f"stage: ensembled probabilities of output {i}: masks {dv:.3e}, sem {ds:.3e}, query {dq:.3e}")
    for i, (a, b) in enumerate(zip(pis_d, exp[4])):
        info = lambda pi: [(s["id"], bool(s["isthing"]), int(s["category_id"]), int(s["query_id"])) for s in pi["segments_infos"][0]]
        if info(a) != info(b):
            failures.append(f"stage: segments of output {i}: {info(a)} vs {info(b)}")
            continue
        pa, pb = a["panoptic_seg_sparses"][0].cpu(), b["panoptic_seg_sparses"][0]
        frac = float((pa != pb).float().mean()) if pa.shape == pb.shape else 1.0
        sa, sb = a["semantic_seg_denses"][0].cpu(), b["semantic_seg_denses"][0]
        fs = float((sa != sb).float().mean())
        da = max(float((a[k][0].cpu().float() - b[k][0].float()).abs().max()) for k in
                 ("ins_uncertainty_denses", "vox_confidence_denses", "vox_uncertainty_denses"))
        print(f"[s10 e2e] stage, output {i}: {len(info(a))} segments identical; panoptic ids differ on {frac:.1e} of the rows, "
              f"semantic labels on {fs:.1e} of the grid, confidence / uncertainty maps by {da:.1e}")
        if frac > 1e-4 or fs > 1e-5 or da > 1e-3:
            failures.append(f"stage: panoptic output {i}: ids {frac:.2e}, labels {fs:.2e}, maps {da:.2e}")
    assert not failures, failures
