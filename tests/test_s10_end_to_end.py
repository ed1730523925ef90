"""ONE whole step at the benchmark's size, HIP against the CPU oracle, end to end.

S10 seed 0 (256 x 256 x 32, ~210 k occupied voxels), MIMO M = 3, light decoder, teacher-forced pruning - the scene
`bench.py` times - through `PascoNet.step_inference` (= the reference's `Net.step_inference`,
net_panoptic_sparse.py:539-608: point MLP + merge, U-Net, mask transformer, `Ensembler.ensemble_sem_compl` /
`ensemble_panop` (ensembler.py:20-187), `panoptic_inference` (helper.py:91-303)), once on the MI355X through libpascohip
and once on the host through the oracle library + torch-CPU (the ~20 s `bench.py`'s `cpu_baseline` spends per scene).

The contract, stated once (DESIGN.md section 2 quotes this file):
  * coordinates of every sparse output: bit-exact, same row order;
  * every floating-point logit tensor (`sem_logits_at_scales`, `voxel_logits`, `query_logits`): element-wise relative
    error <= 1e-3 with an absolute floor of 0.25 mean |y| under the denominator (a logit that cancels to ~0 has no
    meaningful relative error; what the floor is sized against: tests/test_hip_bench_shapes.py), and max error <= 1e-3 of
    mean |y|;
  * ensembled probabilities: |difference| <= 1e-3 (they live in [0, 1]);
  * panoptic segments: same segments (id, thing / stuff, class, query); voxels whose panoptic id differs <= 1e-4 of the rows
    (an arg-max over queries whose two best masks tie to within rounding may fall either way).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    scale = float(b.abs().mean())
    err = float((a - b).abs().max())
    rel = float(((a - b).abs() / (b.abs() + 0.25 * scale)).max())
    return err / max(scale, 1e-30), rel


def test_s10_step_hip_vs_oracle_end_to_end(hip, oracle):
    import bench
    from pasco_amd.graph.synth import TeacherKeep, make_scene
    from pasco_amd.me import backend

    scene_cpu = make_scene(seed=0, n_infers=3, in_channels=283)

    def run(device):
        net = bench.build_net(3, 283, device)
        sc = scene_cpu.to(device)
        tk = TeacherKeep(sc, device)
        with torch.no_grad():
            x = net.prepare_input(sc.in_feats, sc.in_coords)
            ret = net(x, sc.global_min_Cs, sc.global_max_Cs, sc.min_Cs, sc.max_Cs, keep_override=tk)
            conf, sem_probs, panop = net.ensemble(ret, sc.Ts)
            from pasco_amd.graph.panoptic import panoptic_inference
            pis = [panoptic_inference(p["voxel_probs"], p["query_probs"], overlap_threshold=net.overlap_threshold,
                                      object_mask_threshold=net.object_mask_threshold, thing_ids=net.thing_ids,
                                      scene_size=net.ensembler.scene_size,
                                      min_C=torch.zeros(3, dtype=torch.int32, device=x.device), input_query_logit=False,
                                      input_voxel_logit=False) for p in panop]
        return ret, conf, sem_probs, panop, pis

    got = run(torch.device("cuda", 0))
    torch.cuda.synchronize()
    backend.register_checker_backend(oracle)
    try:
        exp = run(torch.device("cpu"))
    finally:
        backend.register_checker_backend(None)

    worst = {"max/mean": 0.0, "elementwise": 0.0}

    def close(a, b, what):
        m, r = _rel(a.cpu(), b)
        worst["max/mean"], worst["elementwise"] = max(worst["max/mean"], m), max(worst["elementwise"], r)
        assert m <= 1e-3, f"{what}: max error {m:.3e} of mean |y|"
        assert r <= 1e-3, f"{what}: element-wise relative error {r:.3e} (floor 0.25 mean |y|)"

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

    # ensembled semantic probabilities + confidences (dense [C, X, Y, Z] / [X, Y, Z] per subnet and for the ensemble)
    for i, (a, b) in enumerate(zip(got[2], exp[2])):
        d = float((a.cpu() - b).abs().max())
        assert d <= 1e-3, f"ensembled semantic probabilities {i}: {d:.3e}"
    for i, (a, b) in enumerate(zip(got[1], exp[1])):
        assert float((a.cpu() - b).abs().max()) <= 1e-3, f"semantic confidence {i}"

    # panoptic ensembling: same union rows, probabilities to 1e-3
    flips = 0.0
    for i, (a, b) in enumerate(zip(got[3], exp[3])):
        assert torch.equal(a["voxel_probs"].C.cpu(), b["voxel_probs"].C), f"rows of the ensembled masks {i}"
        assert float((a["voxel_probs"].F.cpu() - b["voxel_probs"].F).abs().max()) <= 1e-3, f"ensembled mask probabilities {i}"
        assert float((a["sem_probs"].F.cpu() - b["sem_probs"].F).abs().max()) <= 1e-3, f"ensembled sem probabilities {i}"
        assert float((a["query_probs"].cpu() - b["query_probs"]).abs().max()) <= 1e-3, f"ensembled query probabilities {i}"
    # panoptic inference: segments and per-voxel ids
    for i, (a, b) in enumerate(zip(got[4], exp[4])):
        info = lambda pi: [(s["id"], bool(s["isthing"]), int(s["category_id"]), int(s["query_id"])) for s in pi["segments_infos"][0]]
        assert info(a) == info(b), f"segments of output {i}: {info(a)} vs {info(b)}"
        ca = torch.tensor([s["confidence"] for s in a["segments_infos"][0]])
        cb = torch.tensor([s["confidence"] for s in b["segments_infos"][0]])
        assert torch.allclose(ca, cb, rtol=1e-3, atol=1e-4)
        pa, pb = a["panoptic_seg_sparses"][0].cpu(), b["panoptic_seg_sparses"][0]
        frac = float((pa != pb).float().mean())
        flips = max(flips, frac)
        assert frac <= 1e-4, f"panoptic ids of output {i}: {frac:.2e} of the rows differ"
        sa, sb = a["semantic_seg_denses"][0].cpu(), b["semantic_seg_denses"][0]
        assert float((sa != sb).float().mean()) <= 1e-5
    print(f"[s10 e2e] panoptic ids: at most {flips:.1e} of the rows differ; segments identical")
