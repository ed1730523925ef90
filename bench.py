#!/usr/bin/env python
"""bench.py - scenes/sec of PaSCo MIMO-3 inference on the synthetic 256x256x32 (~10 % occupancy)
scene S10 (BASELINE.json metric / SURVEY.md 8(d)).

    python bench.py --gpus N --steps K --warmup W

One "step" = one scene through the hot path: point-feature MLP + voxel max (CylinderFeat), MIMO
input merge, sparse U-Net encoder / dense bottleneck / generative decoder, and the mask-transformer
decoder (the reference's `Net.step_inference` up to and including `self.unet3d(...)`,
net_panoptic_sparse.py:548-550,233-245).  Inputs are resident in HBM before the timed region.
Weights are seeded random (no checkpoints offline), BN statistics randomised, light decoder,
teacher-forced pruning.  N > 1: one process per GPU (torchrun), every rank runs its
own scenes (weak scaling, no data-path collective); time = max over ranks.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel =
k=3 sparse convolution, measured live with HIP events on the launch stream) and `cpu_baseline`
(the CPU oracle running the same graph on the host cores - the ONLY place the oracle is used here).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 (matrix) = vector rate
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense BF16/FP16 MFMA peak
HBM_PEAK_GBS = 8000.0


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/r1_pmc_conv.json, produced by tools/pmc_summary.py; FETCH_SIZE doubled per
    MI355X_MICROARCH.md).  PMC counters cannot be collected from inside the process, so this is a
    recorded measurement, or null when none is committed."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_conv.json")
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def roofline_object(per_kernel, steps):
    """`roofline` for the dominant kernel (most time in the timed region), the other conv kernel beside it."""
    def entry(name, s):
        avg = s["time_s"] / s["launches"]
        tf = s["flops"] / s["time_s"] / 1e12
        gbs = s["bytes_alg"] / s["time_s"] / 1e9
        e = {"kernel": name, "launches_per_step": s["launches"] / steps, "avg_launch_us": round(avg * 1e6, 2),
             "ms_per_step": round(s["time_s"] / steps * 1e3, 3), "flops_per_launch": s["flops"] / s["launches"],
             "alg_bytes_per_launch": s["bytes_alg"] / s["launches"], "useful_TFLOPs": round(tf, 3),
             "alg_GBps": round(gbs, 1), "traffic": pmc_traffic(name)}
        if name == "k_conv_mfma":    # exact fp32 MFMA: matrix-pipe bound (SURVEY.md section 7 roofline check)
            e.update(bound="mfma", achieved=round(tf, 3), peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                     frac=round(tf / F32_MFMA_PEAK_TFLOPS, 4), alg_frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 4))
        else:                        # split-precision products: the matrix pipe is ~1/5 busy, the gather side bounds
            e.update(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=round(gbs / HBM_PEAK_GBS, 4),
                     mfma_frac_of_f16_peak=round(3.0 * tf / F16_MFMA_PEAK_TFLOPS, 4))
        if s["k3_launches"]:
            e["k3_only"] = {"launches_per_step": s["k3_launches"] / steps,
                            "TFLOPs": round(s["k3_flops"] / max(s["k3_time_s"], 1e-12) / 1e12, 3),
                            "ms_per_step": round(s["k3_time_s"] / steps * 1e3, 3)}
        return e

    names = sorted((n for n in per_kernel if n.startswith("k_conv")), key=lambda n: -per_kernel[n]["time_s"])
    out = entry(names[0], per_kernel[names[0]])
    out["conv_ms_per_step"] = round(sum(per_kernel[n]["time_s"] for n in names) / steps * 1e3, 3)
    if len(names) > 1:
        out["other_conv_kernel"] = entry(names[1], per_kernel[names[1]])
    sp = per_kernel.get("k_split_rows")
    if sp:     # operand preparation of the split kernel (one pass per conv input, not per gather)
        out["operand_split"] = {"kernel": "k_split_rows", "launches_per_step": sp["launches"] / steps,
                                "ms_per_step": round(sp["time_s"] / steps * 1e3, 3),
                                "alg_GBps": round(sp["bytes_alg"] / sp["time_s"] / 1e9, 1),
                                "traffic": pmc_traffic("k_split_rows")}
    return out


def build_net(n_infers, in_channels, device, heavy=False):
    from pasco_amd.graph import PascoNet
    torch.manual_seed(1234)
    net = PascoNet(n_classes=20, n_infers=n_infers, in_channels=in_channels, f=64, num_queries=100,
                   heavy_decoder=heavy)
    g = torch.Generator().manual_seed(4321)
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) * 0.4 + 0.8)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
    return net.eval().to(device)


def run_scene(net, scene, teacher, window=None):
    """One step = the reference's `Net.forward(return_ensemble=True)` preceded by its input stage:
    point MLP + merge, U-Net + mask transformer, semantic + panoptic ensembling.  `window` (optional
    list) receives HIP-event pairs around the reference's own "inference time" window (`self.unet3d`)."""
    x = net.prepare_input(scene.in_feats, scene.in_coords)
    if window is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ret = net(x, scene.global_min_Cs, scene.global_max_Cs, scene.min_Cs, scene.max_Cs, keep_override=teacher)
    if window is not None:
        e1.record()
        window.append((e0, e1))
    ssc_conf, sem_probs, panop = net.ensemble(ret, scene.Ts)
    return ret, panop


def cpu_baseline(n_infers, in_channels, n1_full, budget_s=30.0):
    """Same graph, oracle backend (C + OpenMP) + torch CPU for the dense parts, host cores only."""
    from oracle.build import build_oracle
    from pasco_amd.me import backend
    from pasco_amd.me.backend import CBackend
    from pasco_amd.graph.synth import make_scene, TeacherKeep
    cores = os.cpu_count() or 1
    torch.set_num_threads(min(cores, 128))
    backend.register_checker_backend(CBackend(build_oracle(), "pho_", "cpu"))
    try:
        net = build_net(n_infers, in_channels, "cpu")
        small = make_scene(0, n_infers=n_infers, in_channels=in_channels, grid=(128, 128, 16))
        tk = TeacherKeep(small, "cpu")
        with torch.no_grad():
            t0 = time.time()
            run_scene(net, small, tk)
            t_small = time.time() - t0
        n1_small = int(small.occ.sum())
        ratio = n1_full / n1_small
        if t_small * ratio <= budget_s:
            full = make_scene(0, n_infers=n_infers, in_channels=in_channels)
            tk = TeacherKeep(full, "cpu")
            with torch.no_grad():
                t0 = time.time()
                run_scene(net, full, tk)
                t_full = time.time() - t0
            return dict(value=1.0 / t_full, unit="scenes/s", cores=min(cores, 128), kind="port",
                        sample=f"1 full S10 scene (seed 0, M={n_infers}), {t_full:.2f} s, no warm-up; "
                               "oracle C/OpenMP sparse ops + torch-CPU dense ops")
        return dict(value=1.0 / (t_small * ratio), unit="scenes/s", cores=min(cores, 128), kind="port",
                    sample=f"1 scene on a 128x128x16 grid ({n1_small} occupied voxels, {t_small:.2f} s), scaled by "
                           f"the occupied-voxel ratio {ratio:.2f} to S10; oracle C/OpenMP sparse ops + torch-CPU dense ops")
    finally:
        backend.register_checker_backend(None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-infers", type=int, default=3)
    ap.add_argument("--in-channels", type=int, default=283)
    ap.add_argument("--heavy", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-launch HIP events")
    ap.add_argument("--no-exact", action="store_true", help="skip the extra exact-fp32-MFMA timing")
    ap.add_argument("--conv-precision", choices=["f32", "f16x3"], default="f16x3",
                    help="f16x3 (default) = conv products as 3 x f16 split MFMA with fp32 accumulation (error vs fp64 <= "
                         "the fp32-MFMA path); f32 = every product on the exact fp32 MFMA")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (MI355X); the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from pasco_amd.me.backend import hip_backend
    from pasco_amd.graph.synth import make_scene, TeacherKeep
    from pasco_amd.graph.profiling import ConvProfiler

    be = hip_backend()   # raises if libpascohip.so is missing
    from pasco_amd.graph import fused
    fused.set_conv_precision(args.conv_precision)
    net = build_net(args.n_infers, args.in_channels, device, heavy=args.heavy)
    scene = make_scene(seed=rank, n_infers=args.n_infers, in_channels=args.in_channels).to(device)
    teacher = TeacherKeep(scene, device)
    prof = ConvProfiler()
    prof.wrap(be)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out, _ = run_scene(net, scene, teacher)
        n1 = int(out["sem_logits_at_scales"][1][0].F.shape[0])
        window = []
        prof.enabled = not args.no_profile
        # the cyclic garbage collector is paused over the timed steps (as a serving loop would): a generation-2
        # pass over the step's many small Python objects costs ~27 ms whenever it lands inside a step
        import gc
        gc.collect()
        gc.freeze()              # model / caches built during warm-up: out of the collector's reach from here on
        gc_was_on = gc.isenabled()
        if os.environ.get("PASCO_BENCH_GC", "0") != "1":
            gc.disable()
        barrier()
        t0 = time.perf_counter()
        marks = [t0]
        for _ in range(args.steps):
            out, panop = run_scene(net, scene, teacher, window)
            marks.append(time.perf_counter())      # host-side enqueue clock of each step (diagnostic, stderr only)
        barrier()
        elapsed = time.perf_counter() - t0
        if gc_was_on:
            gc.enable()
        prof.enabled = False
        if rank == 0:
            per = [round((b - a) * 1e3, 1) for a, b in zip(marks[:-1], marks[1:])]
            print(f"[bench] per-step host ms: {per}", file=sys.stderr, flush=True)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same step with every product on the exact fp32 MFMA, reported next to the headline
    exact = None
    if args.conv_precision == "f16x3" and world == 1 and not args.no_exact:
        fused.set_conv_precision("f32")
        k2 = max(3, args.steps // 2)
        with torch.no_grad():
            run_scene(net, scene, teacher)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k2):
                run_scene(net, scene, teacher)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / k2
        exact = {"value": round(1.0 / dt, 4), "unit": "scenes/s", "ms_per_step": round(dt * 1e3, 3), "steps": k2}
        fused.set_conv_precision(args.conv_precision)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        res = {
            "metric": f"scenes/sec (256x256x32, ~10% occ) PaSCo MIMO-{args.n_infers}",
            "value": round(value, 4), "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.conv_precision == "f32" else "f32 (conv products as 3 x f16 split MFMA, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"PaSCo MIMO M={args.n_infers} ({'heavy' if args.heavy else 'light'} decoder, f=64, "
                                   f"100 queries, {args.in_channels}-ch points), S10 scene 256x256x32 "
                                   f"({int(scene.occ.sum())} occupied voxels, {n1} kept at stride 1), 1 scene/step/GPU",
                       "stages": "point MLP + voxel max, MIMO merge, sparse U-Net (encoder, dense bottleneck, "
                                 "generative decoder), mask transformer, semantic + panoptic ensembling "
                                 "(= Net.forward(return_ensemble=True) + its input stage)",
                       "pruning": "teacher-forced", "parallelism": f"scene-parallel x{world}, no collective"},
        }
        unet_ms = sum(a.elapsed_time(b) for a, b in window) / max(len(window), 1)
        res["unet_window_ms"] = round(unet_ms, 3)   # the reference's own "inference time" window (README.md:448-449)
        res["unet_window_scenes_per_s"] = round(world * 1e3 / unet_ms, 4) if unet_ms > 0 else None
        if not args.no_profile:
            per_kernel = prof.summary()
            if per_kernel:
                res["roofline"] = roofline_object(per_kernel, args.steps)
        if exact is not None:
            res["exact_fp32_mfma"] = exact
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(args.n_infers, args.in_channels, int(scene.occ.sum()))
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "scenes/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
