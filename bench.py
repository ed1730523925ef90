#!/usr/bin/env python
"""bench.py - scenes/sec of PaSCo MIMO-3 inference on the synthetic 256x256x32 (~10 % occupancy)
scene S10 (BASELINE.json metric / SURVEY.md 8(d)).

    python bench.py --gpus N --steps K --warmup W

One "step" = one scene through the hot path: point-feature MLP + voxel max (CylinderFeat), MIMO
input merge, sparse U-Net encoder / dense bottleneck / generative decoder, the mask-transformer
decoder (the reference's `Net.step_inference` up to and including `self.unet3d(...)`,
net_panoptic_sparse.py:548-550,233-245) and semantic + panoptic ensembling.  Inputs are resident in HBM
before the timed region.  Weights are seeded random (no checkpoints offline), BN statistics randomised,
light decoder, teacher-forced pruning.  The timed loop rotates over four different scenes
(seeds 4 r .. 4 r + 3 on rank r).

N > 1: one process per GPU.  Started without a launcher (`python bench.py --gpus N`) the script re-executes
itself under `torch.distributed.run`; under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.
  --mode scenes        (default) every rank runs its own scenes: weak scaling, no data-path collective
  --mode subnet-heads  config C4: MIMO M = --n-infers (8), every rank runs the shared trunk on the SAME scene and
                       only its own subnet heads; one RCCL all-gather of per-voxel logits; strong scaling
Time = max over ranks between two barrier + device-sync fences.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant convolution kernel,
measured live with HIP events on the launch stream, per layer class with both roofs), `cpu_baseline` (the CPU
oracle running the same graph on the host cores - the ONLY place the oracle is used here) and `configs` (short
driver-timed rows of BASELINE.json's other single-GPU configurations).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 (matrix) = vector rate
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense BF16/FP16 MFMA peak
HBM_PEAK_GBS = 8000.0
SPLIT_KERNELS = ("k_conv_dma", "k_conv_h2", "k_conv_f16x3", "k_conv_rl", "k_conv_win", "k_conv_wop", "k_conv_wop2", "k_conv_wide", "k_conv_grid", "k_conv_lin")   # 3 f16 MFMAs per product


# ------------------------------------------------------------------------------------------------------
# roofline
# ------------------------------------------------------------------------------------------------------
def pmc_record(kernel):
    """The newest committed PMC pass that lists `kernel`: profiles/r*_pmc_conv.json (tools/pmc_summary.py from
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command; FETCH_SIZE doubled per
    MI355X_MICROARCH.md).  PMC counters cannot be read from inside the process: a recorded measurement, with
    the commit it was taken at, or None."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_conv.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            k = doc["kernels"][kernel]
            return {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "commit": doc.get("commit"),
                    "file": os.path.relpath(path, ROOT)}
        except Exception:
            continue
    return None


def pmc_traffic(kernel):
    r = pmc_record(kernel)
    return None if r is None else r["hbm_bytes_per_launch"]


HBM_COPY_GBS = None      # stream-copy rate measured on THIS box (measure_stream_copy), the achievable HBM roof


def measure_stream_copy(device, nbytes=1 << 30, reps=5):
    """Achievable HBM rate of this box: a float4 device-to-device copy of 1 GiB (read + write counted), best of `reps`
    ('measure achievable with a stream-copy kernel on the box'; MI355X_MICROARCH.md quotes ~6.3 TB/s)."""
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    best = 0.0
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    return best


def _roofs(name, flops, bytes_alg, time_s, bytes_min=None):
    """Both roofs of one group of launches: algorithmic bytes against the HBM peak, matrix work against the
    MFMA peak of the arithmetic actually used; `bound` = the roof the group sits closer to."""
    tf = flops / time_s / 1e12
    gbs = bytes_alg / time_s / 1e9
    hbm_frac = gbs / HBM_PEAK_GBS
    if name in SPLIT_KERNELS:
        mfma_frac, mfma_peak, mfma_tf = 3.0 * tf / F16_MFMA_PEAK_TFLOPS, F16_MFMA_PEAK_TFLOPS, 3.0 * tf
    else:
        mfma_frac, mfma_peak, mfma_tf = tf / F32_MFMA_PEAK_TFLOPS, F32_MFMA_PEAK_TFLOPS, tf
    e = {"useful_TFLOPs": round(tf, 3), "alg_GBps": round(gbs, 1), "alg_frac_of_hbm_peak": round(hbm_frac, 4),
         "mfma_TFLOPs_issued": round(mfma_tf, 2), "mfma_frac_of_peak": round(mfma_frac, 4)}
    if bytes_min is not None:       # compulsory bytes (every input row once): what an ideal on-chip reuse would leave
        e["min_GBps"] = round(bytes_min / time_s / 1e9, 1)
        e["min_frac_of_hbm_peak"] = round(bytes_min / time_s / 1e9 / HBM_PEAK_GBS, 4)
    if HBM_COPY_GBS:                # the same rates against the stream-copy rate of this box
        e["alg_frac_of_box_copy_rate"] = round(gbs / HBM_COPY_GBS, 4)
    if hbm_frac >= mfma_frac:
        e.update(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(hbm_frac, 4))
    else:
        e.update(bound="mfma", achieved=round(mfma_tf, 2), peak=mfma_peak, unit="TFLOP/s", frac=round(mfma_frac, 4))
    return e


def roofline_object(per_kernel, steps, classes=None):
    """`roofline`: the dominant LAYER CLASS (most time per step in the profiled pass) as the headline entry, every class
    (`by_layer_class`) and every kernel name (`by_kernel`) beside it, and the operand-split passes."""
    def entry(name, s):
        avg = s["time_s"] / s["launches"]
        e = {"kernel": name, "launches_per_step": s["launches"] / steps, "avg_launch_us": round(avg * 1e6, 2),
             "ms_per_step": round(s["time_s"] / steps * 1e3, 3), "flops_per_launch": s["flops"] / s["launches"],
             "alg_bytes_per_launch": s["bytes_alg"] / s["launches"],
             "min_bytes_per_launch": s.get("bytes_min", 0.0) / s["launches"]}
        e.update(_roofs(name, s["flops"], s["bytes_alg"], s["time_s"], s.get("bytes_min")))
        rec = pmc_record(name)
        e["traffic"] = None if rec is None else rec["hbm_bytes_per_launch"]
        if rec is not None:
            e["traffic_source"] = {"file": rec["file"], "commit": rec["commit"]}
        if s.get("k3_launches"):
            e["k3_only"] = {"launches_per_step": s["k3_launches"] / steps,
                            "TFLOPs": round(s["k3_flops"] / max(s["k3_time_s"], 1e-12) / 1e12, 3),
                            "ms_per_step": round(s["k3_time_s"] / steps * 1e3, 3)}
        return e

    names = sorted((n for n in per_kernel if n.startswith("k_conv")), key=lambda n: -per_kernel[n]["time_s"])
    by_kernel = [entry(n, per_kernel[n]) for n in names]
    rows = []
    if classes:
        for (cls, kern), s in sorted(classes.items(), key=lambda kv: -kv[1]["time_s"]):
            r = {"class": cls, "kernel": kern, "launches_per_step": s["launches"] / steps,
                 "ms_per_step": round(s["time_s"] / steps * 1e3, 3),
                 "avg_launch_us": round(s["time_s"] / s["launches"] * 1e6, 1),
                 "flops_per_launch": s["flops"] / s["launches"], "alg_bytes_per_launch": s["bytes_alg"] / s["launches"],
                 "min_bytes_per_launch": s.get("bytes_min", 0.0) / s["launches"]}
            r.update(_roofs(kern, s["flops"], s["bytes_alg"], s["time_s"], s.get("bytes_min")))
            rows.append(r)
    if rows:
        # the headline entry = the LAYER CLASS with the most time (a kernel NAME like k_conv_dma serves k = 1 streams, small
        # 3^3 maps and the 245-offset bottleneck at once: its average means nothing and its mix changes from round to round)
        out = dict(rows[0])
        out["dominant_by"] = "layer class (most time per step)"
        same_name = [r for r in rows if r["kernel"] == out["kernel"]]
        rec = pmc_record(out["kernel"])
        out["traffic"] = None if rec is None else rec["hbm_bytes_per_launch"]
        if rec is not None:
            out["traffic_source"] = {"file": rec["file"], "commit": rec["commit"],
                                     "note": "PMC FETCH_SIZE + WRITE_SIZE per launch, mean over the launches of this kernel name"
                                             + ("" if len(same_name) == 1 else f" ({len(same_name)} layer classes run on it)")}
    else:
        out = dict(by_kernel[0])
    on_chip = False
    if out.get("kernel") in ("k_conv_wop", "k_conv_wop2", "k_conv_win") and out.get("alg_bytes_per_launch"):
        # the window kernels (conv_win.hip, conv_wop.hip) gather from LDS: what crosses HBM is the PMC traffic where a record of
        # this kernel name exists, else about the compulsory bytes (every input row once per chunk)
        moved = out.get("traffic") or out.get("min_bytes_per_launch") or 0.0
        on_chip = 0.0 < moved / out["alg_bytes_per_launch"] < 0.5
    if on_chip and out.get("bound") == "hbm":
        # SURVEY.md 8(d)'s algorithmic bytes count every gathered row once per kernel-map pair; this kernel serves the repeats from
        # LDS-resident windows, so the HBM it moves is a fraction of them (the quotient can exceed the HBM peak): the roof it sits
        # under is the matrix pipe, and THAT is the headline - the algorithmic-byte figure stays beside it
        moved = out.get("traffic") or out.get("min_bytes_per_launch")
        out["alg_roof"] = {"bound": "hbm", "achieved": out["achieved"], "peak": out["peak"], "unit": out["unit"], "frac": out["frac"],
                           "note": "algorithmic bytes / time (SURVEY.md 8(d)); the gathers are LDS hits: "
                                   f"{moved / max(out['avg_launch_us'], 1e-9) / 1e3:.0f} GB/s actually cross HBM"}
        out.update(bound="mfma", achieved=out["mfma_TFLOPs_issued"], peak=F16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                   frac=out["mfma_frac_of_peak"])
        out["note"] = ("the class's gathers are served from LDS windows: headline roof = f16 MFMA peak (3 issued products per useful "
                       "one); alg_roof = SURVEY 8(d)'s algorithmic bytes against the HBM peak")
        out["matrix_roof"] = {"bound": "mfma", "achieved": out["mfma_TFLOPs_issued"], "peak": F16_MFMA_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": out["mfma_frac_of_peak"]}
    if HBM_COPY_GBS:
        out["hbm_box_copy_GBps"] = round(HBM_COPY_GBS, 1)       # float4 copy of 1 GiB on this box, read + write
    out["conv_ms_per_step"] = round(sum(per_kernel[n]["time_s"] for n in names) / steps * 1e3, 3)
    out["by_kernel"] = by_kernel
    sp = per_kernel.get("k_split_rows")
    if sp:     # operand preparation of the split kernels (one pass per conv input, not per gather)
        out["operand_split"] = {"kernel": "k_split_rows", "launches_per_step": sp["launches"] / steps,
                                "ms_per_step": round(sp["time_s"] / steps * 1e3, 3),
                                "alg_GBps": round(sp["bytes_alg"] / sp["time_s"] / 1e9, 1),
                                "traffic": pmc_traffic("k_split_rows")}
    if rows:
        out["by_layer_class"] = rows
    return out


# ------------------------------------------------------------------------------------------------------
# workload
# ------------------------------------------------------------------------------------------------------
def build_net(n_infers, in_channels, device, heavy=False, n_classes=20):
    from pasco_amd.graph import PascoNet
    torch.manual_seed(1234)
    net = PascoNet(n_classes=n_classes, n_infers=n_infers, in_channels=in_channels, f=64, num_queries=100,
                   heavy_decoder=heavy)
    g = torch.Generator().manual_seed(4321)
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) * 0.4 + 0.8)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
    return net.eval().to(device)


def run_scene(net, scene, teacher, window=None, panoptic=False):
    """One step = the reference's `Net.forward(return_ensemble=True)` preceded by its input stage:
    point MLP + merge, U-Net + mask transformer, semantic + panoptic ensembling.  `window` (optional
    list) receives HIP-event pairs around the reference's own "inference time" window (`self.unet3d`).
    `panoptic`: also `panoptic_inference` of the M subnets' outputs and the ensemble's = the whole of the reference's
    `Net.step_inference` (net_panoptic_sparse.py:539-608) without its metric bookkeeping."""
    if hasattr(teacher, "begin_step"):
        teacher.begin_step()
    x = net.prepare_input(scene.in_feats, scene.in_coords)
    if window is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ret = net(x, scene.global_min_Cs, scene.global_max_Cs, scene.min_Cs, scene.max_Cs, keep_override=teacher)
    if window is not None:
        e1.record()
        window.append((e0, e1))
    ssc_conf, sem_probs, panop = net.ensemble(ret, scene.Ts)
    if panoptic:
        net.panoptic(panop, ssc_conf)
    return ret, panop


def run_scene_subnet_heads(net, scene, teacher):
    """Config C4 step: shared trunk on every rank, this rank's subnet heads, one exchange, ensembling.  Default exchange
    (round 5): all-to-all of canonical-site SLABS of the resampled masks + a site-sharded ensembler (pasco_amd/graph/dist.py
    `site_sharded_ensemble`: each rank receives (W - 1) / W of ONE mask tensor instead of W - 1 whole ones, results
    bit-identical); PASCO_C4_EXCHANGE=allgather (or f16): the all-gather by subnet of rounds 2 - 4."""
    from pasco_amd.graph.dist import site_sharded_ensemble, subnet_parallel_forward, subnet_parallel_local
    x = net.prepare_input(scene.in_feats, scene.in_coords)
    if os.environ.get("PASCO_C4_EXCHANGE", "slab") == "slab":
        ret = subnet_parallel_local(net, x, scene.global_min_Cs, scene.global_max_Cs, scene.min_Cs, scene.max_Cs,
                                    keep_override=teacher)
        sem_probs, sharded, stats = site_sharded_ensemble(net, ret, scene.Ts)
        ret["exchange"] = stats
        return ret, sharded
    ret = subnet_parallel_forward(net, x, scene.global_min_Cs, scene.global_max_Cs, scene.min_Cs, scene.max_Cs,
                                  keep_override=teacher)
    ssc_conf, sem_probs, panop = net.ensemble(ret, scene.Ts)
    return ret, panop


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def physical_cores():
    """Physical cores of the host (SMT siblings counted once): the C / OpenMP oracle ran 2.7x slower on all 256
    logical CPUs of the 128-core box than on 128 threads."""
    seen = set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return len(seen) or (os.cpu_count() or 1)


def cpu_baseline(n_infers, in_channels, timed=5):
    """The CPU baseline in a FRESH process with the whole host's CPUs: this process (and every thread pool it has started) is
    pinned to its GPU's socket, which would halve the baseline's cores."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--n-infers", str(n_infers), "--in-channels",
           str(in_channels), "--cpu-baseline-scenes", str(timed)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMP_NUM_THREADS")}
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1800)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        raise RuntimeError(f"cpu baseline process failed ({out.returncode}): {out.stderr[-400:]}")
    return json.loads(lines[-1])


def cpu_baseline_here(n_infers, in_channels, timed=5):
    """Same graph, oracle backend (C + OpenMP) + torch CPU for the dense parts, host cores only: ONE warm-up scene on the
    same 256x256x32 grid (thread pools, page faults, weight operand caches), then `timed` full S10 scenes (seeds 0 .. timed - 1),
    median reported (one warm-up + 5 timed scenes, median; ~23 s per scene on the 128-core host)."""
    from oracle.build import build_oracle
    from pasco_amd.me import backend
    from pasco_amd.me.backend import CBackend
    from pasco_amd.graph.synth import make_scene, TeacherKeep
    if hasattr(os, "sched_setaffinity"):           # the GPU loop was pinned to its GPU's socket: the CPU baseline gets the whole host
        try:
            os.sched_setaffinity(0, range(os.cpu_count() or 1))
        except OSError:
            pass
    cores = physical_cores()
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)     # the oracle's OpenMP runtime starts with its first call, below
    backend.register_checker_backend(CBackend(build_oracle(), "pho_", "cpu"))
    try:
        net = build_net(n_infers, in_channels, "cpu")
        times, occ = [], 0
        for k in range(timed + 1):                 # scene 0 twice: once as the warm-up, once timed
            full = make_scene(max(k - 1, 0), n_infers=n_infers, in_channels=in_channels)
            tk = TeacherKeep(full, "cpu")
            with torch.no_grad():
                t0 = time.time()
                run_scene(net, full, tk)
                dt = time.time() - t0
            if k > 0:
                times.append(dt)
            else:
                occ = int(full.occ.sum())
        times.sort()
        med = times[len(times) // 2]
        return dict(value=round(1.0 / med, 5), unit="scenes/s", cores=cores, kind="port",
                    sample=f"median of {timed} full S10 scenes (seeds 0..{timed - 1}, M={n_infers}, {occ} occupied voxels in seed 0; "
                           f"{', '.join(f'{t:.2f}' for t in times)} s) after one warm-up scene on the same grid; oracle C/OpenMP "
                           f"sparse ops + torch-CPU dense ops; {cpu_model()}: {cores} threads = physical cores of "
                           f"{os.cpu_count()} logical CPUs (OpenMP and torch)")
    finally:
        backend.register_checker_backend(None)


# ------------------------------------------------------------------------------------------------------
# the printed line
# ------------------------------------------------------------------------------------------------------
LINE_LIMIT = 6144      # bytes of the ONE JSON line on stdout (the driver keeps ~10 KB of stdout tail: round 5's 20.5 KB line
                       # could not be parsed).  tests/test_bench_contract.py holds a full-size result to this bound.

_ROOF_KEYS = ("class", "kernel", "launches_per_step", "ms_per_step", "avg_launch_us", "flops_per_launch", "alg_bytes_per_launch",
              "min_bytes_per_launch", "bound", "achieved", "peak", "unit", "frac", "traffic", "matrix_roof", "alg_roof", "mfma_frac_of_peak",
              "alg_frac_of_hbm_peak", "min_frac_of_hbm_peak", "hbm_box_copy_GBps", "conv_ms_per_step")


def write_detail(res):
    """Everything the run measured (per kernel, per layer class, allocator, warm-up, per-round step times) goes to a side file;
    the stdout line carries the contract fields only.  Returns the path written (or None)."""
    path = os.environ.get("PASCO_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump(res, f, indent=1)
        return path
    except OSError as e:       # a read-only checkout must not take the line down
        print(f"[bench] detail file not written: {e}", file=sys.stderr, flush=True)
        return None


def _num(x, nd=4):
    if isinstance(x, float):
        return round(x, nd) if abs(x) < 1e6 else float(f"{x:.6g}")
    return x


def compact_line(res, detail_path=None):
    """The ONE JSON line: the driver's contract fields, `config` (workload + also_measured), `roofline` of the headline layer
    class only, `cpu_baseline`, the other configurations as bare numbers.  Bounded by LINE_LIMIT."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "per_rank_ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config")
    line = {k: res[k] for k in keep if k in res}
    rf = res.get("roofline")
    if rf:
        r = {k: _num(rf[k]) for k in _ROOF_KEYS if k in rf}
        src = rf.get("traffic_source")
        if src:
            r["traffic_source"] = f"{src.get('file')} @ {src.get('commit')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, per launch)"
        if rf.get("note"):
            r["note"] = rf["note"][:300]
        # every layer class of the convolution path in one short row each: [class, kernel, launches/step, ms/step, bound, frac]
        rows = rf.get("by_layer_class") or []
        r["classes"] = [[c["class"], c["kernel"], _num(c["launches_per_step"], 1), _num(c["ms_per_step"], 3), c["bound"],
                         _num(c["frac"], 3)] for c in rows[:12]]
        line["roofline"] = r
    if "cpu_baseline" in res:
        cb = dict(res["cpu_baseline"])
        if isinstance(cb.get("sample"), str) and len(cb["sample"]) > 420:
            cb["sample"] = cb["sample"][:417] + "..."
        line["cpu_baseline"] = cb
    if "configs" in res:
        line["configs"] = {k: (v.get("scenes_per_s") if "scenes_per_s" in v else {"error": str(v.get("error"))[:80]})
                           for k, v in res["configs"].items()}
    for k in ("in_flight_1", "exact_fp32_mfma", "gc_enabled"):
        if k in res:
            line[k] = {a: b for a, b in res[k].items() if a in ("value", "ms_per_step", "steps", "unet_window_ms", "device_mallocs")}
    if "step_inference" in res:
        si = res["step_inference"]
        line["step_inference"] = {a: b for a, b in si.items() if a != "what"}
    for k in ("unet_window_ms", "host_cpu_ms_per_step", "fallbacks", "query_graph", "exchange", "bound_note"):
        if k in res:
            line[k] = res[k]
    if "step_ms_by_round" in res:
        line["step_ms_by_round"] = {k: res["step_ms_by_round"][k] for k in ("group", "median", "min", "max")
                                    if k in res["step_ms_by_round"]}
    if "allocator" in res:
        line["device_mallocs_in_timed_loop"] = res["allocator"].get("device_mallocs_in_timed_loop")
    if detail_path:
        line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path
    # last resort: drop optional keys until the line fits (never the contract fields, roofline or cpu_baseline)
    for k in ("step_ms_by_round", "gc_enabled", "query_graph", "host_cpu_ms_per_step", "exchange", "bound_note", "configs"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(k, None)
    if len(json.dumps(line)) > LINE_LIMIT and "roofline" in line:
        line["roofline"].pop("classes", None)
    return line


# ------------------------------------------------------------------------------------------------------
# launcher
# ------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """Launcher self-test (tests/test_dist_gloo.py): the process-group / fence / max-over-ranks / JSON plumbing on
    CPU + gloo with a sleep instead of a scene.  Never a measurement: `data` says so."""
    import torch.distributed as dist
    from pasco_amd.graph.dist import timed_steps
    if world > 1:
        dist.init_process_group("gloo")
    elapsed = timed_steps(lambda: time.sleep(0.005 * (1 + rank)), args.steps, args.warmup)
    per_rank = [elapsed]
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, elapsed)
        per_rank = box
    if rank == 0:
        print(json.dumps({"metric": "dry-run (launcher self-test, no compute)", "value": world * args.steps / elapsed,
                          "unit": "steps/s", "n_gpus": world, "n_ranks": dist.get_world_size() if world > 1 else 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "per_rank_ms_per_step": [p / args.steps * 1e3 for p in per_rank],
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                          "data": "dry-run (no compute)", "config": {"workload": "sleep"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def allocator_state(device):
    """Caching-allocator numbers that explain a slow run: memory reserved / in use, and how often an allocation only succeeded
    after the allocator had to give its cached blocks back to the driver (every big tensor after that is a fresh hipMalloc)."""
    try:
        st = torch.cuda.memory_stats(device)
        return {"reserved_GB": round(st.get("reserved_bytes.all.current", 0) / 2 ** 30, 2),
                "reserved_peak_GB": round(st.get("reserved_bytes.all.peak", 0) / 2 ** 30, 2),
                "allocated_peak_GB": round(st.get("allocated_bytes.all.peak", 0) / 2 ** 30, 2),
                "alloc_retries": int(st.get("num_alloc_retries", 0)), "ooms": int(st.get("num_ooms", 0)),
                "device_mallocs": int(st.get("num_device_alloc", 0)), "device_frees": int(st.get("num_device_free", 0))}
    except Exception as e:      # diagnostics only
        return {"error": f"{type(e).__name__}: {e}"}


def short_row(n_infers, in_channels, n_classes, device, steps=3, unfused=False, heavy=False, me_conv="guarded"):
    """Short timed row of another BASELINE.json configuration on this GPU (1 warm-up + `steps` scenes)."""
    from pasco_amd.graph import fused
    from pasco_amd.graph.synth import make_scene, TeacherKeep
    net = build_net(n_infers, in_channels, device, n_classes=n_classes, heavy=heavy)
    scene = make_scene(seed=0, n_infers=n_infers, in_channels=in_channels).to(device)
    teacher = TeacherKeep(scene, device)
    if unfused:                 # INTEGRATION.md route (a): reference-style module sequence on the plain pasco_amd.me modules
        from pasco_amd.me import modules as me_modules
        fused.set_fusion(False)
        # "guarded" (the modules' default): split-precision kernels + exact fp32 device-side fallback, the dense bottleneck (a
        # torch Conv3d stack in the reference, outside the ME surface) on the implicit-GEMM kernel at the default precision;
        # "exact": every product on the exact fp32 MFMA, modules and bottleneck alike (what rounds 2 - 4 reported)
        if me_conv == "exact":
            fused.set_conv_precision("f32")
        me_modules.set_me_conv(me_conv)
    try:
        with torch.no_grad():
            teacher.prepare(lambda: run_scene(net, scene, teacher))       # scene preparation (also the warm-up step)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                run_scene(net, scene, teacher)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
    finally:
        if unfused:
            fused.set_fusion(True)
            fused.set_conv_precision("f16x3")
            me_modules.set_me_conv("guarded")
    row = {"scenes_per_s": round(1.0 / dt, 3), "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": 1,
           "allocator": allocator_state(device)}
    redone = {k: int(getattr(net, k, 0)) for k in ("range_fallbacks", "input_fallbacks", "optimistic_fallbacks") if getattr(net, k, 0)}
    if redone:          # steps that ran twice (a fallback of PascoNet.forward): the row is then not a clean measurement
        row["redone_steps"] = redone
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--mode", choices=["scenes", "subnet-heads"], default="scenes")
    ap.add_argument("--n-infers", type=int, default=None, help="MIMO subnets (default 3; 8 with --mode subnet-heads)")
    ap.add_argument("--in-channels", type=int, default=283)
    ap.add_argument("--n-classes", type=int, default=20)
    ap.add_argument("--heavy", action="store_true")
    ap.add_argument("--scenes", type=int, default=4, help="different scenes the timed loop rotates over")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="scenes in flight per GPU: worker threads, each with its own HIP stream, take the steps from a shared "
                         "counter (a step's ~60 host synchronisations then overlap with the other scene's kernels); 1 = one "
                         "scene at a time, also measured and reported as `in_flight_1`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-launch HIP events")
    ap.add_argument("--no-exact", action="store_true", help="skip the extra exact-fp32-MFMA timing")
    ap.add_argument("--no-configs", action="store_true", help="skip the short rows of the other configurations")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-scenes", type=int, default=5, help=argparse.SUPPRESS)
    ap.add_argument("--conv-precision", choices=["f32", "f16x3"], default="f16x3",
                    help="f16x3 (default) = conv products as 3 x f16 split MFMA with fp32 accumulation (error vs fp64 <= "
                         "the fp32-MFMA path); f32 = every product on the exact fp32 MFMA")
    args = ap.parse_args()
    if args.n_infers is None:
        args.n_infers = 8 if args.mode == "subnet-heads" else 3

    if args.cpu_baseline_only:           # child of `cpu_baseline`: nothing but the oracle + torch-CPU graph, all host CPUs
        print(json.dumps(cpu_baseline_here(args.n_infers, args.in_channels, args.cpu_baseline_scenes)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (MI355X); the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # each rank (and the worker threads it starts) stays on whole physical cores next to its GPU: PASCO_BENCH_PIN=0 disables
    from pasco_amd.graph.dist import pin_rank
    pinned = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), device_index=local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    from pasco_amd.me.backend import hip_backend
    from pasco_amd.graph.synth import make_scene, TeacherKeep
    from pasco_amd.graph.profiling import ConvProfiler

    be = hip_backend()   # raises if libpascohip.so is missing
    global HBM_COPY_GBS
    if rank == 0 and not args.no_profile:
        HBM_COPY_GBS = measure_stream_copy(device)
    from pasco_amd.graph import fused
    fused.set_conv_precision(args.conv_precision)
    net = build_net(args.n_infers, args.in_channels, device, heavy=args.heavy, n_classes=args.n_classes)
    heads = args.mode == "subnet-heads"
    # scenes-mode: rank r owns its own scenes; subnet-heads: every rank works on the SAME scene sequence
    seeds = [(0 if heads else rank * args.scenes) + i for i in range(args.scenes)]
    scenes = [make_scene(seed=s, n_infers=args.n_infers, in_channels=args.in_channels).to(device) for s in seeds]
    teachers = [TeacherKeep(sc, device) for sc in scenes]
    if not (args.mode == "subnet-heads" and world > 1) and os.environ.get("PASCO_BENCH_TEACHER_LIVE", "0") != "1":
        # scene preparation: the teacher-forced keep's hash lookups are answered once per scene here, outside every timed region
        with torch.no_grad():
            for sc, tk_ in zip(scenes, teachers):
                tk_.prepare(lambda sc=sc, tk_=tk_: run_scene(net, sc, tk_))
        torch.cuda.synchronize()
    step_fn = run_scene_subnet_heads if (heads and world > 1) else (lambda n, s, t, w=None: run_scene(n, s, t, w))
    prof = ConvProfiler()
    prof.wrap(be)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    import threading

    from pasco_amd.graph.serve import SceneServer, vet_cached_blocks
    if heads and world > 1:
        args.in_flight = 1                  # one communicator: collectives are issued from one thread
    ctx = {"window": None, "panoptic": False}

    def one_step(i):
        """Step i of the loop: scene i mod #scenes through the whole hot path."""
        j = i % len(scenes)
        if heads and world > 1:
            return step_fn(net, scenes[j], teachers[j])
        return run_scene(net, scenes[j], teachers[j], ctx["window"], ctx["panoptic"])

    # worker threads, each bound to its own HIP stream, draw step numbers from a shared counter (pasco_amd/graph/serve.py)
    server = SceneServer(device, one_step, in_flight=args.in_flight,
                         switch_interval_ms=float(os.environ.get("PASCO_BENCH_SWITCH_MS", "0.5")))

    def run_steps(first, count, in_flight, window=None, marks=None):
        """Steps first .. first + count - 1 (scene = step mod #scenes), `in_flight` at a time (1 = on the current stream)."""
        ctx["window"] = window
        last = {}

        def done(i, o):
            last["out"], last["panop"] = o
            if marks is not None:
                marks.append(time.perf_counter())

        server.run(range(first, first + count), in_flight=in_flight, on_done=done)
        return last

    # warm-up (untimed): at least --warmup steps and every scene once on the main stream, every scene on every worker stream,
    # then the in-flight loop until the caching allocator has stopped growing; then every large cached block is write-tested
    last = run_steps(0, max(args.warmup, len(scenes)), 1)
    warm_info = server.warm(range(len(scenes)))
    vet = None
    if os.environ.get("PASCO_BENCH_VET", "1") != "0":
        vet = vet_cached_blocks(device, [torch.cuda.current_stream(device)] + server.streams)
        if vet.get("replaced"):              # replacements are new blocks: let the pools settle again
            warm_info["after_vet"] = server.warm(range(len(scenes)), max_rounds=3)
    out = last["out"]
    n1 = int(out["sem_logits_at_scales"][1][0].F.shape[0])
    window = []
    # per-launch HIP events only mean something when one scene runs at a time (kernels of two streams share the GPU):
    # with several scenes in flight the roofline comes from a one-at-a-time pass of its own below
    prof.enabled = (not args.no_profile) and args.in_flight <= 1
    prof_steps = args.steps
    # the cyclic garbage collector is paused over the timed steps (as a serving loop would): a generation-2
    # pass over the step's many small Python objects costs ~27 ms whenever it lands inside a step; the same
    # loop with the collector running is reported beside it (`gc_enabled`)
    gc.collect()
    gc.freeze()              # model / caches built during warm-up: out of the collector's reach from here on
    gc_was_on = gc.isenabled()
    if os.environ.get("PASCO_BENCH_GC", "0") != "1":
        gc.disable()
    # the write test above walked every cached block and the collector has just run: a last untimed stretch of the loop itself, so that
    # the timed steps start from the state a serving loop is in (one box's first process ran its first 20 timed steps 25 - 45 % slow
    # right after them: profiles/r4x_bench_full.json)
    if args.in_flight > 1:
        run_steps(0, max(2 * args.in_flight * len(scenes), args.steps), args.in_flight)
    alloc_log = {"after_warmup": allocator_state(device)}
    barrier()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    marks = [t0]
    last = run_steps(0, args.steps, args.in_flight, window, marks)
    barrier()
    elapsed = time.perf_counter() - t0
    host_cpu_s = time.process_time() - cpu0      # CPU seconds of every thread of this process over the timed loop
    alloc_log["after_timed_loop"] = allocator_state(device)
    out, panop = last["out"], last["panop"]
    prof.enabled = False
    # the timed steps in groups of `in_flight` completions: a box that goes through a slow period (seen in rounds 2 - 3:
    # the HBM-bound launches 6 x slower for ~0.3 s, then normal again) shows here, not only in the mean
    g = max(args.in_flight, 1)
    rounds_ms = [round((marks[min(i + g, len(marks) - 1)] - marks[i]) * 1e3 / (min(i + g, len(marks) - 1) - i), 2)
                 for i in range(0, len(marks) - 1, g)]
    per_rank_rounds = None
    if world > 1:                           # every rank's rounds: a sub-linear curve can then be attributed to a rank / a period
        per_rank_rounds = [None] * world
        dist.all_gather_object(per_rank_rounds, rounds_ms)
    if rank == 0:
        per = [round((b - a) * 1e3, 1) for a, b in zip(marks[:-1], marks[1:])]
        print(f"[bench] per-step host ms: {per}", file=sys.stderr, flush=True)
    one_in_flight = step_inference = None
    if args.in_flight > 1 and world == 1:          # the same K steps, one scene at a time on one stream (no per-launch events)
        run_steps(0, len(scenes), 1)               # untimed: the main stream's pool as the collector-less loop leaves it
        window = []
        mallocs0 = allocator_state(device).get("device_mallocs", 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(0, args.steps, 1, window)
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / args.steps
        one_in_flight = {"value": round(1.0 / dt1, 4), "unit": "scenes/s", "ms_per_step": round(dt1 * 1e3, 3),
                         "steps": args.steps}
        alloc_log["after_in_flight_1"] = allocator_state(device)
        one_in_flight["device_mallocs"] = alloc_log["after_in_flight_1"].get("device_mallocs", 0) - mallocs0
        if window:      # the reference's own timing window (`self.unet3d`, README.md:448-449), one scene at a time
            one_in_flight["unet_window_ms"] = round(sum(a.elapsed_time(b) for a, b in window) / len(window), 3)
        # the same loop with `panoptic_inference` of the M + 1 outputs behind every step: Net.step_inference as a whole
        ctx["panoptic"] = True
        try:
            run_steps(0, len(scenes), 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(0, args.steps, 1)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t0) / args.steps
            step_inference = {"ms_per_step": round(dts * 1e3, 3), "scenes_per_s": round(1.0 / dts, 4), "steps": args.steps,
                              "in_flight": 1, "panoptic_added_ms": round((dts - dt1) * 1e3, 3),
                              "what": "in_flight_1's step + panoptic_inference of the M subnets' outputs and the ensemble's "
                                      "(3 launches each on the sparse rows, one device->host copy of the segment tables)"}
            if args.in_flight > 1:
                run_steps(0, args.in_flight * len(scenes), args.in_flight)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_steps(0, args.steps, args.in_flight)
                torch.cuda.synchronize()
                dtf = (time.perf_counter() - t0) / args.steps
                step_inference["in_flight_%d" % args.in_flight] = {"ms_per_step": round(dtf * 1e3, 3),
                                                                   "scenes_per_s": round(1.0 / dtf, 4)}
        finally:
            ctx["panoptic"] = False
        # the roofline's per-launch HIP events: a separate one-at-a-time pass, so that neither the events nor the pair
        # counts the profiler launches perturb a reported step time
        if not args.no_profile:
            prof.enabled = True
            mallocs0 = allocator_state(device).get("device_mallocs", 0)
            kprof = max(8, args.steps // 2)
            run_steps(0, kprof, 1)
            torch.cuda.synchronize()
            prof.enabled = False
            prof_steps = kprof
            alloc_log["after_profile_pass"] = allocator_state(device)
            alloc_log["device_mallocs_in_profile_pass"] = alloc_log["after_profile_pass"].get("device_mallocs", 0) - mallocs0
    if gc_was_on:
        gc.enable()

    per_rank = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        box = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(box, t)
        per_rank = [float(b.item()) for b in box]
        elapsed = max(per_rank)

    single = world == 1
    gc_row = None
    if single:                    # the same loop with Python's collector running (what an unmanaged serving loop pays)
        k3 = max(4, args.steps // 3)
        gc.enable()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(0, k3, args.in_flight)
        torch.cuda.synchronize()
        gc_row = {"ms_per_step": round((time.perf_counter() - t0) / k3 * 1e3, 3), "steps": k3, "in_flight": args.in_flight}

    # the same step with every product on the exact fp32 MFMA, reported next to the headline
    exact = None
    if args.conv_precision == "f16x3" and single and not args.no_exact:
        fused.set_conv_precision("f32")
        k2 = max(4, args.steps // 3)
        with torch.no_grad():
            run_scene(net, scenes[0], teachers[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(k2):
                run_scene(net, scenes[i % len(scenes)], teachers[i % len(scenes)])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / k2
        exact = {"value": round(1.0 / dt, 4), "unit": "scenes/s", "ms_per_step": round(dt * 1e3, 3), "steps": k2}
        fused.set_conv_precision(args.conv_precision)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        scenes_done = args.steps if heads else world * args.steps     # subnet-heads: ONE scene per step for the whole job
        value = scenes_done / elapsed
        occ = int(scenes[0].occ.sum())
        if heads:
            metric = f"scenes/sec (256x256x32, ~10% occ) PaSCo MIMO-{args.n_infers}, one subnet head per GPU"
            par = f"subnet-heads x{world}: trunk replicated, heads sharded, RCCL all-gather of per-voxel logits"
        else:
            metric = f"scenes/sec (256x256x32, ~10% occ) PaSCo MIMO-{args.n_infers}"
            par = f"scene-parallel x{world}, no collective"
        res = {
            "metric": metric,
            "value": round(value, 4), "unit": "scenes/s", "n_gpus": world,
            "n_ranks_rccl": dist.get_world_size() if world > 1 else 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "per_rank_ms_per_step": [round(p / args.steps * 1e3, 3) for p in per_rank],
            "higher_is_better": True, "scaling": "strong" if heads else "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.conv_precision == "f32" else "f32 (conv products as 3 x f16 split MFMA, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"PaSCo MIMO M={args.n_infers} ({'heavy' if args.heavy else 'light'} decoder, f=64, "
                                   f"100 queries, {args.in_channels}-ch points, {args.n_classes} classes), S10 scenes 256x256x32 "
                                   f"(seed {seeds[0]}: {occ} occupied voxels, {n1} kept at stride 1), "
                                   f"{len(scenes)} different scenes rotated, 1 scene/step" + ("" if heads else "/GPU"),
                       "stages": "point MLP + voxel max, MIMO merge, sparse U-Net (encoder, dense bottleneck, "
                                 "generative decoder), mask transformer, semantic + panoptic ensembling "
                                 "(= Net.forward(return_ensemble=True) + its input stage)",
                       "pruning": "teacher-forced (the keep sets' hash lookups answered at scene preparation)", "parallelism": par},
        }
        res["allocator"] = alloc_log
        res["allocator"]["device_mallocs_in_timed_loop"] = (alloc_log["after_timed_loop"].get("device_mallocs", 0) -
                                                            alloc_log["after_warmup"].get("device_mallocs", 0))
        res["warmup_detail"] = warm_info
        res["warmup_detail"]["allocator_rounding"] = server.allocator_rounding
        if vet is not None:
            res["memory_vet"] = vet
        if rounds_ms:
            srt = sorted(rounds_ms)
            res["step_ms_by_round"] = {"group": max(args.in_flight, 1), "median": srt[len(srt) // 2], "min": srt[0], "max": srt[-1],
                                       "rounds": rounds_ms,
                                       "note": "ms per step over each group of `group` consecutive completions of the timed loop "
                                               "(value / ms_per_step are the mean over all of them)"}
            if per_rank_rounds is not None:
                res["step_ms_by_round"]["per_rank_rounds"] = per_rank_rounds
        if heads:
            res["bound_note"] = ("trunk replicated on every rank: speed-up over 1 GPU is bounded by (T + 8 H) / (T + H) "
                                 "~ 2.2x at S10 (T ~ 2.1 TFLOP trunk, H ~ 0.43 TFLOP per subnet head; SURVEY.md 8(e))")
        if window:
            unet_ms = sum(a.elapsed_time(b) for a, b in window) / len(window)
            res["unet_window_ms"] = round(unet_ms, 3)   # the reference's own "inference time" window (README.md:448-449)
            res["unet_window_scenes_per_s"] = round(world * 1e3 / unet_ms, 4) if unet_ms > 0 else None
        if not args.no_profile:
            per_kernel, classes = prof.summary(by_class=True)
            if per_kernel:
                res["roofline"] = roofline_object(per_kernel, prof_steps, classes)
        res["config"]["in_flight"] = args.in_flight
        res["host_cpu_ms_per_step"] = round(host_cpu_s / args.steps * 1e3, 3)      # CPU time of ALL threads (the workers spin inside HIP's waits: ~one core each)
        res["host_threads"] = {"scene_threads_per_rank": max(args.in_flight, 1), "ranks": world,
                               "cpus_per_rank": len(pinned) if pinned else len(os.sched_getaffinity(0)),
                               "pinned_to_gpu_local_cores": bool(pinned), "omp_num_threads": os.environ.get("OMP_NUM_THREADS"),
                               "note": "host side of a scene = Python launch loop under the GIL; ranks x scene threads "
                                       "compete for cores only across ranks (each rank pinned to its own cores)"}
        res["query_graph"] = net.transformer_predictor.query_graph_state()
        # steps that had to be redone (each costs a second pass): f16 range -> exact fp32, fused input stage -> general route,
        # an optimistic shortcut that did not hold -> checked paths.  All 0 on the S10 scenes.
        res["fallbacks"] = {"f16_range": int(getattr(net, "range_fallbacks", 0)), "input_stage": int(getattr(net, "input_fallbacks", 0)),
                            "optimistic": int(getattr(net, "optimistic_fallbacks", 0))}
        if heads and world > 1 and last.get("out") is not None and "exchange" in last["out"]:
            ex = dict(last["out"]["exchange"])
            ex["MB_sent_per_rank_per_scene"] = round(ex["bytes_sent"] / 1e6, 2)
            ex["MB_received_per_rank_per_scene"] = round(ex.get("bytes_received", 0) / 1e6, 2)
            ex["kind"] = os.environ.get("PASCO_C4_EXCHANGE", "slab")
            res["exchange"] = ex
        if one_in_flight is not None:
            res["in_flight_1"] = one_in_flight
        if step_inference is not None:
            res["step_inference"] = step_inference
        if gc_row is not None:
            res["gc_enabled"] = gc_row
        if exact is not None:
            res["exact_fp32_mfma"] = exact
        # the figures the headline's caveats depend on, repeated inside `config` (a key the driver's record keeps): one scene
        # at a time on one stream, every product on the exact fp32 MFMA, the reference's own `self.unet3d` timing window
        also = {}
        if one_in_flight is not None:
            also["in_flight_1_scenes_per_s"] = one_in_flight["value"]
            also["in_flight_1_ms_per_step"] = one_in_flight["ms_per_step"]
            if "unet_window_ms" in one_in_flight:
                also["in_flight_1_unet_window_ms"] = one_in_flight["unet_window_ms"]
        if step_inference is not None:
            also["step_inference_ms"] = step_inference["ms_per_step"]
            also["step_inference_panoptic_added_ms"] = step_inference["panoptic_added_ms"]
        if exact is not None:
            also["exact_fp32_mfma_scenes_per_s"] = exact["value"]
        if "unet_window_ms" in res:
            also["unet_window_ms"] = res["unet_window_ms"]
        if also:
            res["config"]["also_measured"] = also
        if single and not args.no_configs and not heads and not args.heavy and args.n_infers == 3:
            del scenes, teachers, net, out, panop
            torch.cuda.empty_cache()
            rows = {}
            for name, kw in (("mimo1_semantickitti", dict(n_infers=1, in_channels=283, n_classes=20)),
                             ("mimo3_sscbench_kitti360", dict(n_infers=3, in_channels=8, n_classes=19)),
                             ("mimo8_one_gpu", dict(n_infers=8, in_channels=283, n_classes=20)),
                             # SURVEY.md 8(d) "second row": the logged run's decoder depth (hparams.yaml heavy_decoder: true)
                             ("mimo3_heavy_decoder", dict(n_infers=3, in_channels=283, n_classes=20, heavy=True)),
                             ("mimo1_unfused_me_modules", dict(n_infers=1, in_channels=283, n_classes=20, unfused=True)),
                             ("mimo1_unfused_me_modules_exact", dict(n_infers=1, in_channels=283, n_classes=20, unfused=True,
                                                                     me_conv="exact"))):
                try:
                    rows[name] = short_row(device=device, **kw)
                except Exception as e:  # a side row must never take the headline down
                    rows[name] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
            res["configs"] = rows
        if not args.no_cpu_baseline and single:
            try:
                res["cpu_baseline"] = cpu_baseline(args.n_infers, args.in_channels)
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "scenes/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        detail = write_detail(res)
        line = compact_line(res, detail)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
