"""bench.py's JSON contract pieces that do not need a GPU: the roofline object built from per-kernel records and
the committed PMC traffic lookup."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rec(launches, time_s, flops, bytes_alg, k3=0):
    return dict(launches=launches, time_s=time_s, flops=flops, bytes_alg=bytes_alg, k3_launches=k3,
                k3_time_s=time_s / 2 if k3 else 0.0, k3_flops=flops / 2 if k3 else 0.0)


def test_roofline_object_fields():
    per_kernel = {"k_conv_h2": dict(_rec(1120, 0.27, 4.7e13, 8.1e11, k3=600), bytes_min=2.7e11),
                  "k_conv_mfma": _rec(40, 0.0066, 3.1e11, 1.1e10),
                  "k_split_rows": _rec(410, 0.0103, 0.0, 3.7e10)}
    classes = {("k3 C=64", "k_conv_h2"): dict(launches=60, time_s=0.05, flops=1.0e13, bytes_alg=2.0e11),
               ("k3 C=256", "k_conv_dma"): dict(launches=60, time_s=0.04, flops=2.0e13, bytes_alg=0.6e11)}
    r = bench.roofline_object(per_kernel, steps=10, classes=classes)
    # the headline entry is the LAYER CLASS with the most time; both roofs are reported, `bound` is the one it sits closer to
    assert r["class"] == "k3 C=64" and r["kernel"] == "k_conv_h2" and r["dominant_by"].startswith("layer class")
    hbm_frac = 2.0e11 / 0.05 / 1e9 / 8000.0
    mfma_frac = 3 * 1.0e13 / 0.05 / 1e12 / 2500.0
    assert abs(r["alg_frac_of_hbm_peak"] - hbm_frac) < 1e-3 and abs(r["mfma_frac_of_peak"] - mfma_frac) < 1e-3
    assert r["bound"] == ("hbm" if hbm_frac >= mfma_frac else "mfma") and abs(r["frac"] - max(hbm_frac, mfma_frac)) < 1e-3
    assert r["unit"] == ("GB/s" if r["bound"] == "hbm" else "TFLOP/s")
    assert r["launches_per_step"] == 6.0 and abs(r["avg_launch_us"] - 0.05 / 60 * 1e6) < 0.1
    assert r["alg_bytes_per_launch"] == 2.0e11 / 60 and "traffic" in r
    rows = {x["class"]: x for x in r["by_layer_class"]}
    assert rows["k3 C=256"]["bound"] == "mfma" and rows["k3 C=256"]["kernel"] == "k_conv_dma"
    assert rows["k3 C=64"]["bound"] == "hbm"
    # per kernel name beside it (compulsory bytes B_min of SURVEY.md 8(d) next to the algorithmic ones)
    k = {x["kernel"]: x for x in r["by_kernel"]}
    assert abs(k["k_conv_h2"]["min_frac_of_hbm_peak"] - 2.7e11 / 0.27 / 1e9 / 8000.0) < 1e-3
    assert k["k_conv_h2"]["min_bytes_per_launch"] == 2.7e11 / 1120 and k["k_conv_h2"]["launches_per_step"] == 112.0
    assert k["k_conv_mfma"]["bound"] == "mfma" and k["k_conv_mfma"]["peak"] == 157.3
    assert r["operand_split"]["kernel"] == "k_split_rows" and r["operand_split"]["launches_per_step"] == 41.0
    assert abs(r["conv_ms_per_step"] - (0.27 + 0.0066) / 10 * 1e3) < 1e-3     # the split passes are not a conv kernel
    json.dumps(r)


def test_committed_pmc_file_serves_the_traffic_field():
    """`roofline.traffic` is a recorded PMC measurement: the newest profiles/r*_pmc_conv.json that lists the kernel,
    with the commit it was taken at."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_conv.json")))
    assert files
    for name in ("k_conv_h2", "k_conv_mfma", "k_split_rows"):
        rec = bench.pmc_record(name)
        assert rec is not None and rec["hbm_bytes_per_launch"] > 0 and rec["file"].startswith("profiles/")
        assert bench.pmc_traffic(name) == rec["hbm_bytes_per_launch"]
    assert bench.pmc_traffic("no_such_kernel") is None


def test_printed_line_stays_parseable():
    """Round 5's line was 20.5 KB and the driver could not parse it.  `compact_line` of a FULL-SIZE result (round 5's own,
    profiles/r5last_bench_full.json) must stay under LINE_LIMIT and keep the contract fields, the headline roofline and
    the CPU baseline; everything else lives in the detail file."""
    with open(os.path.join(ROOT, "profiles", "r5last_bench_full.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 16000
    line = bench.compact_line(full, os.path.join(ROOT, "bench_detail.json"))
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT < 8192 and "\n" not in text
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == full["value"] and line["config"]["workload"] == full["config"]["workload"]
    assert line["config"]["also_measured"]["in_flight_1_scenes_per_s"] == full["in_flight_1"]["value"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "alg_bytes_per_launch",
              "launches_per_step", "matrix_roof"):
        assert k in r, k
    assert "by_kernel" not in r and "by_layer_class" not in r
    assert r["frac"] == round(full["roofline"]["frac"], 4) and r["classes"][0][0] == full["roofline"]["class"]
    cb = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb)
    assert set(line["configs"]) == set(full["configs"]) and all(isinstance(v, float) for v in line["configs"].values())
    assert line["detail"] == "bench_detail.json"
    # a pathological result (very long strings everywhere) still fits: optional keys are dropped, never the contract's
    fat = json.loads(json.dumps(full))
    fat["configs"] = {f"row{i}": {"scenes_per_s": 1.0} for i in range(400)}
    fat["exchange"] = {"x": "y" * 5000}
    slim = bench.compact_line(fat)
    assert len(json.dumps(slim)) <= bench.LINE_LIMIT and "roofline" in slim and "cpu_baseline" in slim


def test_window_kernel_headline_sits_under_the_matrix_roof():
    """The LDS-window kernels serve SURVEY 8(d)'s algorithmic gather bytes on chip: algorithmic bytes / time can exceed the HBM
    peak, so the headline roof of such a class is the f16 matrix peak and the algorithmic figure is kept beside it."""
    classes = {("k3 C=64", "k_conv_wop2"): dict(launches=18, time_s=18 * 210e-6, flops=18 * 54.5e9, bytes_alg=18 * 1.847e9,
                                                bytes_min=18 * 0.234e9)}
    per_kernel = {"k_conv_wop2": dict(_rec(18, 18 * 210e-6, 18 * 54.5e9, 18 * 1.847e9), bytes_min=18 * 0.234e9)}
    r = bench.roofline_object(per_kernel, steps=1, classes=classes)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - 3 * 54.5e9 / 210e-6 / 1e12 / 2500.0) < 1e-3
    assert r["alg_roof"]["bound"] == "hbm" and r["alg_roof"]["frac"] > 1.0          # the quotient that is not an HBM number
    line = bench.compact_line({"metric": "m", "value": 1.0, "roofline": r, "config": {"workload": "w"}})
    assert line["roofline"]["bound"] == "mfma" and "alg_roof" in line["roofline"]
