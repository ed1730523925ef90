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
    per_kernel = {"k_conv_h2": _rec(1120, 0.27, 4.7e13, 8.1e11, k3=600),
                  "k_conv_mfma": _rec(40, 0.0066, 3.1e11, 1.1e10),
                  "k_split_rows": _rec(410, 0.0103, 0.0, 3.7e10)}
    r = bench.roofline_object(per_kernel, steps=10)
    assert r["kernel"] == "k_conv_h2" and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - 8.1e11 / 0.27 / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    assert r["launches_per_step"] == 112.0 and abs(r["avg_launch_us"] - 0.27 / 1120 * 1e6) < 0.01
    assert r["other_conv_kernel"]["kernel"] == "k_conv_mfma" and r["other_conv_kernel"]["bound"] == "mfma"
    assert r["other_conv_kernel"]["peak"] == 157.3
    assert r["operand_split"]["kernel"] == "k_split_rows" and r["operand_split"]["launches_per_step"] == 41.0
    assert abs(r["conv_ms_per_step"] - (0.27 + 0.0066) / 10 * 1e3) < 1e-3     # the split passes are not a conv kernel
    json.dumps(r)


def test_committed_pmc_file_serves_the_traffic_field():
    with open(os.path.join(ROOT, "profiles", "r1_pmc_conv.json")) as f:
        kernels = json.load(f)["kernels"]
    for name in ("k_conv_h2", "k_conv_mfma", "k_split_rows"):
        assert name in kernels and kernels[name]["hbm_bytes_per_launch"] > 0
        assert bench.pmc_traffic(name) == kernels[name]["hbm_bytes_per_launch"]
    assert bench.pmc_traffic("no_such_kernel") is None
