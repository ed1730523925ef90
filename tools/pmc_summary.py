"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) for the conv kernels.
usage: pmc_summary.py <fetch_counter_csv> <write_counter_csv> <out.json> [commit-id]
HBM bytes = FETCH_SIZE * 2 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md "HBM") + WRITE_SIZE,
both reported by rocprofv3 in KiB."""
import csv, json, sys
from collections import defaultdict


def per_kernel(path, counter):
    per = defaultdict(float)
    names = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            key = (r.get("Dispatch_Id"), r.get("Agent_Id"))
            per[key] += float(r["Counter_Value"])
            names[key] = r.get("Kernel_Name", "")
    out = {}
    for kern in ("k_conv_dma", "k_conv_lin", "k_conv_wide", "k_conv_grid", "k_conv_wop2", "k_conv_wop", "k_conv_win", "k_attn_feat", "k_attn_split", "k_conv_h2", "k_conv_f16x3", "k_conv_mfma", "k_conv_rl", "k_split_rows", "k_win_build"):
        vals = [v for k, v in per.items() if kern + "<" in names[k] or kern + "(" in names[k] or names[k].startswith("_Z") and kern in names[k]]
        if vals:
            out[kern] = (sum(vals) / len(vals), len(vals))
    return out


def per_shape(path, counter):
    """(kernel instantiation, grid, workgroup) -> (mean KiB per dispatch, dispatches): one row per distinct launch shape, i.e.
    per layer (the maps of a step have distinct row counts), so that a layer CLASS can be read off the committed summary."""
    per = defaultdict(float)
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            key = (r.get("Dispatch_Id"), r.get("Agent_Id"))
            per[key] += float(r["Counter_Value"])
            meta[key] = (r.get("Kernel_Name", "").replace("void ", "").split("(")[0], int(r.get("Grid_Size", 0)), int(r.get("Workgroup_Size", 0)))
    acc = defaultdict(list)
    for k, v in per.items():
        acc[meta[k]].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {"commit": sys.argv[4] if len(sys.argv) > 4 else None, "kernels": {}, "note": "FETCH_SIZE doubled (gfx950 128-B requests tallied at 64 B); Infinity-Cache hits are counted, "
                               "not excluded"}
for kern in fetch:
    if kern in write:
        f, n1 = fetch[kern]
        w, n2 = write[kern]
        res["kernels"][kern] = {"dispatches": [n1, n2], "FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w,
                                "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
fs, ws = per_shape(sys.argv[1], "FETCH_SIZE"), per_shape(sys.argv[2], "WRITE_SIZE")
shapes = []
for key in sorted(fs, key=lambda k: -(2.0 * fs[k][0] + ws.get(k, (0.0, 0))[0]) * fs[k][1]):
    if key not in ws:
        continue
    name, grid, wg = key
    shapes.append({"kernel": name, "workgroups": grid // max(wg, 1), "workgroup_size": wg, "dispatches": [fs[key][1], ws[key][1]],
                   "hbm_MB_per_launch": round((2.0 * fs[key][0] + ws[key][0]) * 1024.0 / 1e6, 2),
                   "fetch_MB": round(2.0 * fs[key][0] * 1024.0 / 1e6, 2), "write_MB": round(ws[key][0] * 1024.0 / 1e6, 2)})
res["by_launch_shape"] = shapes
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "by_launch_shape"}))
