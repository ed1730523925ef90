#!/bin/bash
# GPU occupancy of the in-flight loop: rocprofv3 kernel trace of a short bench run (three scenes in flight), then
#   tools/busy_union.py over the 500 ms window with the most overlap, and the per-kernel average durations of that window next to the
#   one-at-a-time averages of profiles/<tag>_bench_kernel_stats.csv:    tools/inflight_trace.sh <out dir> <one-at-a-time stats csv>
set -u
exec </dev/null
out=$1; ref=$2
mkdir -p "$out"
raw=/tmp/inflight_raw; rm -rf "$raw"; mkdir -p "$raw"
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$raw" -o run -- \
    python bench.py --steps 48 --no-cpu-baseline --no-exact --no-configs --no-profile > "$out/bench.json" 2> "$out/bench.err"
f=$(find "$raw" -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { echo "no kernel trace"; tail -3 "$out/bench.err"; exit 1; }
python tools/busy_union.py "$f" 500 | tee "$out/busy_union.txt"
python - "$f" "$ref" <<'PY' | tee "$out/inflation.txt"
import csv, sys, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
s = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
e = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
names = [r["Kernel_Name"] for r in rows]
# the in-flight timed loop = the densest 1000 ms stretch by number of launches
o = np.argsort(s); s, e = s[o], e[o]; names = [names[i] for i in o]
j = np.searchsorted(s, s + int(1000e6))
i = int(np.argmax(j - np.arange(len(s))))
lo, hi = s[i], s[i] + int(1000e6)
inside = (s >= lo) & (e <= hi)
agg = collections.defaultdict(list)
for k in np.nonzero(inside)[0]:
    agg[names[k].split("(")[0]].append((e[k] - s[k]) / 1e3)
ref = {}
for r in csv.DictReader(open(sys.argv[2])):
    ref[r["Name"].split("(")[0]] = float(r["AverageNs"]) / 1e3
tot_in = sum(sum(v) for v in agg.values())
print(f"densest 1000 ms: {int(inside.sum())} launches, sum of durations {tot_in / 1e3:.1f} ms")
print(f"{'kernel':60s} {'launches':>8s} {'avg us in flight':>17s} {'avg us alone':>13s} {'ratio':>6s} {'ms in window':>13s}")
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
    a = ref.get(name)
    print(f"{name[:60]:60s} {len(v):8d} {np.mean(v):17.1f} {a if a is None else round(a, 1)!s:>13s} {'' if not a else format(np.mean(v) / a, '.2f'):>6s} {sum(v) / 1e3:13.1f}")
PY
rm -rf "$raw"
