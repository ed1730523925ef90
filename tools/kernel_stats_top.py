"""Top kernels of a rocprofv3 kernel_stats.csv per step:  python tools/kernel_stats_top.py <csv> <steps> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"kernel time {tot / 1e6 / steps:.2f} ms per step in {sum(int(r['Calls']) for r in rows) / steps:.0f} launches")
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:n]:
    print(f"{int(r['TotalDurationNs']) / 1e6 / steps:7.3f} ms {int(r['Calls']) / steps:7.1f} calls  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:90]}")
