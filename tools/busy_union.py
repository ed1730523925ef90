"""GPU occupancy of a bench run from a rocprofv3 kernel trace: in the window of the timed steps with several scenes in
flight (the `window_ms`-long window with the most time at >= 2 kernels running), the time with >= 1 kernel running, with
>= 2 running, and the sum of kernel durations.  python tools/busy_union.py <kernel_trace.csv> <window_ms>"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6
s = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
e = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
t = np.concatenate([s, e])
d = np.concatenate([np.ones_like(s), -np.ones_like(e)])
o = np.argsort(t, kind="stable")
t, d = t[o], d[o]
depth = np.cumsum(d)[:-1]
dt = np.diff(t)
b1 = np.concatenate([[0], np.cumsum(dt * (depth >= 1))])
b2 = np.concatenate([[0], np.cumsum(dt * (depth >= 2))])
j = np.searchsorted(t, t + win, side="right") - 1
gain = b2[j] - b2
i = int(np.argmax(gain))
lo, hi = t[i], t[j[i]]
inside = (s >= lo) & (e <= hi)
span = hi - lo
print(f"window {span / 1e6:.1f} ms: kernels {int(inside.sum())}  sum of durations {(e - s)[inside].sum() / 1e6:.1f} ms  "
      f">=1 running {(b1[j[i]] - b1[i]) / 1e6:.1f} ms ({(b1[j[i]] - b1[i]) / span:.3f})  "
      f">=2 running {(b2[j[i]] - b2[i]) / 1e6:.1f} ms ({(b2[j[i]] - b2[i]) / span:.3f})")
