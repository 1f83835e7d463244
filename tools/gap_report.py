#!/usr/bin/env python3
"""GPU idle time between the kernels of the bench's timed steps, from a rocprofv3 --kernel-trace csv.

usage: python tools/gap_report.py <dir with *_kernel_trace.csv> [n_last_steps]
Prints, for the last steps of the run (one step = one Adam launch), the wall time per step, the sum of kernel durations
and the idle share, plus the largest gaps by (previous kernel -> next kernel).
"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if "adam" in r[2].lower() or "multi_tensor" in r[2].lower()]
if len(marks) < n_last + 1:
    marks = [i for i, r in enumerate(rows) if "trace_bwd" in r[2]]
seg = rows[marks[-n_last - 1] + 1: marks[-1] + 1]
wall = seg[-1][1] - rows[marks[-n_last - 1]][1]
busy = sum(e - s for s, e, _ in seg)
print(f"{n_last} steps: wall {wall / n_last / 1e3:.1f} us/step, kernels {busy / n_last / 1e3:.1f} us/step, idle {100 * (1 - busy / wall):.1f} %, {len(seg) / n_last:.1f} launches/step")
gaps = defaultdict(lambda: [0, 0])
prev = rows[marks[-n_last - 1]]
for r in seg:
    g = r[0] - prev[1]
    key = (prev[2][:40], r[2][:40])
    gaps[key][0] += max(g, 0); gaps[key][1] += 1
    prev = r
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {t / n_last / 1e3:7.1f} us/step  x{c / n_last:.1f}  {a} -> {b}")
per = defaultdict(lambda: [0, 0])
for s, e, k in seg:
    per[k[:60]][0] += e - s; per[k[:60]][1] += 1
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {t / n_last / 1e3:7.1f} us/step  x{c / n_last:.1f}  {k}")
