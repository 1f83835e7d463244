#!/usr/bin/env python3
"""The kernel sequence of the bench's last fwd+bwd step, from a rocprofv3 --kernel-trace csv (which launches are glue?).
usage: python tools/step_sequence.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
bw = [i for i, r in enumerate(rows) if "trace_bwd" in r[2]]
# the last timed fwd+bwd step: from after the previous step's last kernel (the one before this step's nurbs_fwd) to its Adam
last = bw[-6] if len(bw) > 6 else bw[-1]       # (the kernel_ms loops at the end launch trace_bwd alone)
start = max(i for i in range(last) if "nurbs_fwd" in rows[i][2])
end = next(i for i in range(last, len(rows)) if "multi_tensor" in rows[i][2] or i == len(rows) - 1)
t0 = rows[start][0]
for s, e, k in rows[start - 3:end + 2]:
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {k[:110]}")
