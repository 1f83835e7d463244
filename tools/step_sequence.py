#!/usr/bin/env python3
"""The kernel sequence of the bench's last timed fwd+bwd step, from a rocprofv3 --kernel-trace csv (which launches are glue?):
everything from the step's nurbs_fwd launch to its Adam launch.
usage: python tools/step_sequence.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"]) for r in csv.DictReader(open(f)))
adam = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[2]]
end = adam[-1]
start = max(i for i in range(end) if "nurbs_fwd" in rows[i][2])
prev = adam[-2] if len(adam) > 1 else start
t0 = rows[start][0]
print(f"previous step's Adam ended {(rows[start][0] - rows[prev][1]) / 1e3:.1f} us before this step's first kernel")
for s, e, k, g in rows[start:end + 1]:
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  grid {g:>10}  {k[:100]}")
