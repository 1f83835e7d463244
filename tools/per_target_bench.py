#!/usr/bin/env python3
"""ops.per_target_sum alone (get_bitmaps_per_target, heliostat_ray_tracer.py:563-608): us per call, back-to-back launches."""
import json, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd import ops
dev = torch.device("cuda:0")
out = {}
for H in (125, 1000):
    flux = torch.rand(H, 256, 256, device=dev)
    tix = torch.zeros(H, dtype=torch.int32, device=dev)
    for _ in range(5): ops.per_target_sum(flux, tix, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): r = ops.per_target_sum(flux, tix, 1)
    e1.record(); torch.cuda.synchronize()
    out[f"H{H}"] = {"us": e0.elapsed_time(e1) / 50 * 1e3, "checksum": float(r.double().sum())}
print(json.dumps(out))
