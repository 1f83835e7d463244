#!/usr/bin/env python3
"""Measured HBM bandwidth of this GPU: device-to-device copy and a read-only reduction over 4 GiB (SURVEY 8d asks
for the measured peak next to the nominal 8 TB/s)."""
import json
import torch

dev = torch.device("cuda:0")
n = 1 << 30                       # 4 GiB of fp32
src = torch.ones(n, dtype=torch.float32, device=dev)
dst = torch.empty_like(src)


def timed(fn, steps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


ms_copy = timed(lambda: dst.copy_(src))
ms_read = timed(lambda: src.sum())
ms_fill = timed(lambda: dst.fill_(1.0))
print(json.dumps({"bytes": 4 * n, "copy_GBps": 2 * 4 * n / ms_copy / 1e6, "read_GBps": 4 * n / ms_read / 1e6,
                  "write_GBps": 4 * n / ms_fill / 1e6, "device": torch.cuda.get_device_name(0)}))
