#!/usr/bin/env python3
"""art_adam_step alone: back-to-back launches on one tensor, HIP events around the row (us per launch), for a few sizes and
with / without the edge lock; torch.optim.Adam (fused) beside it."""
import json, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from artist_amd.optim import Adam

dev = torch.device("cuda:0")


def row(make, n_launch=200):
    opt, p = make()
    for _ in range(20):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_launch):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_launch * 1e3


out = {}
for H in (125, 1000, 8000):
    shape = (H, 4, 10, 10, 3)
    def ours(lock):
        def make():
            p = torch.randn(shape, device=dev, requires_grad=True)
            p.grad = torch.randn_like(p)
            return Adam([p], lr=1e-4, lock_outer_edges=lock), p
        return make
    def theirs():
        p = torch.randn(shape, device=dev, requires_grad=True)
        p.grad = torch.randn_like(p)
        return torch.optim.Adam([p], lr=1e-4, fused=True), p
    out[f"H{H}"] = {"elements": int(torch.tensor(shape).prod()), "art_us": row(ours(False)), "art_lock_us": row(ours(True)), "torch_fused_us": row(theirs)}
print(json.dumps(out))
