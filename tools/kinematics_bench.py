"""Time art_rigid_body_fwd / bwd for fields of H heliostats (linear actuators, 4 evaluations) -> one JSON line each."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from artist_amd.kinematics import initial_orientation_offsets, rigid_body_orientations  # noqa: E402

dev = torch.device("cuda:0")
out = []
for H in (125, 1000, 2000, 16000):
    g = torch.Generator(device="cpu").manual_seed(H)
    pos = torch.cat([torch.rand((H, 2), generator=g) * 200 - 100, torch.rand((H, 1), generator=g) * 3, torch.ones((H, 1))], 1).to(dev)
    rot = (torch.randn((H, 4), generator=g) * 0.01).to(dev).requires_grad_(True)
    trans = (torch.randn((H, 9), generator=g) * 0.05).to(dev).requires_grad_(True)
    nonopt = torch.zeros((H, 7, 2))
    nonopt[:, 0], nonopt[:, 1, 1], nonopt[:, 2], nonopt[:, 3] = 1.0, 1.0, 0.0, 70000.0
    nonopt[:, 4], nonopt[:, 5], nonopt[:, 6] = 155000.0, 0.335, 0.338
    nonopt = nonopt.to(dev)
    opt = torch.tensor([[0.02, 1.0], [0.075, 0.078]]).repeat(H, 1, 1).to(dev).requires_grad_(True)
    inc = torch.nn.functional.normalize(torch.tensor([0.2, 0.9, -0.39, 0.0]), dim=0).repeat(H, 1).to(dev)
    aim = torch.tensor([0.0, -5.0, 50.0, 1.0]).repeat(H, 1).to(dev)
    off = initial_orientation_offsets(dev)[0]

    def fwd():
        return rigid_body_orientations(1, pos, rot, trans, nonopt, opt, off, inc, aim, None, 4, 1e-4)[0]

    w = torch.randn((H, 4, 4), device=dev)
    for _ in range(3):
        torch.autograd.grad(fwd(), (rot, trans, opt), grad_outputs=w)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    K = 50
    e[0].record()
    for _ in range(K):
        o = fwd()
    e[1].record()
    for _ in range(K):
        torch.autograd.grad(o, (rot, trans, opt), grad_outputs=w, retain_graph=True)
    e[2].record()
    torch.cuda.synchronize()
    rec = {"H": H, "fwd_ms": e[0].elapsed_time(e[1]) / K, "bwd_ms": e[1].elapsed_time(e[2]) / K,
           "finite": bool(torch.isfinite(o).all())}
    print(json.dumps(rec), flush=True)
