"""The optimiser step of ARTIST's reconstruction epochs as one HIP kernel per parameter tensor.

``artist_amd.optim.Adam`` takes the arguments of ``torch.optim.Adam`` that the reference uses
(``artist/optim/surface_reconstructor.py:452-455``, ``kinematics_reconstructor.py``: a list of tensors and ``lr``; betas, eps,
weight decay and ``maximize`` are honoured as well), is a ``torch.optim.Optimizer`` - ``param_groups``, ``state_dict``, learning-rate
schedulers and ``zero_grad`` work as with torch's - and keeps torch's state layout (``step``, ``exp_avg``, ``exp_avg_sq``).
The update rule is ``torch.optim.adam._single_tensor_adam`` in fp32; tests compare it with torch's over several steps.
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = ["Adam"]


class Adam(torch.optim.Optimizer):
    """``torch.optim.Adam`` on the gfx950 kernel ``art_adam_step`` (no amsgrad / capturable / differentiable modes).

    ``lock_outer_edges`` (an addition): for parameters shaped ``[..., nu, nv, 3]`` treat the first two components of the gradient
    of every net's first / last row and column as zero, as ``SurfaceReconstructor.lock_control_points_on_outer_edges`` does
    before the reference's step (surface_reconstructor.py:1155-1224: the outline is kept, z stays free) - without a pass over
    the gradient."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 maximize: bool = False, lock_outer_edges: bool = False):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize,
                                      lock_outer_edges=lock_outer_edges))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda":
                    raise _lib.ArtistHipError(f"artist_amd.optim.Adam runs on the GPU only (parameter on {p.device}); there is no CPU fallback")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise ValueError("artist_amd.optim.Adam steps contiguous float32 parameters")
                grad = p.grad
                if grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                if grad.dtype != torch.float32 or not grad.is_contiguous():
                    grad = grad.float().contiguous()
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] = int(state["step"]) + 1
                nu = nv = 0
                if group["lock_outer_edges"]:
                    if p.dim() < 3 or p.shape[-1] != 3:
                        raise ValueError("lock_outer_edges needs parameters shaped [..., nu, nv, 3]")
                    nu, nv = int(p.shape[-3]), int(p.shape[-2])
                with torch.cuda.device(p.device):
                    rc = lib.art_adam_step(p.data_ptr(), grad.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                           p.numel(), float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                                           float(group["weight_decay"]), state["step"], 1 if group["maximize"] else 0, nu, nv,
                                           torch.cuda.current_stream(p.device).cuda_stream)
                _lib.check(rc, "art_adam_step")
        return loss
