"""Multi-GPU pieces of the hot path: heliostat sharding and the bitmap / gradient reduce.

One process per GPU (``torch.distributed``; backend ``"nccl"`` is RCCL on ROCm and runs over xGMI
inside a node).  Heliostats are independent through the trace, so the path shards with NO
data-path collective; the only exchange is the reduce of the per-target flux bitmaps
(``[T,256,256]`` fp32 = 256 KB per target, latency-bound) and - when training - of the control-point
gradients, exactly the two collectives the reference issues
(tutorials/02_heliostat_raytracing_distributed_tutorial.py:185-190,
artist/optim/surface_reconstructor.py:767-777).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .sampling import RestrictedDistributedSampler


def owned_heliostats(n_heliostats: int, world_size: int, rank: int) -> list[int]:
    """Rows of the ``[H,...]`` tensors owned by ``rank``: heliostat ``i`` -> rank ``i mod min(H, world)``
    (RestrictedDistributedSampler with one sample per heliostat, artist/raytracing/sampling.py:129-146)."""
    return list(RestrictedDistributedSampler(n_heliostats, n_heliostats, world_size, rank))


def all_reduce_sum(tensor: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce; a no-op in a single-process run (same silent fallback as
    artist/util/env.py:70-84)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


def all_reduce_sum_async(tensor: torch.Tensor, group=None):
    """Start an in-place SUM all-reduce and return a ``wait()``-able handle (None in a single-process run), so that
    the 256 KB flux reduce travels over xGMI while the backward kernels run."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def reduce_flux_per_target(flux_local: torch.Tensor, target_idx_local: torch.Tensor, n_targets: int,
                           per_target_sum, group=None) -> torch.Tensor:
    """Local per-heliostat bitmaps -> field-wide per-target bitmaps: local segment sum
    (``per_target_sum`` = ``HeliostatRayTracer.get_bitmaps_per_target`` or ``ops.per_target_sum``)
    followed by one all-reduce.  Invariant (tests/test_distributed_gloo.py): the result equals the
    single-rank ``get_bitmaps_per_target`` of the whole field up to fp32 summation order."""
    return all_reduce_sum(per_target_sum(flux_local, target_idx_local, n_targets), group)
