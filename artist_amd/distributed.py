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

import os

import torch
import torch.distributed as dist

from .sampling import RestrictedDistributedSampler


def owned_heliostats(n_heliostats: int, world_size: int, rank: int) -> list[int]:
    """Rows of the ``[H,...]`` tensors owned by ``rank``: heliostat ``i`` -> rank ``i mod min(H, world)``
    (RestrictedDistributedSampler with one sample per heliostat, artist/raytracing/sampling.py:129-146)."""
    return list(RestrictedDistributedSampler(n_heliostats, n_heliostats, world_size, rank))


def _exchanging(group=None) -> bool:
    """Is there anybody to exchange with?  ``ARTIST_AMD_COLLECTIVES_AT_WORLD_1=1`` makes a one-rank process group issue its
    collectives too - a rehearsal of the RCCL calls (streams, async handles, dtypes) on a box with a single GPU."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("ARTIST_AMD_COLLECTIVES_AT_WORLD_1", "0") == "1"


def all_reduce_sum(tensor: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce; a no-op in a single-process run (same silent fallback as
    artist/util/env.py:70-84)."""
    if _exchanging(group):
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


def all_reduce_sum_async(tensor: torch.Tensor, group=None):
    """Start an in-place SUM all-reduce and return a ``wait()``-able handle (None in a single-process run), so that
    the 256 KB flux reduce travels over xGMI while the backward kernels run."""
    if _exchanging(group):
        return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


class _AllReduceSum(torch.autograd.Function):
    """SUM all-reduce inside the autograd graph; see ``all_reduce_sum_autograd``."""

    @staticmethod
    def forward(ctx, tensor, group, backward):
        ctx.group, ctx.backward_mode = group, backward
        out = tensor.detach().clone(memory_format=torch.contiguous_format)
        if _exchanging(group):
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.backward_mode == "local" or not _exchanging(ctx.group):
            return grad_out, None, None
        grad = grad_out.contiguous().clone()
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=ctx.group)
        return grad, None, None


def all_reduce_sum_autograd(tensor: torch.Tensor, group=None, backward: str = "sum") -> torch.Tensor:
    """Differentiable SUM all-reduce of the receiver flux: what ``torch.distributed.nn.functional.all_reduce`` does in
    ``AimPointOptimizer`` (artist/optim/aim_point_optimizer.py:515-519) - the total flux of all ranks stays in the autograd
    graph, so a loss on the REDUCED bitmap (flux integral, local flux density, KL against the target distribution) reaches
    every rank's heliostats.  Not in place: the input keeps its value.  A no-op in a single-process run.

    ``backward="sum"`` (default) is the reference's: the backward pass all-reduces the incoming gradient, so when every
    rank evaluates the same loss on the same reduced flux - as the reference's optimiser does - a rank's parameters
    receive ``world_size`` x the single-process gradient (Adam, which the reference uses, does not see the factor).
    ``backward="local"`` passes the rank's own upstream gradient through (no second collective): exactly the
    single-process gradient in that situation."""
    if backward not in ("sum", "local"):
        raise ValueError("backward must be 'sum' (torch.distributed.nn.functional.all_reduce) or 'local'")
    return _AllReduceSum.apply(tensor, group, backward)


class _PendingRows:
    """Handle of ``gather_owned_rows(..., async_op=True)``: ``wait()`` returns the gathered tensor."""

    def __init__(self, work, finish):
        self._work, self._finish = work, finish

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
        return self._finish()


def gather_owned_rows(local_rows: torch.Tensor, n_total: int, group=None, async_op: bool = False):
    """Rows owned by each rank (heliostat ``i`` -> rank ``i mod N``) -> the whole ``[n_total, ...]`` tensor on every rank.

    This is what the reference's ``all_reduce(SUM)`` of the control-point gradients produces
    (artist/optim/surface_reconstructor.py:767-777) when every heliostat lives on exactly one rank: the summands are
    row-disjoint, so the sum IS a gather - with (N-1)/N of the tensor on the wire instead of 2(N-1)/N, and no zero fill.
    Ragged shards (``n_total`` not a multiple of the world size) fall back to that all-reduce.
    ``async_op=True`` returns a handle whose ``wait()`` gives the tensor: the exchange then travels while the caller's
    optimiser step (which needs the own rows only) runs."""
    if not _exchanging(group):
        return _PendingRows(None, lambda: local_rows) if async_op else local_rows
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if n_total % world != 0:
        full = local_rows.new_zeros((n_total,) + tuple(local_rows.shape[1:]))
        full[rank::world] = local_rows
        if async_op:
            return _PendingRows(dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group, async_op=True), lambda: full)
        return all_reduce_sum(full, group)
    rows = local_rows.shape[0]
    gathered = local_rows.new_empty((world * rows,) + tuple(local_rows.shape[1:]))     # rank-major concatenation
    work = dist.all_gather_into_tensor(gathered, local_rows.contiguous(), group=group, async_op=async_op)

    def finish():
        # row r * rows + j of `gathered` is heliostat j * world + r
        return gathered.reshape((world, rows) + tuple(local_rows.shape[1:])).transpose(0, 1).reshape(
            (n_total,) + tuple(local_rows.shape[1:]))
    return _PendingRows(work, finish) if async_op else finish()


def reduce_flux_per_target(flux_local: torch.Tensor, target_idx_local: torch.Tensor, n_targets: int,
                           per_target_sum, group=None) -> torch.Tensor:
    """Local per-heliostat bitmaps -> field-wide per-target bitmaps: local segment sum
    (``per_target_sum`` = ``HeliostatRayTracer.get_bitmaps_per_target`` or ``ops.per_target_sum``)
    followed by one all-reduce.  Invariant (tests/test_distributed_gloo.py): the result equals the
    single-rank ``get_bitmaps_per_target`` of the whole field up to fp32 summation order."""
    return all_reduce_sum(per_target_sum(flux_local, target_idx_local, n_targets), group)
