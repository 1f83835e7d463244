"""CPU suite: the N>1 path with real processes (gloo, world_size 2 and 3).

What is under test is the product's sharding + reduce logic (`artist_amd.distributed`,
`RestrictedDistributedSampler`): which heliostat rows a rank owns, that no row is replicated or lost, and
that local per-target sums followed by ONE all-reduce reproduce the single-rank per-target bitmaps.  The
per-rank flux itself comes from the CPU oracle here (no GPU in this container); on the GPU box
tests/test_gpu_parity.py::test_full_size_properties checks the same invariant with the HIP kernels.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_dir):
    import oracle
    from artist_amd import RestrictedDistributedSampler
    from artist_amd.distributed import all_reduce_sum, gather_owned_rows, owned_heliostats, reduce_flux_per_target
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = dict(np.load(GOLDEN / f"{case}.npz"))
        H = d["aligned_points"].shape[0]
        n_distinct = int((d["active_mask"] > 0).sum())
        own = list(RestrictedDistributedSampler(H, n_distinct, world, rank))
        if n_distinct == H:
            assert own == owned_heliostats(H, world, rank)
        T = d["target_centers"].shape[0]
        res = d["resolution"]
        if own:
            flux, fac = oracle.trace_fwd(
                d["aligned_points"][own], d["aligned_normals"][own], d["incident"][own], d["distortions_u"][own],
                d["distortions_e"][own], d["target_idx"][own], d["target_centers"], d["target_normals"],
                d["target_dims"], res, float(d["ray_magnitude"]), float(d["extinction"]), float(d["reflectivity"]))
        else:
            flux = np.zeros((0, int(res[1]), int(res[0])), np.float32)
        # rows a rank owns are bit-identical to the same rows of the single-rank result
        full, _ = oracle.trace_fwd(
            d["aligned_points"], d["aligned_normals"], d["incident"], d["distortions_u"], d["distortions_e"],
            d["target_idx"], d["target_centers"], d["target_normals"], d["target_dims"], res,
            float(d["ray_magnitude"]), float(d["extinction"]), float(d["reflectivity"]))
        assert np.array_equal(flux, full[own])

        def per_target_sum(f, tix, n_targets):
            return torch.from_numpy(oracle.per_target(f.numpy(), tix.numpy(), n_targets))

        reduced = reduce_flux_per_target(torch.from_numpy(flux), torch.from_numpy(d["target_idx"][own].astype(np.int32)),
                                         T, per_target_sum)
        expect = oracle.per_target(full, d["target_idx"], T)
        np.testing.assert_allclose(reduced.numpy(), expect, rtol=1e-6, atol=1e-6 * float(expect.max() + 1))
        # ownership is a partition
        counts = torch.zeros(H)
        counts[own] = 1
        all_reduce_sum(counts)
        assert bool((counts == 1).all())
        # gradient exchange pattern (surface_reconstructor.py:767-777): row-disjoint grads, SUM == gather
        g = torch.zeros(H, 3)
        g[own] = torch.arange(H, dtype=torch.float32)[own, None] + 1
        all_reduce_sum(g)
        assert torch.equal(g, (torch.arange(H, dtype=torch.float32) + 1)[:, None].expand(H, 3))
        # ... which gather_owned_rows exploits (even shards: all-gather; ragged: the all-reduce above)
        for n_rows in (world * 4, world * 4 + 1):
            want = (torch.arange(n_rows, dtype=torch.float32) + 1)[:, None, None].expand(n_rows, 2, 3)
            got = gather_owned_rows(want[owned_heliostats(n_rows, world, rank)].contiguous(), n_rows)
            assert torch.equal(got, want), (n_rows, got[:, 0, 0])
            pending = gather_owned_rows(want[owned_heliostats(n_rows, world, rank)].contiguous(), n_rows, async_op=True)
            assert torch.equal(pending.wait(), want)                     # the overlapped form bench.py uses
        (out_dir / f"ok_{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "small_deg3"), (2, "mid_256"), (3, "small_deg2_tilted")])
def test_sharded_flux_reduces_to_single_rank(tmp_path, world, case):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, tmp_path), nprocs=world, join=True)
    assert sorted(p.name for p in tmp_path.iterdir()) == [f"ok_{r}" for r in range(world)]


def _autograd_worker(rank, world, port, out_dir):
    """A loss on the REDUCED flux (aim_point_optimizer.py:515-519 keeps the all-reduce in the graph): each rank's rows of
    the gradient against the single-process gradient."""
    from artist_amd.distributed import all_reduce_sum_autograd, owned_heliostats
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, npix = 7, 48
        gen = torch.Generator().manual_seed(3)
        bitmaps = torch.rand((H, npix), generator=gen, dtype=torch.float64)       # each heliostat's unit image
        target = torch.rand((npix,), generator=gen, dtype=torch.float64) * H / 2
        theta0 = torch.rand((H,), generator=gen, dtype=torch.float64) + 0.5

        def loss_of(total):                                                        # pixel loss + a flux-integral term
            return ((total - target) ** 2).sum() + 0.1 * total.sum() ** 2

        theta = theta0.clone().requires_grad_(True)
        loss_of((theta[:, None] * bitmaps).sum(0)).backward()
        single = theta.grad.clone()

        own = owned_heliostats(H, world, rank)
        for mode, factor in (("sum", float(world)), ("local", 1.0)):
            local = theta0[own].clone().requires_grad_(True)
            flux_local = (local[:, None] * bitmaps[own]).sum(0)
            before = flux_local.detach().clone()
            total = all_reduce_sum_autograd(flux_local, backward=mode)
            assert torch.equal(flux_local.detach(), before)                        # not in place
            torch.testing.assert_close(total.detach(), (theta0[:, None] * bitmaps).sum(0), rtol=1e-13, atol=0)
            loss_of(total).backward()
            torch.testing.assert_close(local.grad, factor * single[own], rtol=1e-12, atol=0)
        with pytest.raises(ValueError):
            all_reduce_sum_autograd(torch.zeros(3), backward="mean")
        (out_dir / f"ok_{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_differentiable_flux_all_reduce(tmp_path, world):
    port = _free_port()
    mp.spawn(_autograd_worker, args=(world, port, tmp_path), nprocs=world, join=True)
    assert sorted(p.name for p in tmp_path.iterdir()) == [f"ok_{r}" for r in range(world)]


def test_differentiable_flux_all_reduce_is_the_identity_in_one_process():
    from artist_amd.distributed import all_reduce_sum_autograd
    x = torch.arange(6.0, requires_grad=True)
    y = all_reduce_sum_autograd(x * 2.0)
    (y ** 2).sum().backward()
    assert torch.equal(y.detach(), torch.arange(6.0) * 2.0) and torch.equal(x.grad, 8.0 * torch.arange(6.0))
