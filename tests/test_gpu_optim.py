"""``artist_amd.optim.Adam`` (one HIP kernel per parameter tensor, ``art_adam_step``) against ``torch.optim.Adam`` as ARTIST's
reconstructors use it (artist/optim/surface_reconstructor.py:452-455, :779): the same parameters after several steps, learning-rate
schedulers and ``state_dict`` round trips work as with torch's, and the edge lock equals the reference's
``lock_control_points_on_outer_edges`` followed by a plain step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _grads(shape, steps, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(shape, generator=g) * 10.0 ** float(torch.randint(-6, 2, (1,), generator=g))).to(DEV) for _ in range(steps)]


@pytest.mark.parametrize("shape,kw", [((125, 4, 10, 10, 3), dict(lr=1e-3)), ((7, 1, 6, 6, 3), dict(lr=5e-2, betas=(0.8, 0.9), eps=1e-6)),
                                      ((1001,), dict(lr=1e-2, weight_decay=0.1)), ((3, 5), dict(lr=1e-2, maximize=True))])
def test_adam_equals_torch_adam(shape, kw):
    from artist_amd.optim import Adam
    p0 = torch.randn(shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    a, b = p0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    ours, ref = Adam([a], **kw), torch.optim.Adam([b], foreach=False, fused=False, **kw)
    sched_a = torch.optim.lr_scheduler.ExponentialLR(ours, gamma=0.9)
    sched_b = torch.optim.lr_scheduler.ExponentialLR(ref, gamma=0.9)
    for k, g in enumerate(_grads(shape, 12, 2)):
        a.grad, b.grad = g.clone(), g.clone()
        ours.step(); ref.step()
        sched_a.step(); sched_b.step()
        # same rule, fp32 both (torch divides by bias_correction2_sqrt where the kernel multiplies by its reciprocal, and its
        # device code contracts a * b + c into one FMA): the updates agree to ~1e-6 of their size, the parameters - of
        # magnitude ~1 - to a couple of ULPs of the largest ones (emulated on the CPU: <= 2.4e-7 after 12 steps)
        diff = float((a.detach() - b.detach()).abs().max())
        assert diff <= 6e-7 * max(1.0, float(b.detach().abs().max())), (k, diff)
    sa, sb = ours.state_dict()["state"][0], ref.state_dict()["state"][0]
    assert int(sa["step"]) == int(sb["step"]) == 12
    for key in ("exp_avg", "exp_avg_sq"):          # (moments that nearly cancel: tolerance relative to the largest entry)
        torch.testing.assert_close(sa[key], sb[key], rtol=2e-6, atol=2e-6 * float(sb[key].abs().max()))
    # state_dict round trip: a fresh optimiser continues the same trajectory
    c = a.detach().clone().requires_grad_(True)
    again = Adam([c], **kw)
    import copy
    again.load_state_dict(copy.deepcopy(ours.state_dict()))      # (load_state_dict keeps the tensors it is given)
    g = _grads(shape, 1, 3)[0]
    a.grad, c.grad = g.clone(), g.clone()
    ours.step(); again.step()
    assert torch.equal(a.detach(), c.detach())


def test_adam_edge_lock_equals_the_reference_recipe():
    """surface_reconstructor.py:1155-1224 + :779: the u and v components of the gradient of every net's outer-edge control points are
    zeroed, then Adam steps."""
    from artist_amd.optim import Adam
    shape = (9, 4, 7, 5, 3)
    p0 = torch.randn(shape, generator=torch.Generator().manual_seed(4)).to(DEV)
    a, b = p0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    ours, ref = Adam([a], lr=1e-3, lock_outer_edges=True), Adam([b], lr=1e-3)
    for g in _grads(shape, 5, 5):
        a.grad = g.clone()
        locked = g.clone()
        for edge in (locked[:, :, 0], locked[:, :, -1], locked[:, :, :, 0], locked[:, :, :, -1]):
            edge[..., :2] = 0                                     # u and v of the edge control points; z stays free
        b.grad = locked
        ours.step(); ref.step()
        assert torch.equal(a.detach(), b.detach())
    assert torch.equal(a.detach()[:, :, 0, :, :2], p0[:, :, 0, :, :2]) and torch.equal(a.detach()[:, :, :, -1, :2], p0[:, :, :, -1, :2])
    assert not torch.equal(a.detach()[:, :, 0, :, 2], p0[:, :, 0, :, 2])
    assert not torch.equal(a.detach()[:, :, 1:-1, 1:-1], p0[:, :, 1:-1, 1:-1])


def test_adam_refuses_the_cpu():
    from artist_amd import ArtistHipError
    from artist_amd.optim import Adam
    p = torch.zeros(4, requires_grad=True)
    p.grad = torch.ones(4)
    with pytest.raises(ArtistHipError, match="no CPU fallback"):
        Adam([p]).step()
