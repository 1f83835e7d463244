"""Seeded sweep over small random scenes x launch geometries (GPU): the lean kernels' block / facet / packing / chunking
logic has many corners (blocks that are not whole trips, facets smaller than a wave, one-point blocks, sample chunks with
packed edge points, windows of a few KB); every case is checked against the oracle - bitmaps, ray counters, gradients."""
import os

import numpy as np
import pytest
import torch

import oracle
from test_gpu_parity import DEV, _random_scene, n, rel_l2

pytestmark = pytest.mark.gpu

KNOBS = [
    {},
    {"ARTIST_HIP_FWD_TILE_KB": "4"},
    {"ARTIST_HIP_FWD_TILE_KB": "8", "ARTIST_HIP_BWD_PACK": "128"},
    {"ARTIST_HIP_FWD_BLOCKS": "4096", "ARTIST_HIP_FWD_MINCHUNK": "1"},
    {"ARTIST_HIP_FWD_PBLOCK": "96", "ARTIST_HIP_BWD_PBLOCK": "200"},
    {"ARTIST_HIP_BWD_PACK": "0", "ARTIST_HIP_FWD_MULTIPASS": "1"},
    {"ARTIST_HIP_LEAN": "0"},
    {"ARTIST_HIP_PERSISTENT": "0", "ARTIST_HIP_FWD_TILE_KB": "20"},
]


def _case(seed):
    rng = np.random.default_rng(seed)
    F = int(rng.choice([1, 1, 2, 3, 4, 6]))
    M = int(rng.choice([1, 7, 63, 64, 65, 250, 777, 1024, 1300]))
    H = int(rng.integers(1, 6))
    R = int(rng.choice([1, 2, 7, 8, 9, 16, 33]))
    W = int(rng.choice([2, 17, 64, 96, 256]))
    Hh = int(rng.choice([2, 31, 64, 128]))
    if H * R * F * M > 400_000:
        M = max(1, 400_000 // (H * R * F))
    spread = float(rng.choice([2e-3, 6e-3, 2e-2]))
    return H, R, F, M, W, Hh, spread


@pytest.mark.parametrize("seed", range(64))
def test_random_scene_and_geometry(seed, monkeypatch):
    from artist_amd import trace_rays
    H, R, F, M, W, Hh, spread = _case(seed)
    P = F * M
    knobs = KNOBS[seed % len(KNOBS)]
    hint = M if (seed // len(KNOBS)) % 2 == 0 and F > 1 else 0
    o, nrm, inc, both, tix, c, pn, dims, res = _random_scene(H, R, P, W, Hh, seed=1000 + seed, spread=spread)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    bd = both.to(DEV)
    od, nd = o.to(DEV).requires_grad_(True), nrm.to(DEV).requires_grad_(True)
    flux, fac = trace_rays(od, nd, inc.to(DEV), bd[..., 0], bd[..., 1], tix.to(DEV), c.to(DEV), pn.to(DEV), dims.to(DEV),
                           ray_magnitude=0.7, extinction=0.05, reflectivity=0.9, resolution=res, points_per_facet=hint)
    o_flux, o_fac = oracle.trace_fwd(o.numpy(), nrm.numpy(), inc.numpy(), both[..., 0].numpy(), both[..., 1].numpy(),
                                     tix.numpy(), c.numpy(), pn.numpy(), dims.numpy(), res, 0.7, 0.05, 0.9)
    label = dict(seed=seed, H=H, R=R, F=F, M=M, res=res, hint=hint, knobs=knobs)
    np.testing.assert_array_equal(n(fac), o_fac, err_msg=str(label))
    scale = float(np.abs(o_flux).max()) + 1e-30
    np.testing.assert_allclose(n(flux), o_flux, rtol=0, atol=2e-3 * scale, err_msg=str(label))
    assert abs(float(flux.detach().sum()) - float(o_flux.sum())) <= 1e-5 * abs(float(o_flux.sum())) + 1e-6, label
    w = torch.rand(flux.shape, generator=torch.Generator().manual_seed(seed)).to(DEV)
    (flux * w).sum().backward()
    go, gn = oracle.trace_bwd(o.numpy(), nrm.numpy(), inc.numpy(), both[..., 0].numpy(), both[..., 1].numpy(), tix.numpy(),
                              c.numpy(), pn.numpy(), dims.numpy(), res, n(w), 0.7, 0.05, 0.9)
    assert np.isfinite(n(od.grad)).all() and np.isfinite(n(nd.grad)).all(), label
    if np.linalg.norm(go) > 0:
        assert rel_l2(n(od.grad), go) < 2e-5 and rel_l2(n(nd.grad), gn) < 2e-5, (label, rel_l2(n(od.grad), go), rel_l2(n(nd.grad), gn))
