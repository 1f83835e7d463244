"""Host-side partition of the distortion samples over ranks.

Mirrors ``artist/raytracing/sampling.py`` (DistortionsDataset :10-85,
RestrictedDistributedSampler :88-157): distinct heliostat ``i`` is owned by rank
``i mod min(n_distinct, world_size)`` together with its ``n_samples / n_distinct`` contiguous
replicas; ranks beyond the number of distinct heliostats stay idle - samples are never replicated.
"""
from __future__ import annotations

from typing import Iterator

import torch


class DistortionsDataset(torch.utils.data.Dataset):
    """The ``[H,R,P]`` (u, e) distortion views of a light source (artist/raytracing/sampling.py:22-85).

    ``rows`` (an extension for heliostat-sharded runs): keep only these heliostat samples - the ones a
    rank owns - so that a rank holds ``len(rows) / H`` of the 8 B/ray buffer instead of all of it.  The values are the
    SAME as the rows of the full seeded tensor (the rank-sharding invariant needs that), by one of two rules:

    * the light source offers ``get_distortions_rows(rows, ...)`` (``artist_amd.scene.Sun``): it draws row ``i`` from
      a stream of its own, and the unsharded ``get_distortions`` of that class is defined as all rows of that recipe;
    * ARTIST's own ``Sun`` on the CPU: ``torch.manual_seed(seed)`` once, then the ``MultivariateNormal`` is sampled
      heliostat by heliostat - rows of other ranks are drawn and dropped.  PyTorch's CPU normal generator works in
      blocks of 16 values, so this reproduces the one-shot ``sample((H,R,P))`` bit for bit whenever ``2 R P`` is a
      multiple of 16 (checked in tests/test_host_logic.py); otherwise, or for any other light source, the full tensor is
      sampled and sliced as before.
    """

    def __init__(self, light_source, number_of_points_per_heliostat: int, number_of_active_heliostats: int,
                 random_seed: int = 7, rows=None) -> None:
        self.rows = None if rows is None else [int(r) for r in rows]
        if self.rows is not None and len(self.rows) == number_of_active_heliostats and \
                self.rows == list(range(number_of_active_heliostats)):
            self.rows = None
        sliced = None
        if self.rows is not None:
            sliced = self._sample_rows(light_source, number_of_points_per_heliostat, number_of_active_heliostats, random_seed)
        if sliced is not None:
            self.distortions_u, self.distortions_e = sliced
            return
        self.distortions_u, self.distortions_e = light_source.get_distortions(
            number_of_points=number_of_points_per_heliostat,
            number_of_active_heliostats=number_of_active_heliostats,
            random_seed=random_seed,
        )
        if self.rows is not None:
            idx = torch.tensor(self.rows, dtype=torch.long, device=self.distortions_u.device)
            both = torch.stack((self.distortions_u.index_select(0, idx), self.distortions_e.index_select(0, idx)), dim=-1)
            self.distortions_u, self.distortions_e = both[..., 0], both[..., 1]      # one interleaved buffer again

    def _sample_rows(self, light_source, n_points, n_heliostats, seed):
        if not self.rows:
            # an idle rank (more ranks than active heliostats, sampling.py:107-157 `number_of_active_ranks`): no rows
            n_rays = int(getattr(light_source, "number_of_rays", 0) or 0)
            device = getattr(getattr(getattr(light_source, "distribution", None), "loc", None), "device", None)
            both = torch.empty((0, n_rays, int(n_points), 2), device=device)
            return both[..., 0], both[..., 1]
        if hasattr(light_source, "get_distortions_rows"):
            got = light_source.get_distortions_rows(self.rows, number_of_points=n_points,
                                                    number_of_active_heliostats=n_heliostats, random_seed=seed)
            if got is not None:
                return got
        dist = getattr(light_source, "distribution", None)
        n_rays = getattr(light_source, "number_of_rays", None)
        if not isinstance(dist, torch.distributions.MultivariateNormal) or n_rays is None or dist.loc.device.type != "cpu" \
                or (2 * int(n_rays) * int(n_points)) % 16 != 0:
            return None
        wanted = set(self.rows)
        kept = {}
        torch.manual_seed(seed)                                   # artist/scene/sun.py:224
        for h in range(max(wanted) + 1):
            row = dist.sample((1, int(n_rays), int(n_points)))
            if h in wanted:
                kept[h] = row
        both = torch.cat([kept[h] for h in self.rows]) if self.rows else torch.empty((0, int(n_rays), int(n_points), 2))
        u, e = both.permute(3, 0, 1, 2)                           # sun.py:229-233
        return u, e

    def __len__(self) -> int:
        return self.distortions_u.shape[0]

    def __getitem__(self, idx: int) -> tuple[torch.Tensor, torch.Tensor]:
        return self.distortions_u[idx], self.distortions_e[idx]


class RestrictedDistributedSampler(torch.utils.data.Sampler):
    """artist/raytracing/sampling.py:107-157."""

    def __init__(self, number_of_samples: int, number_of_active_heliostats: int, world_size: int = 1,
                 rank: int = 0) -> None:
        super().__init__()
        number_of_samples = int(number_of_samples)
        number_of_active_heliostats = int(number_of_active_heliostats)
        number_of_active_ranks = min(number_of_active_heliostats, world_size)
        self.rank_indices: list[int] = []
        if rank < number_of_active_ranks:
            per_heliostat = number_of_samples // number_of_active_heliostats
            for index in range(number_of_active_heliostats):
                if index % number_of_active_ranks == rank:
                    start = index * per_heliostat
                    self.rank_indices.extend(range(start, start + per_heliostat))

    def __iter__(self) -> Iterator[int]:
        return iter(self.rank_indices)

    def __len__(self) -> int:
        return len(self.rank_indices)
