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
    """Holds the full ``[H,R,P]`` (u, e) distortion views of a light source
    (artist/raytracing/sampling.py:22-85)."""

    def __init__(self, light_source, number_of_points_per_heliostat: int, number_of_active_heliostats: int,
                 random_seed: int = 7) -> None:
        self.distortions_u, self.distortions_e = light_source.get_distortions(
            number_of_points=number_of_points_per_heliostat,
            number_of_active_heliostats=number_of_active_heliostats,
            random_seed=random_seed,
        )

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
