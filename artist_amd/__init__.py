"""artist_amd - MI355X-native (gfx950) implementation of ARTIST's heliostat ray-tracing hot path.

Scope: NURBS surface points + normals, ray-mirror reflection, sun-shape
scattering, receiver-plane intersection and the bilinear scatter-add flux bitmap, forward and
backward, behind ARTIST's own call surface:

    artist_amd.HeliostatRayTracer   <-> artist.raytracing.heliostat_ray_tracer.HeliostatRayTracer
    artist_amd.NURBSSurfaces        <-> artist.nurbs.NURBSSurfaces
    artist_amd.crop_flux_distributions_around_center, get_center_of_mass, PixelLoss, KLDivergenceLoss, FocalSpotLoss
                                    <-> artist.flux.bitmap / artist.optim.loss (the per-epoch flux epilogue)
    artist_amd.RigidBody            <-> artist.field.kinematics_rigid_body.RigidBody (ideal + linear actuators)
    artist_amd.Scenario             <-> artist.scenario.scenario.Scenario (load_scenario_from_hdf5, index_mapping)
    artist_amd.optim.Adam           <-> torch.optim.Adam as the reconstructors use it (the epoch's optimiser step)

All arithmetic runs in hand-written HIP kernels (``artist_amd/csrc``) reached through the C ABI in
``include/artist_hip.h``; there is no CPU fallback.
"""
from ._lib import ArtistHipError, build, lib  # noqa: F401
from .nurbs import NURBSSurfaces, create_nurbs_evaluation_grid, create_planar_nurbs_control_points  # noqa: F401
from .flux import (FocalSpotLoss, KLDivergenceLoss, PixelLoss, bitmap_coordinates_to_target_coordinates,  # noqa: F401
                   crop_and_kl_loss, crop_and_pixel_loss, crop_flux_distributions_around_center, get_center_of_mass)
from .kinematics import Actuators, RigidBody  # noqa: F401
from .ops import align_surfaces, nurbs_surface_points_and_normals, per_target_sum, trace_rays  # noqa: F401
from . import optim  # noqa: F401
from .raytracing import HeliostatRayTracer  # noqa: F401
from .scenario import Scenario, open_scenario_file  # noqa: F401
from .sampling import DistortionsDataset, RestrictedDistributedSampler  # noqa: F401

__version__ = "0.1.0"
