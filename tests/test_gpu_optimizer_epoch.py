"""One REAL optimiser run of the reference, reproduced on the GPU (``-m gpu``).

``tests/golden/surface_reconstructor_epochs.npz`` holds three epochs of ARTIST's own ``SurfaceReconstructor.reconstruct_surfaces``
(artist/optim/surface_reconstructor.py:842-1152, unmodified; ``tests/golden/generate_golden.py::surface_reconstructor_epochs``: three
heliostats, one training sample each, flat model surfaces against flux measured from deflected ones, regulariser weights zero): per
epoch the control points at the start, the orientation matrices, the cropped predicted flux, the per-sample flux loss, the total loss,
the control-point gradient after ``_synchronize_and_lock_gradients``, the learning rate and the control points after
``optimizer.step()`` - plus the first epoch in fp64 on the same rays as the yardstick.  Here the same epoch is assembled from
``artist_amd``'s classes - NURBS evaluation + alignment, trace, crop, PixelLoss, the flux-integral constraint
(surface_reconstructor.py:600-655), backward, edge lock, Adam - and must land on the reference's numbers: "the optimiser code runs
untouched" as a parity statement, not a convergence property."""
import numpy as np
import pytest
import torch

from conftest import rel_l2, sun_distortions

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def t(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x), dtype=dtype).to(DEV)


def n(x):
    return x.detach().cpu().numpy()


class _Epoch:
    """The reference's epoch (surface_reconstructor.py:476-590, 967-1079) on artist_amd's classes."""

    def __init__(self, d):
        from artist_amd.scene import SolarTower, TowerTargetAreasPlanar
        self.d = d
        self.H = d["incident_train"].shape[0]
        self.P = 4 * d["eval_points"].shape[2]
        self.uv = t(d["eval_points"]).expand(self.H, -1, -1, -1)
        self.canting, self.transl = t(d["canting"]), t(d["facet_translations"])
        self.incident, self.tix = t(d["incident_train"]), t(d["target_idx_train"], torch.long)
        self.measured = t(d["flux_measured_train"])
        planar = TowerTargetAreasPlanar([f"t{i}" for i in range(d["target_centers"].shape[0])], t(d["target_centers"]),
                                        t(d["target_normals"]), t(d["target_dims"]))
        self.tower = SolarTower([planar], device=DEV)
        self.planar = planar
        # HeliostatRayTracer(random_seed=heliostat_group_rank = 0) samples [active samples, rays, points] on the CPU (sun.py:224-233)
        self.du, self.de = (x.to(DEV) for x in sun_distortions(self.H, int(d["n_rays"]), self.P, float(d["covariance"]), seed=int(d["seed"])))
        self.reference_integrals = None
        self.lam = torch.zeros(self.H, device=DEV)

    def loss(self, cp, orientation):
        from artist_amd import NURBSSurfaces, PixelLoss, crop_flux_distributions_around_center, ops
        d = self.d
        ap, an = NURBSSurfaces(torch.from_numpy(d["degrees"]), cp, device=DEV).calculate_surface_points_and_normals(
            self.uv, self.canting, self.transl, orientations=orientation)
        flux, _ = ops.trace_rays(ap.reshape(self.H, self.P, 4), an.reshape(self.H, self.P, 4), self.incident, self.du, self.de, self.tix,
                                 self.planar.centers, self.planar.normals, self.planar.dimensions, float(d["ray_magnitude"]),
                                 float(d["extinction"]), float(d["reflectivity"]), tuple(int(v) for v in d["resolution"]),
                                 points_per_facet=self.P // 4)
        cropped = crop_flux_distributions_around_center(flux, self.tower, self.tix)
        per_sample = PixelLoss()(cropped, self.measured, reduction_dimensions=(1, 2))
        per_heliostat = per_sample.view(self.H, int(d["number_of_train_samples"])).mean(dim=-1)
        # the Augmented-Lagrangian flux-integral constraint (surface_reconstructor.py:600-655)
        integrals = cropped.sum(dim=(1, 2))
        if self.reference_integrals is None:
            self.reference_integrals = integrals.detach()
        rel = (integrals - self.reference_integrals) / (self.reference_integrals + float(d["epsilon"]))
        violation = torch.clamp(-float(d["energy_tolerance"]) - rel, min=0.0).view(self.H, -1).mean(dim=-1)
        constraint = self.lam * violation + 0.5 * float(d["rho_flux_integral"]) * violation ** 2
        total = torch.mean(per_heliostat + constraint)
        return total, cropped, per_sample, violation

    def after_backward(self, violation):
        with torch.no_grad():                                            # surface_reconstructor.py:1041-1047
            self.lam = torch.clamp(self.lam + float(self.d["rho_flux_integral"]) * violation, min=0.0)


def _lock(grad):
    """lock_control_points_on_outer_edges (surface_reconstructor.py:1155-1224): u and v of the edge control points."""
    g = grad.clone()
    for edge in (g[:, :, 0], g[:, :, -1], g[:, :, :, 0], g[:, :, :, -1]):
        edge[..., :2] = 0
    return g


def test_each_epoch_of_the_reference_run_is_reproduced(golden):
    """Every epoch from the reference's own starting point: cropped flux at the trace-stage tolerance, losses, and the locked
    control-point gradient within the reference's own fp32-vs-fp64 distance."""
    d = golden("surface_reconstructor_epochs")
    ep = _Epoch(d)
    grad_yard = rel_l2(d["grad_locked"][0], d["grad_locked_f64_epoch0"])
    flux_yard = rel_l2(d["cropped_flux"][0], d["cropped_flux_f64_epoch0"])
    print(f"the reference's own fp32-vs-fp64 distance, first epoch: cropped flux {flux_yard:.2e}, locked gradient {grad_yard:.2e}")
    for e in range(d["cp_start"].shape[0]):
        cp = t(d["cp_start"][e]).requires_grad_(True)
        total, cropped, per_sample, violation = ep.loss(cp, t(d["orientation"][e]))
        total.backward()
        ep.after_backward(violation)
        flux_err = rel_l2(n(cropped), d["cropped_flux"][e])
        grad_err = rel_l2(n(_lock(cp.grad)), d["grad_locked"][e])
        print(f"epoch {e}: cropped flux {flux_err:.2e}, total loss {float(total):.7f} (reference {d['total_loss'][e]:.7f}), "
              f"locked gradient {grad_err:.2e}")
        assert flux_err < max(2e-5, 2 * flux_yard), (e, flux_err, flux_yard)
        np.testing.assert_allclose(n(per_sample), d["flux_loss_per_sample"][e], rtol=2e-5)
        np.testing.assert_allclose(float(total.detach()), d["total_loss"][e], rtol=2e-5)
        assert float(violation.abs().max()) == 0.0                       # (the constraint is inactive in this run, as in the reference's)
        assert grad_err < max(3 * grad_yard, 5e-4), (e, grad_err, grad_yard)
        if e == 0:       # ... and as close to the fp64 run as the reference's own fp32 run is
            assert rel_l2(n(_lock(cp.grad)), d["grad_locked_f64_epoch0"]) < max(3 * grad_yard, 5e-4)


@pytest.mark.parametrize("fused_lock", [False, True])
def test_the_whole_run_lands_on_the_reference_control_points(golden, fused_lock):
    """The three epochs chained - artist_amd.optim.Adam stepping (edge lock as a separate pass over the gradient, like the
    reference, or inside the Adam kernel), the reference's learning rates - against the control points the reference ended each epoch
    with.  Adam's first steps move every coordinate by ~lr whatever the gradient's size, so a coordinate whose gradient is almost zero
    may step the other way: compared in units of the learning rate."""
    from artist_amd.optim import Adam
    d = golden("surface_reconstructor_epochs")
    ep = _Epoch(d)
    cp = t(d["cp_start"][0]).requires_grad_(True)
    optimizer = Adam([cp], lr=float(d["lr"][0]), lock_outer_edges=fused_lock)
    for e in range(d["cp_start"].shape[0]):
        for group in optimizer.param_groups:
            group["lr"] = float(d["lr"][e])                              # (ExponentialLR's schedule, as recorded)
        optimizer.zero_grad()
        total, _, _, violation = ep.loss(cp, t(d["orientation"][e]))
        total.backward()
        ep.after_backward(violation)
        if not fused_lock:
            cp.grad = _lock(cp.grad)
        optimizer.step()
        moved = (n(cp) - d["cp_after"][e]) / float(d["lr"][e])
        np.testing.assert_allclose(float(total.detach()), d["total_loss"][e], rtol=1e-4)
        frac_close = float((np.abs(moved) < 0.05).mean())
        print(f"epoch {e}: total loss {float(total):.7f} (reference {d['total_loss'][e]:.7f}); control points within 0.05 lr of the "
              f"reference's: {100 * frac_close:.2f} %, largest difference {np.abs(moved).max():.3f} lr")
        assert frac_close > 0.97 and np.abs(moved).max() < 2.0 * (e + 1) + 0.1, (e, frac_close, np.abs(moved).max())
    # the edge control points kept their outline (u, v), z moved
    start = d["cp_start"][0]
    assert np.array_equal(n(cp)[:, :, 0, :, :2], start[:, :, 0, :, :2]) and np.array_equal(n(cp)[:, :, :, -1, :2], start[:, :, :, -1, :2])
    assert not np.array_equal(n(cp)[:, :, 0, :, 2], start[:, :, 0, :, 2])


# ---------------------------------------------------------------------------------------------------------------------------
# KinematicsReconstructor, flux-driven mode (artist/optim/kinematics_reconstructor.py:886-1065), on the reference's own scenario
# file: tests/golden/kinematics_reconstructor_epochs.npz (generate_golden.py::kinematics_reconstructor_epochs).
# ---------------------------------------------------------------------------------------------------------------------------
def _kinematics_epoch_setup(d):
    import pathlib

    from artist_amd.scenario import Scenario, open_scenario_file
    path = pathlib.Path(__file__).resolve().parent / "golden" / "scenarios" / "test_blocking.h5"
    with open_scenario_file(path) as scenario_file:
        scenario = Scenario.load_scenario_from_hdf5(scenario_file=scenario_file,
                                                    number_of_surface_points_per_facet=torch.tensor([int(v) for v in d["points_per_facet"]]),
                                                    device=DEV)
    scenario.set_number_of_rays(number_of_rays=int(d["n_rays"]))
    group = scenario.heliostat_field.heliostat_groups[0]
    # the training samples: the first calibration sample of every heliostat but heliostat_3 (the reference's train/test split)
    mask = torch.zeros(len(group.names), dtype=torch.int32, device=DEV)
    mask[torch.as_tensor(d["heliostats"], device=DEV)] = 1
    return scenario, group, mask


def _kinematics_loss(scenario, group, mask, d, rotation):
    """One epoch's loss (kinematics_reconstructor.py:535-622) on artist_amd's classes; ``rotation`` [N,4] is the learnable tensor."""
    from artist_amd import FocalSpotLoss, HeliostatRayTracer
    kin = group.kinematics
    kin.rotation_deviation_parameters = rotation
    incident, tix = t(d["incident_train"]), t(d["target_idx_train"], torch.long)
    group.activate_heliostats(active_heliostats_mask=mask, device=DEV)
    group.align_surfaces_with_motor_positions(motor_positions=t(d["motor_positions_train"]), active_heliostats_mask=mask, device=DEV)
    tracer = HeliostatRayTracer(scenario=scenario, heliostat_group=group, blocking_active=False, random_seed=int(d["seed"]),
                                bitmap_resolution=torch.tensor([int(v) for v in d["resolution"]]))
    # A light source on the device draws every heliostat sample from a Philox stream of its own (artist_amd/scene.py); the
    # reference's run sampled on the CPU (sun.py:224-234, seed = the group's rank): swap that sample in.
    law = tracer.light_source.distribution
    torch.manual_seed(int(d["seed"]))
    sample = torch.distributions.MultivariateNormal(law.loc.cpu(), law.covariance_matrix.cpu()).sample(
        (int(mask.sum()), int(d["n_rays"]), group.active_surface_points.shape[1]))
    tracer.distortions_dataset.distortions_u, tracer.distortions_dataset.distortions_e = sample.permute(3, 0, 1, 2)
    flux = tracer.trace_rays(incident_ray_directions=incident, active_heliostats_mask=mask, target_area_indices=tix, device=DEV)[0]
    per_sample = FocalSpotLoss(scenario=scenario)(prediction=flux, ground_truth=t(d["flux_measured_train"]), target_area_indices=tix,
                                                  reduction_dimensions=(1, 2), device=DEV)
    per_heliostat = per_sample.view(-1, 1).median(dim=1).values                 # one training sample per heliostat
    return per_heliostat.mean(), per_sample, flux


def test_each_epoch_of_the_reference_kinematics_run_is_reproduced(golden):
    """Three epochs of the reference's own ``KinematicsReconstructor`` (flux-driven: motor positions -> orientations with the
    learnable rotation deviations -> alignment -> trace -> FocalSpotLoss): flux, per-sample loss (a distance in metres), total loss
    and the gradient w.r.t. the rotation deviations - through art_trace_bwd, art_align_bwd and art_rigid_body_bwd - from the
    reference's starting point of each epoch, within the reference's own fp32-vs-fp64 distance."""
    d = golden("kinematics_reconstructor_epochs")
    scenario, group, mask = _kinematics_epoch_setup(d)
    grad_yard = rel_l2(d["grad"][0], d["grad_f64_epoch0"])
    flux_yard = rel_l2(d["flux_predicted"][0], d["flux_predicted_f64_epoch0"])
    loss_yard = float(np.abs(d["loss_per_sample"][0] - d["loss_per_sample_f64_epoch0"]).max())
    print(f"the reference's own fp32-vs-fp64 distance, first epoch: flux {flux_yard:.2e}, gradient {grad_yard:.2e}, loss {loss_yard:.2e} m")
    for e in range(d["rotation_start"].shape[0]):
        rotation = t(d["rotation_start"][e]).requires_grad_(True)
        total, per_sample, flux = _kinematics_loss(scenario, group, mask, d, rotation)
        total.backward()
        flux_err, grad_err = rel_l2(n(flux), d["flux_predicted"][e]), rel_l2(n(rotation.grad), d["grad"][e])
        print(f"epoch {e}: flux {flux_err:.2e}, total loss {float(total.detach()):.7f} m (reference {d['total_loss'][e]:.7f}), gradient {grad_err:.2e}")
        assert flux_err < max(3 * flux_yard, 2e-5), (e, flux_err, flux_yard)
        np.testing.assert_allclose(n(per_sample), d["loss_per_sample"][e], rtol=0, atol=max(3 * loss_yard, 2e-5))
        np.testing.assert_allclose(float(total.detach()), d["total_loss"][e], rtol=0, atol=max(3 * loss_yard, 2e-5))
        assert grad_err < max(3 * grad_yard, 5e-4), (e, grad_err, grad_yard)
        unused = np.setdiff1d(np.arange(rotation.shape[0]), d["heliostats"])
        assert float(rotation.grad[torch.as_tensor(unused, device=DEV)].abs().max()) == 0.0       # heliostat_3 is not traced


def test_the_whole_kinematics_run_lands_on_the_reference_parameters(golden):
    """... and the three epochs chained with ``artist_amd.optim.Adam`` at the reference's learning rates: the rotation deviations
    after every step, in units of the learning rate (Adam's first steps move every coordinate by ~lr)."""
    from artist_amd.optim import Adam
    d = golden("kinematics_reconstructor_epochs")
    scenario, group, mask = _kinematics_epoch_setup(d)
    rotation = t(d["rotation_start"][0]).requires_grad_(True)
    optimizer = Adam([rotation], lr=float(d["lr"][0]))
    used = torch.as_tensor(d["heliostats"], device=DEV)
    for e in range(d["rotation_start"].shape[0]):
        for pg in optimizer.param_groups:
            pg["lr"] = float(d["lr"][e])
        optimizer.zero_grad()
        total, _, _ = _kinematics_loss(scenario, group, mask, d, rotation)
        total.backward()
        optimizer.step()
        moved = (n(rotation) - d["rotation_after"][e]) / float(d["lr"][e])
        print(f"epoch {e}: total loss {float(total.detach()):.7f} m (reference {d['total_loss'][e]:.7f}); largest difference of the rotation "
              f"deviations {np.abs(moved).max():.4f} lr")
        np.testing.assert_allclose(float(total.detach()), d["total_loss"][e], rtol=0, atol=1e-4)
        # (a coordinate whose gradient is almost zero may step the other way: the traced heliostats' coordinates with a sizeable
        #  gradient must agree closely, the others within the two steps that a sign flip costs)
        big = np.abs(d["grad"][e]) > 1e-3 * np.abs(d["grad"][e]).max()
        assert np.abs(moved[big]).max() < 0.05, np.abs(moved[big]).max()
        assert np.abs(moved).max() < 2.0 * (e + 1) + 0.1
    assert not torch.equal(rotation.detach()[used], t(d["rotation_start"][0])[used])
