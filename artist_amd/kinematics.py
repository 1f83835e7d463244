"""Rigid-body kinematics of a heliostat group, on MI355X.

Drop-in for ``artist.field.kinematics_rigid_body.RigidBody`` (artist/field/kinematics_rigid_body.py:15-634) with
its ideal and linear actuators (artist/field/actuators_ideal.py, actuators_linear.py): same constructor, same
attributes (``active_*`` tables set by ``HeliostatGroup.activate_heliostats``, heliostat_group.py:273-315), same
two methods.  The whole chain - actuator geometry, the two joints' 4x4 transforms, the iterative alignment with
its field-wide stopping rule - runs as one small kernel per evaluation (``art_rigid_body_fwd``, no host round trip
for the stopping rule) instead of ~60 ATen ops per iteration, and its gradients w.r.t. the deviation parameters and the optimisable actuator parameters (what the
kinematics reconstructor learns) come from ``art_rigid_body_bwd``.
"""
from __future__ import annotations

import torch

from . import _lib
from .ops import _f32c, _require_cuda, _stream

__all__ = ["RigidBody", "Actuators", "rigid_body_orientations"]

ACTUATOR_ROWS_IDEAL = 4       # type, clockwise, min, max motor position (artist/util/indices.py)
ACTUATOR_ROWS_LINEAR = 7      # ... + increment, offset, pivot radius


class RigidBodyOrientations(torch.autograd.Function):
    """``mode`` 0: motor positions -> orientations; 1: incident ray directions + aim points -> orientations.
    Returns (orientations [H,4,4], motor_positions [H,2]); only ``orientations`` is differentiable - w.r.t. the
    deviation parameters, the optimisable actuator parameters and, in mode 0, the given motor positions (what
    ``AimPointOptimizer`` learns, artist/optim/aim_point_optimizer.py:384-405)."""

    @staticmethod
    def forward(ctx, mode, positions, rot_dev, trans_dev, act_nonopt, act_opt, offsets, incident, aim, motor_positions,
                max_iter, min_eps):
        dev = _require_cuda(positions, rot_dev, trans_dev, act_nonopt, offsets)
        positions, rot_dev, trans_dev, act_nonopt, offsets = (_f32c(t) for t in (positions, rot_dev, trans_dev,
                                                                                 act_nonopt, offsets))
        H = positions.shape[0]
        rows = act_nonopt.shape[1] if act_nonopt.dim() == 3 else -1
        if rows not in (ACTUATOR_ROWS_IDEAL, ACTUATOR_ROWS_LINEAR) or act_nonopt.shape != (H, rows, 2):
            raise ValueError("actuator_parameters_non_optimizable must be [H,4,2] (ideal) or [H,7,2] (linear)")
        if positions.shape != (H, 4) or rot_dev.shape != (H, 4) or trans_dev.shape != (H, 9):
            raise ValueError("positions [H,4], rotation deviations [H,4] and translation deviations [H,9] expected")
        linear = rows == ACTUATOR_ROWS_LINEAR
        if linear:
            if act_opt is None or act_opt.shape != (H, 2, 2):
                raise ValueError("linear actuators need actuator_parameters_optimizable of shape [H,2,2]")
            act_opt = _f32c(act_opt)
        else:
            act_opt = None
        offsets = offsets.reshape(4, 4)
        if mode == 1:
            incident, aim = _f32c(incident), _f32c(aim)
            if incident.shape != (H, 4) or aim.shape != (H, 4):
                raise ValueError("incident_ray_directions and aim_points must be [number_of_active_heliostats, 4]")
            motor = torch.empty((H, 2), dtype=torch.float32, device=dev)
        else:
            if motor_positions.shape != (H, 2):
                raise ValueError("motor_positions must be [number_of_active_heliostats, 2]")
            motor = _f32c(motor_positions).clone()
        orientations = torch.empty((H, 4, 4), dtype=torch.float32, device=dev)
        scratch = torch.empty((H + int(max_iter) + 1,), dtype=torch.float32, device=dev)
        evaluations = torch.zeros((1,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().art_rigid_body_fwd(
                mode, positions.data_ptr(), rot_dev.data_ptr(), trans_dev.data_ptr(), act_nonopt.data_ptr(), rows,
                act_opt.data_ptr() if linear else None, offsets.data_ptr(),
                incident.data_ptr() if mode == 1 else None, aim.data_ptr() if mode == 1 else None, H,
                int(max_iter), float(min_eps), motor.data_ptr(), orientations.data_ptr(), scratch.data_ptr(),
                evaluations.data_ptr(), _stream(dev))
        _lib.check(rc, "art_rigid_body_fwd")
        ctx.mode, ctx.rows = mode, rows
        ctx.save_for_backward(positions, rot_dev, trans_dev, act_nonopt, act_opt, offsets,
                              incident if mode == 1 else None, aim if mode == 1 else None, motor, evaluations)
        ctx.mark_non_differentiable(motor)
        return orientations, motor

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_orientations, _grad_motor):
        positions, rot_dev, trans_dev, act_nonopt, act_opt, offsets, incident, aim, motor, evaluations = ctx.saved_tensors
        dev = positions.device
        H = positions.shape[0]
        linear = ctx.rows == ACTUATOR_ROWS_LINEAR
        grad_orientations = _f32c(grad_orientations)
        grad_rot = torch.empty_like(rot_dev)
        grad_trans = torch.empty_like(trans_dev)
        grad_opt = torch.empty_like(act_opt) if linear else None
        grad_motor = torch.empty_like(motor) if ctx.mode == 0 and ctx.needs_input_grad[9] else None
        with torch.cuda.device(dev):
            rc = _lib.lib().art_rigid_body_bwd(
                ctx.mode, positions.data_ptr(), rot_dev.data_ptr(), trans_dev.data_ptr(), act_nonopt.data_ptr(), ctx.rows,
                act_opt.data_ptr() if linear else None, offsets.data_ptr(),
                incident.data_ptr() if ctx.mode == 1 else None, aim.data_ptr() if ctx.mode == 1 else None, H,
                motor.data_ptr(), evaluations.data_ptr(), grad_orientations.data_ptr(), grad_rot.data_ptr(),
                grad_trans.data_ptr(), grad_opt.data_ptr() if linear else None,
                grad_motor.data_ptr() if grad_motor is not None else None, _stream(dev))
        _lib.check(rc, "art_rigid_body_bwd")
        return None, None, grad_rot, grad_trans, None, grad_opt, None, None, None, grad_motor, None, None


def rigid_body_orientations(mode, positions, rot_dev, trans_dev, act_nonopt, act_opt, offsets, incident=None, aim=None,
                            motor_positions=None, max_iter=4, min_eps=1e-4):
    """Functional form of the two ``RigidBody`` methods; returns (orientations [H,4,4], motor positions [H,2])."""
    return RigidBodyOrientations.apply(mode, positions, rot_dev, trans_dev, act_nonopt, act_opt, offsets, incident, aim,
                                       motor_positions, max_iter, min_eps)


class Actuators(torch.nn.Module):
    """The actuator parameter tables of ``artist.field.actuators.Actuators`` (artist/field/actuators.py:5-75); the
    conversions between motor positions and joint angles happen inside the kinematics kernel."""

    def __init__(self, non_optimizable_parameters: torch.Tensor,
                 optimizable_parameters: torch.Tensor = torch.tensor([]), device: torch.device | None = None) -> None:
        super().__init__()
        self.non_optimizable_parameters = non_optimizable_parameters
        self.optimizable_parameters = optimizable_parameters
        self.active_non_optimizable_parameters = torch.empty_like(non_optimizable_parameters, device=device)
        self.active_optimizable_parameters = torch.empty_like(optimizable_parameters, device=device)


def initial_orientation_offsets(device) -> torch.Tensor:
    """``[1,4,4]`` rotation taking the sampled surface orientation (0,0,1) to the kinematics' south (0,-1,0):
    axis-angle (pi/2, 0, 0), i.e. ``rotate_e(acos 0)`` (kinematics_rigid_body.py:176-190, rotations.py:7-64)."""
    angle = torch.arccos(torch.zeros((), dtype=torch.float32))
    c, s = torch.cos(angle), torch.sin(angle)
    m = torch.eye(4, dtype=torch.float32)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m[None].to(device)


class RigidBody(torch.nn.Module):
    """Same constructor, attributes and methods as ``artist.field.kinematics_rigid_body.RigidBody``."""

    def __init__(self, number_of_heliostats: int, heliostat_positions: torch.Tensor, initial_orientations: torch.Tensor,
                 translation_deviation_parameters: torch.Tensor, rotation_deviation_parameters: torch.Tensor,
                 actuator_parameters_non_optimizable: torch.Tensor,
                 actuator_parameters_optimizable: torch.Tensor = torch.tensor([]),
                 device: torch.device | None = None) -> None:
        super().__init__()
        device = torch.device(device) if device is not None else heliostat_positions.device
        self.number_of_heliostats = number_of_heliostats
        self.heliostat_positions = heliostat_positions
        self.initial_orientations = initial_orientations
        self.motor_positions = torch.zeros((number_of_heliostats, 2), device=device)
        self.translation_deviation_parameters = translation_deviation_parameters
        self.rotation_deviation_parameters = rotation_deviation_parameters

        self.number_of_active_heliostats = 0
        self.active_heliostat_positions = torch.empty_like(heliostat_positions, device=device)
        self.active_initial_orientations = torch.empty_like(initial_orientations, device=device)
        self.active_translation_deviation_parameters = torch.empty_like(translation_deviation_parameters, device=device)
        self.active_rotation_deviation_parameters = torch.empty_like(rotation_deviation_parameters, device=device)
        self.active_motor_positions = torch.empty_like(self.motor_positions, device=device)

        rows = actuator_parameters_non_optimizable.shape[1]
        if rows not in (ACTUATOR_ROWS_IDEAL, ACTUATOR_ROWS_LINEAR):
            raise ValueError("actuator_parameters_non_optimizable must be [H,4,2] (ideal) or [H,7,2] (linear)")
        self.actuators = Actuators(non_optimizable_parameters=actuator_parameters_non_optimizable,
                                   optimizable_parameters=actuator_parameters_optimizable.to(device), device=device)
        self.kinematics_standard_orientation = torch.tensor([0.0, -1.0, 0.0, 0.0], device=device)
        self.initial_orientation_offsets = initial_orientation_offsets(device)
        self.homogeneous_origin = torch.tensor([0.0, 0.0, 0.0, 1.0], device=device)

    def activate_all(self) -> None:
        """One active copy of every heliostat (what ``activate_heliostats`` does with an all-ones mask)."""
        self.number_of_active_heliostats = self.number_of_heliostats
        self.active_heliostat_positions = self.heliostat_positions
        self.active_initial_orientations = self.initial_orientations
        self.active_translation_deviation_parameters = self.translation_deviation_parameters
        self.active_rotation_deviation_parameters = self.rotation_deviation_parameters
        self.active_motor_positions = self.motor_positions
        self.actuators.active_non_optimizable_parameters = self.actuators.non_optimizable_parameters
        self.actuators.active_optimizable_parameters = self.actuators.optimizable_parameters

    def _run(self, mode, incident=None, aim=None, motor_positions=None, max_iter=4, min_eps=1e-4):
        opt = self.actuators.active_optimizable_parameters
        return rigid_body_orientations(
            mode, self.active_heliostat_positions, self.active_rotation_deviation_parameters,
            self.active_translation_deviation_parameters, self.actuators.active_non_optimizable_parameters,
            opt if opt.numel() > 0 else None, self.initial_orientation_offsets, incident, aim, motor_positions,
            max_iter, min_eps)

    def motor_positions_to_orientations(self, motor_positions: torch.Tensor,
                                        device: torch.device | None = None) -> torch.Tensor:
        """kinematics_rigid_body.py:510-538."""
        return self._run(0, motor_positions=motor_positions)[0]

    def incident_ray_directions_to_orientations(self, incident_ray_directions: torch.Tensor, aim_points: torch.Tensor,
                                                device: torch.device | None = None, max_num_iterations: int = 4,
                                                min_eps: float = 0.0001) -> torch.Tensor:
        """kinematics_rigid_body.py:540-634; also sets ``active_motor_positions`` like the reference (:632)."""
        orientations, motor = self._run(1, incident=incident_ray_directions, aim=aim_points,
                                        max_iter=max_num_iterations, min_eps=min_eps)
        self.active_motor_positions = motor
        return orientations
