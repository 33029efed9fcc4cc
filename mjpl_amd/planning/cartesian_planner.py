"""Cartesian path following (reference src/mjpl/planning/cartesian_planner.py:11-104): the host
loop around the batched IK solver -- interpolate the poses, solve IK seeded with the previous
waypoint, keep the candidates that obey the constraints (and the collision interval check), take
the one closest to the previous waypoint."""
from __future__ import annotations

import numpy as np

from ..constraint.constraint_interface import Constraint
from ..constraint.utils import obeys_constraints
from ..inverse_kinematics.ik_solver_interface import IKSolver
from ..lie import SE3
from .utils import _valid_collision_interval


def _interpolate_poses(pose_from: SE3, pose_to: SE3, lin_threshold: float, ori_threshold: float) -> list[SE3]:
    if lin_threshold <= 0.0:
        raise ValueError("`lin_threshold` must be > 0.0")
    if ori_threshold <= 0.0:
        raise ValueError("`ori_threshold` must be > 0.0")
    diff = pose_to.minus(pose_from)
    lin_steps = int(np.ceil(np.linalg.norm(diff[:3]) / lin_threshold))
    ori_steps = int(np.ceil(np.linalg.norm(diff[3:]) / ori_threshold))
    num_steps = max(lin_steps, ori_steps, 1)
    return [pose_from.interpolate(pose_to, alpha) for alpha in np.linspace(0, 1, num_steps + 1)]


def cartesian_plan(q_init: np.ndarray, poses: list[SE3], site: str, solver: IKSolver,
                   constraints: list[Constraint], collision_interval_check=None,
                   lin_threshold: float = 0.01, ori_threshold: float = 0.1) -> list[np.ndarray]:
    if not site:
        raise ValueError("`site` must be defined.")
    interpolated = [poses[0]]
    for i in range(len(poses) - 1):
        interpolated.extend(_interpolate_poses(poses[i], poses[i + 1], lin_threshold, ori_threshold)[1:])
    waypoints = [q_init]
    for p in interpolated:
        configs = [q for q in solver.solve_ik(p, site, q_init_guess=waypoints[-1])
                   if obeys_constraints(q, constraints)
                   and (not collision_interval_check
                        or _valid_collision_interval(waypoints[-1], q, *collision_interval_check))]
        if not configs:
            return []
        waypoints.append(min(configs, key=lambda q: np.linalg.norm(q - waypoints[-1])))
    return waypoints
