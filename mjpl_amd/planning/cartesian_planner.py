"""Cartesian path following (reference src/mjpl/planning/cartesian_planner.py:11-104): the host
loop around the batched IK solver.  Poses are densified so that neighbours are at most
``lin_threshold`` / ``ori_threshold`` apart; each pose is solved with the previous waypoint as
the seed; of the candidates that obey the constraints (and, if asked, the collision interval
from the previous waypoint) the one nearest to the previous waypoint is kept."""
from __future__ import annotations

import math

import numpy as np

from ..constraint.constraint_interface import Constraint
from ..constraint.utils import obeys_constraints
from ..inverse_kinematics.ik_solver_interface import IKSolver
from ..lie import SE3
from .utils import _valid_collision_interval


def _interpolate_poses(pose_from: SE3, pose_to: SE3, lin_threshold: float, ori_threshold: float) -> list[SE3]:
    """``pose_from`` ... ``pose_to`` inclusive, decoupled translation / rotation distances (:11-42)."""
    for name, value in (("lin_threshold", lin_threshold), ("ori_threshold", ori_threshold)):
        if value <= 0.0:
            raise ValueError(f"`{name}` must be > 0.0")
    tangent = pose_to.minus(pose_from)
    need = (math.ceil(float(np.linalg.norm(tangent[:3])) / lin_threshold),
            math.ceil(float(np.linalg.norm(tangent[3:])) / ori_threshold), 1)
    alphas = np.linspace(0, 1, max(need) + 1)
    return [pose_from.interpolate(pose_to, float(a)) for a in alphas]


def cartesian_plan(q_init: np.ndarray, poses: list[SE3], site: str, solver: IKSolver,
                   constraints: list[Constraint], collision_interval_check=None,
                   lin_threshold: float = 0.01, ori_threshold: float = 0.1) -> list[np.ndarray]:
    """Waypoints from ``q_init`` through IK solutions of the densified ``poses``; ``[]`` as soon as
    one pose has no admissible solution (:44-104)."""
    if not site:
        raise ValueError("`site` must be defined.")
    dense = list(poses[:1])
    for a, b in zip(poses[:-1], poses[1:]):
        dense += _interpolate_poses(a, b, lin_threshold, ori_threshold)[1:]

    def admissible(q, prev):
        if not obeys_constraints(q, constraints):
            return False
        return not collision_interval_check or _valid_collision_interval(prev, q, *collision_interval_check)

    path = [q_init]
    for pose in dense:
        prev = path[-1]
        good = [q for q in solver.solve_ik(pose, site, q_init_guess=prev) if admissible(q, prev)]
        if not good:
            return []
        path.append(good[int(np.argmin([np.linalg.norm(q - prev) for q in good]))])
    return path
