"""PoseConstraint behind the reference's plug-in surface (reference
src/mjpl/constraint/pose_constraint.py:11-147), with the projection running on the MI355X.

Scalar ``valid_config`` / ``apply`` keep the reference's semantics (a projected copy or
``None``); ``valid_configs`` / ``apply_batch`` are the batched forms a frontier planner uses
(one lane per configuration in ``k_pose_apply``).  The constructor arguments, defaults and
``ValueError`` cases are the reference's (:18-55); ``reference_frame`` is an
:class:`mjpl_amd.lie.SE3` (the stand-in for ``mink.SE3``).
"""
from __future__ import annotations

import numpy as np

from .. import engine as _engine
from ..lie import SE3, SO3
from .constraint_interface import Constraint


class PoseConstraint(Constraint):
    projects = True  # apply() moves configurations: planners must take its steps one at a time

    def __init__(self, model, site: str, reference_frame: SE3,
                 x_translation: tuple[float, float] = (-np.inf, np.inf),
                 y_translation: tuple[float, float] = (-np.inf, np.inf),
                 z_translation: tuple[float, float] = (-np.inf, np.inf),
                 roll: tuple[float, float] = (-np.inf, np.inf),
                 pitch: tuple[float, float] = (-np.inf, np.inf),
                 yaw: tuple[float, float] = (-np.inf, np.inf),
                 tolerance: float = 0.001, q_step: float = 0.05,
                 engine: _engine.Engine | None = None, device: int = 0, max_iters: int = 1000) -> None:
        if tolerance < 0.0:
            raise ValueError("`tolerance` must be >= 0.")
        if q_step <= 0.0:
            raise ValueError("`q_step` must be > 0.")
        self.model = model
        self.C = np.array([x_translation, y_translation, z_translation, roll, pitch, yaw], dtype=np.float64)
        self.C_T_world = reference_frame.inverse()  # :63
        self.site = site
        self.site_id = model.site(site).id
        self.tolerance = tolerance
        self._q_step = float(q_step)
        self.engine = engine if engine is not None else _engine.Engine(model, device=device)
        self._proj = _engine.PoseProjector(self.engine, site, self.C_T_world.wxyz_xyz[:4],
                                           self.C_T_world.wxyz_xyz[4:], self.C, tolerance, q_step, max_iters)

    # the reference's examples assign q_step on a live constraint
    # (examples/franka_constrained_move_to_pose.py:72-75)
    @property
    def q_step(self) -> float:
        return self._q_step

    @q_step.setter
    def q_step(self, value: float) -> None:
        if value <= 0.0:
            raise ValueError("`q_step` must be > 0.")
        self._q_step = float(value)
        self._proj.set_q_step(self._q_step)

    # -- Constraint interface (scalar)
    def valid_config(self, q: np.ndarray) -> bool:
        return bool(self._proj.valid(np.asarray(q, dtype=np.float64)[None, :])[0])

    def apply(self, q_old: np.ndarray, q: np.ndarray) -> np.ndarray | None:
        out, ok, _ = self._proj.apply(np.asarray(q_old, dtype=np.float64)[None, :],
                                      np.asarray(q, dtype=np.float64)[None, :])
        return out[0] if ok[0] else None

    # -- batched forms
    def valid_configs(self, Q: np.ndarray) -> np.ndarray:
        return self._proj.valid(Q)

    def apply_batch(self, Q_old: np.ndarray, Q: np.ndarray):
        """Row-wise ``apply``: (projected rows, ok mask, projection steps)."""
        return self._proj.apply(Q_old, Q)

    def site_poses(self, Q: np.ndarray):
        """World poses of the constrained site for rows of qpos: (xpos [N, 3], xmat [N, 3, 3])."""
        _, xpos, xmat = self._proj.valid(Q, poses=True)
        return xpos, xmat

    def site_pose(self, q: np.ndarray) -> SE3:
        """utils.site_pose (src/mjpl/utils.py:60-75) for this constraint's site."""
        xpos, xmat = self.site_poses(np.asarray(q, dtype=np.float64)[None, :])
        return SE3.from_rotation_and_translation(SO3.from_matrix(xmat[0]), xpos[0])
