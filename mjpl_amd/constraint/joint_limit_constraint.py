"""Joint-range box test (reference src/mjpl/constraint/joint_limit_constraint.py:10-23).

Host-side only: SURVEY.md section 2 row 7 keeps it off the GPU (trivial; it is the
pre-filter the batched planners apply with numpy before a launch).
"""
from __future__ import annotations

import numpy as np

from .constraint_interface import Constraint


class JointLimitConstraint(Constraint):
    projects = False  # apply() never moves a configuration: batched extension is allowed

    def __init__(self, model) -> None:
        rng = np.asarray(model.jnt_range, dtype=np.float64)
        self.lower, self.upper = rng[:, 0].copy(), rng[:, 1].copy()

    def valid_config(self, q: np.ndarray) -> bool:
        return bool(np.all(np.logical_and(self.lower <= q, q <= self.upper)))

    def apply(self, q_old: np.ndarray, q: np.ndarray) -> np.ndarray | None:
        return q if self.valid_config(q) else None

    def valid_configs(self, Q: np.ndarray) -> np.ndarray:
        """Row-wise test of full-nq configurations [N, nq] -> bool [N]."""
        Q = np.asarray(Q)
        return np.logical_and(self.lower <= Q, Q <= self.upper).all(axis=1)
