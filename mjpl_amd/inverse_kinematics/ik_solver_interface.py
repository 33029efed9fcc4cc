"""The IK plug-in type (reference src/mjpl/inverse_kinematics/ik_solver_interface.py:7-28)."""
from __future__ import annotations

import abc

import numpy as np

from ..lie import SE3


class IKSolver(abc.ABC):
    @abc.abstractmethod
    def solve_ik(self, pose: SE3, site: str, q_init_guess: np.ndarray | None) -> list[np.ndarray]:
        """Joint configurations whose ``site`` frame is at ``pose`` (world frame); ``[]`` if none."""
