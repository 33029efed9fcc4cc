from .hip_ik_solver import HipIKSolver
from .ik_solver_interface import IKSolver

__all__ = ("HipIKSolver", "IKSolver")
