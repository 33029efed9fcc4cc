"""IK plug-in type and the batched GPU solver that fills MinkIKSolver's role."""
from . import hip_ik_solver as _hip
from . import ik_solver_interface as _iface

IKSolver = _iface.IKSolver
HipIKSolver = _hip.HipIKSolver

__all__ = ["IKSolver", "HipIKSolver"]
