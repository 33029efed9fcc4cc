"""mjpl_amd -- MI355X-native batched RRT collision validation behind mjpl's Constraint /
planner plug-in surface.  The compute path is ``libmjpl_hip.so``
(mjpl_amd/csrc, C ABI in include/mjpl_hip.h); there is no CPU fallback in this package.
"""
from .constraint import (CollisionConstraint, CollisionRuleset, Constraint, JointLimitConstraint,
                         PoseConstraint, apply_constraints, obeys_constraints)
from .inverse_kinematics import HipIKSolver, IKSolver
from .lie import SE3, SO3
from .model import Model, ModelBuilder, load_mjcf, parse_mjcf
from .planning import (RRT, DeviceBiRRT, EdgeValidator, HipEdgeValidator, Node, ParallelBiRRT, Tree,
                       cartesian_plan, path_length, smooth_path)
from .utils import all_joints, qpos_idx, qvel_idx, random_config, site_pose


def comm_unique_id() -> bytes:
    """The 128-byte RCCL id rank 0 hands to the other ranks (``DeviceBiRRT(comm=(id, rank, world))``)."""
    from .engine import comm_unique_id as _f
    return _f()

__all__ = (
    "CollisionConstraint", "CollisionRuleset", "Constraint", "JointLimitConstraint", "PoseConstraint",
    "SE3", "SO3", "site_pose", "HipIKSolver", "IKSolver", "cartesian_plan",
    "apply_constraints", "obeys_constraints", "Model", "ModelBuilder", "load_mjcf", "parse_mjcf",
    "RRT", "Node", "Tree", "path_length", "smooth_path", "ParallelBiRRT", "DeviceBiRRT", "EdgeValidator", "HipEdgeValidator",
    "all_joints", "qpos_idx", "qvel_idx", "random_config", "comm_unique_id",
)
