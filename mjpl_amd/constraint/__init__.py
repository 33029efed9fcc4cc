"""Constraint plug-in surface of the MI355X build: the ABC, the three concrete constraints and
the two composition helpers (obeys / apply).  Collision and pose validation run in
``libmjpl_hip.so``; joint limits are a NumPy box test."""
from . import collision_constraint as _cc
from . import constraint_interface as _ci
from . import joint_limit_constraint as _jl
from . import pose_constraint as _pc
from . import utils as _u

Constraint = _ci.Constraint
CollisionConstraint, CollisionRuleset = _cc.CollisionConstraint, _cc.CollisionRuleset
JointLimitConstraint = _jl.JointLimitConstraint
PoseConstraint = _pc.PoseConstraint
obeys_constraints, apply_constraints = _u.obeys_constraints, _u.apply_constraints

__all__ = ["Constraint", "CollisionConstraint", "CollisionRuleset", "JointLimitConstraint", "PoseConstraint",
           "obeys_constraints", "apply_constraints"]
