from .collision_constraint import CollisionConstraint, CollisionRuleset
from .constraint_interface import Constraint
from .joint_limit_constraint import JointLimitConstraint
from .pose_constraint import PoseConstraint
from .utils import apply_constraints, obeys_constraints

__all__ = ("CollisionConstraint", "CollisionRuleset", "Constraint", "JointLimitConstraint", "PoseConstraint",
           "apply_constraints", "obeys_constraints")
