"""Planners on top of the batched constraints: the reference-shaped serial RRT, the
frontier-sharded bi-RRT (one batched launch per extension round), Cartesian path following and
the path utilities."""
from . import cartesian_planner as _cart
from . import parallel_rrt as _prrt
from . import rrt as _rrt
from . import tree as _tree
from . import utils as _utils

RRT = _rrt.RRT
ParallelBiRRT, EdgeValidator, HipEdgeValidator = _prrt.ParallelBiRRT, _prrt.EdgeValidator, _prrt.HipEdgeValidator
DeviceBiRRT = _prrt.DeviceBiRRT
Node, Tree = _tree.Node, _tree.Tree
cartesian_plan = _cart.cartesian_plan
smooth_path, path_length = _utils.smooth_path, _utils.path_length

__all__ = ["RRT", "DeviceBiRRT", "ParallelBiRRT", "EdgeValidator", "HipEdgeValidator", "Node", "Tree", "cartesian_plan",
           "smooth_path", "path_length"]
