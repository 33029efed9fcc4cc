from .cartesian_planner import cartesian_plan
from .parallel_rrt import EdgeValidator, HipEdgeValidator, ParallelBiRRT
from .rrt import RRT
from .tree import Node, Tree
from .utils import path_length, smooth_path

__all__ = ("RRT", "Node", "Tree", "path_length", "smooth_path", "ParallelBiRRT", "EdgeValidator",
           "HipEdgeValidator", "cartesian_plan")
