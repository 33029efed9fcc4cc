"""CBiRRT driver (reference src/mjpl/planning/rrt.py:17-237): same constructor, validation,
sampling stream and tree-swap logic, so that for a fixed seed it makes the same sequence of
constraint queries as the reference planner.
"""
from __future__ import annotations

import time

import numpy as np

from .. import utils as _utils
from ..constraint.constraint_interface import Constraint
from ..constraint.utils import obeys_constraints
from .tree import Node, Tree
from .utils import _combine_paths, _constrained_extend


class RRT:
    def __init__(self, model, planning_joints: list[str], constraints: list[Constraint],
                 collision_interval_check=None, max_planning_time: float = 10.0,
                 epsilon: float = 0.05, seed: int | None = None,
                 goal_biasing_probability: float = 0.05) -> None:
        if not planning_joints:
            raise ValueError("`planning_joints` cannot be empty.")
        if max_planning_time <= 0.0:
            raise ValueError("`max_planning_time` must be > 0.0")
        if epsilon <= 0.0:
            raise ValueError("`epsilon` must be > 0.0")
        if not 0.0 <= goal_biasing_probability <= 1.0:
            raise ValueError("`goal_biasing_probability` must be within [0.0, 1.0].")
        self.model = model
        self.planning_joints = planning_joints
        self.constraints = constraints
        self.collision_interval_check = collision_interval_check
        self.max_planning_time = max_planning_time
        self.epsilon = epsilon
        self.seed = seed
        self.goal_biasing_probability = goal_biasing_probability
        self.iterations = 0  # sample/extend rounds of the last plan (diagnostics)

    # -- pose goals need an IK solver, which is outside the hot path
    def plan_to_pose(self, q_init, pose, site: str, solver=None) -> list[np.ndarray]:
        return self.plan_to_poses(q_init, [pose], site, solver)

    def plan_to_poses(self, q_init, poses, site: str, solver=None) -> list[np.ndarray]:
        if solver is None:  # rrt.py:130-136 builds a MinkIKSolver with these arguments
            from ..inverse_kinematics import HipIKSolver
            solver = HipIKSolver(model=self.model, joints=self.planning_joints, constraints=self.constraints,
                                 seed=self.seed, max_attempts=5)
        goals = [q for p in poses for q in solver.solve_ik(p, site, q_init_guess=q_init)
                 if obeys_constraints(q, self.constraints)]
        return [] if not goals else self.plan_to_configs(q_init, goals)

    def plan_to_config(self, q_init: np.ndarray, q_goal: np.ndarray) -> list[np.ndarray]:
        return self.plan_to_configs(q_init, [q_goal])

    def plan_to_configs(self, q_init: np.ndarray, q_goals: list[np.ndarray]) -> list[np.ndarray]:
        if not obeys_constraints(q_init, self.constraints):
            raise ValueError("q_init is not a valid configuration")
        for q in q_goals:
            if not obeys_constraints(q, self.constraints):
                raise ValueError(f"The following goal config is not a valid configuration: {q}")

        q_idx = _utils.qpos_idx(self.model, self.planning_joints)
        fixed = [i for i in range(self.model.nq) if i not in q_idx]
        for q in q_goals:
            if not np.allclose(q_init[fixed], q[fixed], rtol=0, atol=1e-12):
                raise ValueError(
                    f"The following goal config has values for joints outside of the planner's "
                    f"planning joints that don't match q_init: {q}. q_init is {q_init}, and the "
                    f"planning joints are {self.planning_joints}")

        for q in q_goals:  # direct connection (:175-177)
            if np.linalg.norm(q - q_init) <= self.epsilon:
                return [q_init, q]

        start_tree = Tree(Node(q_init))
        # multi-goal: every goal hangs off a sink at +inf that nearest-neighbour never returns
        sink = Node(np.full_like(q_init, np.inf, dtype=np.float64))
        goal_nodes = [Node(q, sink) for q in q_goals]
        goal_tree = Tree(sink)
        for n in goal_nodes:
            goal_tree.add_node(n)

        rng = np.random.default_rng(seed=self.seed)
        lo, hi = self.model.jnt_range.T
        grow, other = start_tree, goal_tree
        swapped = False
        self.iterations = 0
        t0 = time.time()
        while time.time() - t0 < self.max_planning_time:
            self.iterations += 1
            if rng.random() <= self.goal_biasing_probability:
                q_rand = q_init if swapped else goal_nodes[rng.integers(0, len(goal_nodes))].q
            else:
                q_rand = q_init.copy()
                q_rand[q_idx] = rng.uniform(lo, hi)[q_idx]  # njnt draws per sample, as :206
            reached_a = _constrained_extend(q_rand, grow, self.epsilon, self.constraints,
                                            self.collision_interval_check)
            reached_b = _constrained_extend(reached_a, other, self.epsilon, self.constraints,
                                            self.collision_interval_check)
            if np.array_equal(reached_a, reached_b):
                path = _combine_paths(start_tree, start_tree.nearest_neighbor(reached_a),
                                      goal_tree, goal_tree.nearest_neighbor(reached_a))
                return path[:-1]  # drop the sink
            grow, other = other, grow
            swapped = not swapped
        return []
