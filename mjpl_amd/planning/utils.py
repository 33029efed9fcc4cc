"""CBiRRT building blocks (reference src/mjpl/planning/utils.py:9-249): constrained extend,
edge interval check, path shortcutting.  Same names, arguments, return values and errors as
the reference; the interval check goes to the engine in ONE launch when the constraint
offers ``valid_interval`` (mjpl_amd.CollisionConstraint does).
"""
from __future__ import annotations

import numpy as np

from ..constraint.constraint_interface import Constraint
from ..constraint.utils import apply_constraints
from .tree import Node, Tree


def path_length(waypoints: list[np.ndarray]) -> float:
    """Sum of Euclidean segment lengths in configuration space (:90-102)."""
    pts = np.asarray(waypoints, dtype=np.float64)
    if len(pts) < 2:
        return 0.0
    return float(np.linalg.norm(pts[1:] - pts[:-1], axis=1).sum())


def _step(start: np.ndarray, target: np.ndarray, max_step_dist: float) -> np.ndarray:
    """Move ``start`` toward ``target`` by at most ``max_step_dist`` (:167-185)."""
    if max_step_dist <= 0.0:
        raise ValueError("`max_step_dist` must be > 0.0")
    if np.array_equal(start, target):
        return start.copy()
    delta = target - start
    dist = np.linalg.norm(delta)
    return start + (delta / dist) * min(max_step_dist, dist)


def _valid_collision_interval(start: np.ndarray, end: np.ndarray, step_dist: float,
                              constraint) -> bool:
    """Do the configurations strictly between ``start`` and ``end``, ``step_dist`` apart, obey
    ``constraint``?  (:188-216).  End points are not checked."""
    if step_dist <= 0.0:
        raise ValueError("`step_dist` must be > 0")
    batched = getattr(constraint, "valid_interval", None)
    if batched is not None:
        return bool(batched(start, end, step_dist))
    # generic constraint: walk the interval on the host, stop at the first violation
    cur = _step(start, end, step_dist)
    while not np.array_equal(cur, end):
        if not constraint.valid_config(cur):
            return False
        cur = _step(cur, end, step_dist)
    return True


def _batch_capable(constraints: list[Constraint], collision_interval_check) -> bool:
    """True when a whole extension can be validated in batched launches: no constraint projects
    (``projects`` is False) and every one offers row-wise ``valid_configs`` (and the interval
    constraint row-wise ``valid_intervals``)."""
    for c in constraints:
        if getattr(c, "projects", True) or not hasattr(c, "valid_configs"):
            return False
    return collision_interval_check is None or hasattr(collision_interval_check[1], "valid_intervals")


def _constrained_extend_batched(q_target: np.ndarray, tree: Tree, eps: float, constraints: list[Constraint],
                                collision_interval_check, equality_threshold: float) -> np.ndarray:
    """``_constrained_extend`` when no constraint projects.  The candidate configurations then do
    not depend on the verdicts -- q_{k+1} = _step(q_k, q_target, eps) -- so the whole chain towards
    the target is generated first and validated in one batch per constraint (plus one for the
    intervals); the tree receives the nodes before the first failing step, exactly the nodes the
    step-by-step loop would have added."""
    node = tree.nearest_neighbor(q_target)
    chain = [node.q]
    while not np.array_equal(q_target, chain[-1]):
        q_prev = chain[-1]
        q = _step(q_prev, q_target, eps)
        if np.linalg.norm(q - q_prev) < equality_threshold:
            break
        if np.linalg.norm(q_target - q) > np.linalg.norm(q_target - q_prev):
            break
        chain.append(q)
    if len(chain) == 1:
        return chain[0]
    Q = np.stack(chain)
    ok = np.ones(len(chain) - 1, dtype=bool)
    for c in constraints:
        idx = np.flatnonzero(ok)
        if len(idx) == 0:
            break
        ok[idx] = np.asarray(c.valid_configs(Q[1:][idx]), dtype=bool)
    if collision_interval_check is not None and ok.any():
        step_dist, cc = collision_interval_check
        n_ok = len(ok) if ok.all() else int(np.argmin(ok))  # only the steps before the first failure matter
        ok[:n_ok] &= np.asarray(cc.valid_intervals(Q[:-1][:n_ok], Q[1:][:n_ok], step_dist), dtype=bool)
    n_add = len(ok) if ok.all() else int(np.argmin(ok))
    for k in range(1, n_add + 1):
        node = Node(chain[k], node)
        tree.add_node(node)
    return chain[n_add]


def _constrained_extend(q_target: np.ndarray, tree: Tree, eps: float, constraints: list[Constraint],
                        collision_interval_check=None, equality_threshold: float = 1e-8) -> np.ndarray:
    """CBiRRT Algorithm 2 (:105-164): grow ``tree`` from its node nearest to ``q_target`` in
    steps of at most ``eps``; returns the configuration reached."""
    if _batch_capable(constraints, collision_interval_check):
        return _constrained_extend_batched(q_target, tree, eps, constraints, collision_interval_check,
                                           equality_threshold)
    node = tree.nearest_neighbor(q_target)
    q_prev = node.q
    q_cur = node.q
    while not np.array_equal(q_target, q_cur):
        q_cur = apply_constraints(q_prev, _step(q_cur, q_target, eps), constraints)
        # stop rules (:151-160): constraint failure | no progress | moved away from the target |
        # colliding interval between the last node and the new configuration
        if q_cur is None:
            return q_prev
        if np.linalg.norm(q_cur - q_prev) < equality_threshold:
            return q_prev
        if np.linalg.norm(q_target - q_cur) > np.linalg.norm(q_target - q_prev):
            return q_prev
        if collision_interval_check is not None and not _valid_collision_interval(
                q_prev, q_cur, *collision_interval_check):
            return q_prev
        node = Node(q_cur, node)
        tree.add_node(node)
        q_prev = q_cur
    return q_cur


def _combine_paths(start_tree: Tree, start_tree_node: Node, goal_tree: Tree,
                   goal_tree_node: Node) -> list[np.ndarray]:
    """Root(start_tree) ... start_tree_node -> goal_tree_node ... root(goal_tree) (:219-249);
    a shared junction configuration appears once."""
    head = [n.q for n in reversed(start_tree.get_path(start_tree_node))]
    tail = [n.q for n in goal_tree.get_path(goal_tree_node)]
    if np.array_equal(head[-1], tail[0]):
        head = head[:-1]
    return head + tail


def smooth_path(waypoints: list[np.ndarray], constraints: list[Constraint],
                collision_interval_check=None, eps: float = 0.05, num_tries: int = 100,
                seed: int | None = None, sparse: bool = False) -> list[np.ndarray]:
    """CBiRRT Algorithm 3 (:9-87): ``num_tries`` random shortcut attempts."""
    if not waypoints:
        raise ValueError("`waypoints` cannot be empty.")
    if eps <= 0.0:
        raise ValueError("`eps` must be > 0.")
    if num_tries <= 0:
        raise ValueError("`num_tries` must be > 0.")

    path = waypoints
    rng = np.random.default_rng(seed=seed)
    for _ in range(num_tries):
        i = rng.integers(0, len(path) - 1)
        j = rng.integers(i + 1, len(path))
        tree = Tree(Node(path[i]))
        reached = _constrained_extend(path[j], tree, eps, constraints, collision_interval_check)
        if not np.array_equal(reached, path[j]):
            continue
        # projections move configurations arbitrarily: keep the shortcut only if it is shorter
        chain = [n.q for n in tree.get_path(tree.nearest_neighbor(reached))]  # path[j] ... path[i]
        if path_length(chain) < path_length(path[i:j + 1]):
            if sparse:
                path = path[:i + 1] + path[j:]
            else:
                chain.reverse()
                path = path[:i] + chain[:-1] + path[j:]
    return path
