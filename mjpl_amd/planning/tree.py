"""Planner tree (reference src/mjpl/planning/tree.py:8-85): nodes are unique by ``q``,
nearest neighbour is Euclidean, paths run node -> root.

Same public surface (``Node``, ``Tree.nodes``, ``add_node``, ``nearest_neighbor``,
``get_path``, ``in``) and error behaviour; storage differs: configurations also live in one
growing float64 matrix so the nearest-neighbour query is a single vectorised reduction
(the reference scans a Python set, O(N) interpreter calls per query).
"""
from __future__ import annotations

import numpy as np


class Node:
    """A configuration and its parent.  Equality and hash look at ``q`` only (:15-23)."""

    __slots__ = ("q", "parent", "_key")

    def __init__(self, q: np.ndarray, parent: "Node | None" = None):
        object.__setattr__(self, "q", q)
        object.__setattr__(self, "parent", parent)
        object.__setattr__(self, "_key", np.asarray(q).tobytes())

    def __setattr__(self, name, value):
        raise AttributeError("Node is immutable")

    def __hash__(self):
        return hash(self._key)

    def __eq__(self, other):
        return isinstance(other, Node) and np.array_equal(self.q, other.q)

    def __repr__(self):
        return f"Node(q={self.q}, parent={'None' if self.parent is None else 'Node(...)'})"


class Tree:
    def __init__(self, root: Node):
        if root.parent:
            raise ValueError("The root node should have no parent.")
        self.nodes = {root}
        self._order = [root]
        self._Q = np.empty((64, np.asarray(root.q).size), dtype=np.float64)
        self._Q[0] = root.q

    def __contains__(self, node: Node) -> bool:
        return node in self.nodes

    def __len__(self) -> int:
        return len(self._order)

    def add_node(self, node: Node):
        if not node.parent:
            raise ValueError("Node does not have a parent.")
        if node in self.nodes:
            raise ValueError(f"A node with q={node.q} already exists in the tree.")
        if node.parent not in self.nodes:
            raise ValueError("Node's parent is not in the tree.")
        n = len(self._order)
        if n == len(self._Q):
            self._Q = np.concatenate([self._Q, np.empty_like(self._Q)])
        self._Q[n] = node.q
        self._order.append(node)
        self.nodes.add(node)

    def nearest_neighbor(self, q: np.ndarray) -> Node:
        """Closest node to ``q`` (ties: earliest inserted; the reference leaves ties open,
        test/test_tree.py:91-92).  A root at +inf is never nearest (rrt.py:180-184)."""
        with np.errstate(invalid="ignore", over="ignore"):
            d = self._Q[: len(self._order)] - q
            d2 = np.einsum("ij,ij->i", d, d)
        d2[np.isnan(d2)] = np.inf
        return self._order[int(np.argmin(d2))]

    def get_path(self, node: Node) -> list[Node]:
        if node not in self.nodes:
            raise ValueError("Node is not in the tree.")
        path = []
        cur = node
        while cur is not None:
            path.append(cur)
            cur = cur.parent
        return path

    def configurations(self) -> np.ndarray:
        """All node configurations in insertion order, [len(tree), nq] (a view)."""
        return self._Q[: len(self._order)]
