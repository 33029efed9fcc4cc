"""Constraint composition (reference src/mjpl/constraint/utils.py:6-43)."""
from __future__ import annotations

import numpy as np

from .constraint_interface import Constraint


def obeys_constraints(q: np.ndarray, constraints: list[Constraint]) -> bool:
    """Conjunction over ``constraints`` in list order, stopping at the first failure (:16-19)."""
    return all(c.valid_config(q) for c in constraints)


def apply_constraints(q_old: np.ndarray, q: np.ndarray,
                      constraints: list[Constraint]) -> np.ndarray | None:
    """Apply every constraint in list order, then re-validate the result against all of them,
    because a later projection may break an earlier constraint (:37-43)."""
    cur = q
    for c in constraints:
        cur = c.apply(q_old, cur)
        if cur is None:
            return None
    return cur if obeys_constraints(cur, constraints) else None
