"""The plug-in type of the drop-in boundary.

Mirrors ``mjpl.constraint.constraint_interface.Constraint``
(reference src/mjpl/constraint/constraint_interface.py:6-33): a constraint answers
``valid_config(q)`` and ``apply(q_old, q)``.  Anything that implements these two methods
can be mixed with :class:`mjpl_amd.CollisionConstraint` in a constraint list.
"""
from __future__ import annotations

import abc

import numpy as np


class Constraint(abc.ABC):
    @abc.abstractmethod
    def valid_config(self, q: np.ndarray) -> bool:
        """True iff the full configuration ``q`` (length nq) satisfies the constraint."""

    @abc.abstractmethod
    def apply(self, q_old: np.ndarray, q: np.ndarray) -> np.ndarray | None:
        """A configuration derived from ``q`` that satisfies the constraint, or None when no
        such configuration can be produced.  ``q_old`` is an earlier configuration."""
