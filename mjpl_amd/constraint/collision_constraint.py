"""Collision constraint backed by the MI355X engine -- the drop-in for
``mjpl.CollisionConstraint`` (reference src/mjpl/constraint/collision_constraint.py:8-33).

Scalar ``valid_config`` / ``apply`` keep the reference's semantics exactly (full-nq ``q``,
``apply`` returns the same array object or None), so an instance can be handed to the
reference-shaped planners unchanged, also as the ``collision_interval_check`` constraint
(rrt.py:28, planning/utils.py:12,110).  The batched methods are what the hot path is for.

There is no CPU fallback: constructing one without the HIP library or without a gfx950
device raises.
"""
from __future__ import annotations

import numpy as np

from .. import engine as _engine
from .constraint_interface import Constraint


class CollisionRuleset:
    """Which body pairs may touch (reference collision_constraint.py:36-95).

    The engine folds this rule into its static pair list; this class is the host-side
    statement of the same integer logic, for callers that hold a contact list already.
    """

    def __init__(self, model, allowed_collision_bodies: list[tuple[str, str]] = []) -> None:
        self.model = model
        self.allowed_collisions: np.ndarray | None = None
        if allowed_collision_bodies:
            ids = np.array([[model.body(a).id, model.body(b).id] for a, b in allowed_collision_bodies])
            ids.sort(axis=1)  # (a, b) and (b, a) name the same pair
            self.allowed_collisions = ids[None, :, :]
        self._allowed_set = set() if self.allowed_collisions is None else {
            (int(a), int(b)) for a, b in self.allowed_collisions[0]}

    def obeys_ruleset(self, collision_geometries: np.ndarray) -> bool:
        cg = np.asarray(collision_geometries)
        if cg.ndim != 2 or cg.shape[1] != 2:
            raise ValueError("`collision_geometries` must be a nx2 matrix.")
        if cg.shape[0] == 0:
            return True
        if not self._allowed_set:
            return False
        bodies = np.sort(np.asarray(self.model.geom_bodyid)[cg.astype(np.int64)], axis=1)
        return all((int(a), int(b)) in self._allowed_set for a, b in bodies)


class CollisionConstraint(Constraint):
    """Batched collision validation on one MI355X.

    Args:
        model: :class:`mjpl_amd.model.Model` (stands in for ``mujoco.MjModel``).
        allowed_collision_bodies: body-name pairs whose contacts never invalidate a
            configuration; empty means any contact invalidates (collision_constraint.py:86-88).
        device: HIP device ordinal.
    """

    projects = False  # apply() never moves a configuration: batched extension is allowed

    def __init__(self, model, allowed_collision_bodies: list[tuple[str, str]] = [],
                 device: int = 0) -> None:
        self.model = model
        self.cr = CollisionRuleset(model, allowed_collision_bodies)
        self.engine = _engine.Engine(model, allowed_collision_bodies, device=device)
        self._plan_idx = np.arange(model.nq, dtype=np.int32)
        self._plan_base = np.asarray(model.qpos0, dtype=np.float64).copy()
        self._full = True

    # ---- reference surface ---------------------------------------------------------
    def valid_config(self, q: np.ndarray) -> bool:
        q = np.asarray(q, dtype=np.float64)
        if q.shape != (self.model.nq,):
            raise ValueError(f"q must have shape ({self.model.nq},)")
        self._ensure_full()
        return bool(self.engine.check_configs(q[None, :], _engine.AOS)[0])

    def apply(self, q_old: np.ndarray, q: np.ndarray) -> np.ndarray | None:
        return q if self.valid_config(q) else None

    # ---- batched surface -----------------------------------------------------------
    def set_planning(self, qidx, qpos_base) -> None:
        """Batches passed to the ``*_planning`` methods hold only these qpos columns; every
        other joint stays at ``qpos_base`` (planners keep them at q_init, rrt.py:205-206)."""
        self._plan_idx = np.asarray(qidx, dtype=np.int32).copy()
        self._plan_base = np.asarray(qpos_base, dtype=np.float64).copy()
        self._full = False
        self.engine.set_planning(self._plan_idx, self._plan_base)

    def _ensure_full(self):
        if not self._full:
            self.engine.set_planning(np.arange(self.model.nq, dtype=np.int32),
                                     np.asarray(self.model.qpos0, dtype=np.float64))
            self._full = True

    def _ensure_planning(self):
        if self._full:
            self.engine.set_planning(self._plan_idx, self._plan_base)
            self._full = False

    def valid_configs(self, Q: np.ndarray) -> np.ndarray:
        """Full-nq configurations [N, nq] -> bool [N]."""
        self._ensure_full()
        return self.engine.check_configs(np.asarray(Q, dtype=np.float64), _engine.AOS).astype(bool)

    def valid_interval(self, start: np.ndarray, end: np.ndarray, step_dist: float) -> bool:
        """``_valid_collision_interval(start, end, step_dist, self)`` in one launch
        (planning/utils.py:188-216): interior waypoints only."""
        if step_dist <= 0.0:
            raise ValueError("`step_dist` must be > 0")
        self._ensure_full()
        v = self.engine.check_edges(np.asarray(start, float)[None, :], np.asarray(end, float)[None, :],
                                    step_dist, _engine.AOS, interior_only=True)
        return bool(v[0])

    def valid_intervals(self, starts: np.ndarray, ends: np.ndarray, step_dist: float) -> np.ndarray:
        """Row-wise ``valid_interval``: full-nq edges [N, nq] -> bool [N], one launch."""
        if step_dist <= 0.0:
            raise ValueError("`step_dist` must be > 0")
        self._ensure_full()
        return self.engine.check_edges(np.asarray(starts, dtype=np.float64), np.asarray(ends, dtype=np.float64),
                                       step_dist, _engine.AOS, interior_only=True).astype(bool)

    def valid_configs_planning(self, Q: np.ndarray, layout: int = _engine.AOS) -> np.ndarray:
        self._ensure_planning()
        return self.engine.check_configs(Q, layout).astype(bool)

    def valid_edges_planning(self, QA: np.ndarray, QB: np.ndarray, step_dist: float,
                             layout: int = _engine.AOS, first_bad: bool = False,
                             interior_only: bool = False):
        """Validated edges (endpoint + interior waypoints) over planning columns."""
        if step_dist <= 0.0:
            raise ValueError("`step_dist` must be > 0")
        self._ensure_planning()
        r = self.engine.check_edges(QA, QB, step_dist, layout, first_bad=first_bad,
                                    interior_only=interior_only)
        return (r[0].astype(bool), r[1]) if first_bad else r.astype(bool)
