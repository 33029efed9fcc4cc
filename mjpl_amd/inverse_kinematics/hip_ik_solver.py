"""Batched IK seeds on the MI355X in the role of the reference's MinkIKSolver
(src/mjpl/inverse_kinematics/mink_ik_solver.py:12-116).

The reference runs ONE damped QP descent per attempt and restarts from ``random_config`` on
failure (:108-115).  Here one attempt is a whole batch: the initial guess plus ``num_seeds - 1``
configurations drawn like ``random_config`` draws them, all iterated by ``k_ik_solve`` (damped
least squares, joint limits clamped, held joints untouched); converged seeds are then filtered
by the constraints -- collision in one batched launch -- and returned closest-to-the-guess
first.  Constructor arguments, defaults and ``ValueError`` cases follow the reference (:15-54);
``qp_solver`` / ``tasks`` have no counterpart.  Parity is tolerance-level, as in the reference's
own test (test/test_mink_ik_solver.py:64-70): solutions meet the pose tolerances and obey the
constraints.
"""
from __future__ import annotations

import numpy as np

from .. import engine as _engine
from .. import utils as _utils
from ..constraint.constraint_interface import Constraint
from ..constraint.utils import obeys_constraints
from ..lie import SE3
from .ik_solver_interface import IKSolver


def distinct_solutions(sols: np.ndarray, q0: np.ndarray, tol: float = 1e-6) -> list[np.ndarray]:
    """The rows of `sols` closest to `q0` first (cartesian_planner.py:101-102 picks that one), a row dropped when an earlier
    KEPT one is within `tol` of it.  Two rows that close are as close in their distance to q0: only runs of neighbours
    whose distances differ by less than `tol` can hold such a pair -- none, as a rule -- and only those are looked at
    pair by pair (one call per pair of a couple of hundred solutions was a third of a planning call's time)."""
    dist = np.linalg.norm(sols - q0, axis=1)
    order = np.argsort(dist, kind="stable")
    sols, dist = sols[order], dist[order]
    kept = np.ones(len(sols), dtype=bool)
    tight = np.flatnonzero(np.diff(dist) < tol)
    k = 0
    while k < len(tight):
        lo = hi = int(tight[k])
        while k < len(tight) and int(tight[k]) == hi:  # the run lo .. hi + 1 of neighbours
            hi += 1
            k += 1
        for j in range(lo + 1, hi + 1):
            prev = np.flatnonzero(kept[lo:j]) + lo
            if len(prev) and np.linalg.norm(sols[prev] - sols[j], axis=1).min() < tol:
                kept[j] = False
    return [q.copy() for q in sols[kept]]


class HipIKSolver(IKSolver):
    def __init__(self, model, joints: list[str], constraints: list[Constraint] = [],
                 pos_tolerance: float = 1e-3, ori_tolerance: float = 1e-3, seed: int | None = None,
                 max_attempts: int = 1, iterations: int = 500, num_seeds: int = 256,
                 engine: _engine.Engine | None = None, device: int = 0, restarts: int = 8, quick_iterations: int = 64):
        if not joints:
            raise ValueError("`joints` cannot be empty.")
        if max_attempts < 1:
            raise ValueError("`max_attempts` must be > 0.")
        if iterations < 1:
            raise ValueError("`iterations` must be > 0.")
        if num_seeds < 1:
            raise ValueError("`num_seeds` must be > 0.")
        self.model = model
        self.joints = joints
        self.constraints = constraints
        self.pos_tolerance, self.ori_tolerance = pos_tolerance, ori_tolerance
        self.seed, self.max_attempts, self.iterations = seed, max_attempts, iterations
        self.num_seeds = num_seeds
        self.restarts = restarts  # in-kernel re-draws of a stalled seed (the reference's restart, :108-115)
        # A launch lasts as long as its slowest seed -- the full iteration budget whenever one seed does not converge,
        # 23 us per iteration -- while most seeds that converge at all do so within ~50 iterations: an attempt first runs
        # with this many iterations and only if NO seed came back valid with the whole budget (0: always the whole budget).
        # The caller's guess (row 0 of the first attempt) is the exception: if the quick pass did not solve IT, it alone
        # runs again with the whole budget, so which solution is closest to the guess does not depend on this number.
        self.quick_iterations = quick_iterations
        self._owns_engine = engine is None
        self.engine = engine if engine is not None else _engine.Engine(model, device=device)
        self.q_idx = np.asarray(_utils.qpos_idx(model, joints), dtype=np.int64)
        self.movable = np.zeros(model.njnt, np.uint8)
        self.movable[[model.joint(j).id for j in joints]] = 1
        self.stats: dict = {}

    def close(self) -> None:
        """Release the engine this solver created for itself (a shared one is left alone)."""
        if self._owns_engine and self.engine is not None:
            self.engine.close()
        self.engine = None

    def _seeds(self, q_start: np.ndarray, rng) -> np.ndarray:
        """Row 0 = the guess; the others re-draw the solver's joints uniformly in their ranges
        (what random_config draws, src/mjpl/utils.py:100-103)."""
        lo, hi = self.model.jnt_range.T
        Q = np.repeat(q_start[None, :], self.num_seeds, axis=0)
        if self.num_seeds > 1:
            draw = rng.uniform(lo, hi, size=(self.num_seeds - 1, self.model.njnt))
            Q[1:, self.q_idx] = draw[:, self.q_idx]
        return Q

    def _obey(self, Q: np.ndarray) -> np.ndarray:
        ok = np.ones(len(Q), bool)
        for c in self.constraints:
            if not ok.any():
                break
            batch = getattr(c, "valid_configs", None)
            idx = np.flatnonzero(ok)
            if batch is not None:
                ok[idx] = np.asarray(batch(Q[idx]), dtype=bool)
            else:
                ok[idx] = [bool(c.valid_config(q)) for q in Q[idx]]
        return ok

    def solve_batch(self, pose: SE3, site: str, Q_start: np.ndarray, iterations: int | None = None):
        """All rows of Q_start as seeds of one launch -> (Q, converged & obeys constraints, err)."""
        Q, ok, iters, err = self.engine.ik_solve(
            site, pose.translation(), pose.rotation().wxyz, Q_start, self.movable,
            pos_tolerance=self.pos_tolerance, ori_tolerance=self.ori_tolerance,
            iterations=self.iterations if iterations is None else iterations,
            restarts=self.restarts, restart_seed=0 if self.seed is None else self.seed + 1)
        good = ok.copy()
        if good.any():
            idx = np.flatnonzero(good)
            good[idx] = self._obey(Q[idx])
        self.stats = dict(seeds=len(Q), converged=int(ok.sum()), valid=int(good.sum()),
                          mean_iters=float(iters.mean()) if len(iters) else 0.0)
        return Q, good, err

    def solve_ik(self, pose: SE3, site: str, q_init_guess: np.ndarray | None) -> list[np.ndarray]:
        q0 = np.asarray(self.model.qpos0 if q_init_guess is None else q_init_guess, dtype=np.float64).copy()
        for attempt in range(self.max_attempts):
            rng = np.random.default_rng(None if self.seed is None else self.seed + attempt)
            seeds = self._seeds(q0, rng)
            good = np.zeros(0, bool)
            if 0 < self.quick_iterations < self.iterations:
                Q, good, _ = self.solve_batch(pose, site, seeds, iterations=self.quick_iterations)
                if good.any() and q_init_guess is not None and attempt == 0 and not good[0]:
                    # The guess itself (row 0) gets the WHOLE budget whatever the quick pass found: the Cartesian planner
                    # takes the solution closest to the guess (cartesian_planner.py:101-102), and a guess that needs more
                    # than quick_iterations must not lose to a far-away re-draw that converged quickly.  One seed, one launch.
                    quick_stats = self.stats
                    Q0, g0, _ = self.solve_batch(pose, site, seeds[:1])
                    self.stats = dict(quick_stats, guess_with_whole_budget=bool(g0[0]), guess_iters=self.stats["mean_iters"])
                    if g0[0]:
                        Q, good = Q.copy(), good.copy()
                        Q[0], good[0] = Q0[0], True
            if not good.any():
                Q, good, _ = self.solve_batch(pose, site, seeds)
            if good.any():
                sols = Q[good]
                # distinct solutions, closest to the guess first (cartesian_planner.py:101-102 picks that one)
                return distinct_solutions(sols, q0)
        return []


__all__ = ("HipIKSolver", "distinct_solutions", "obeys_constraints")
