"""Frontier-sharded bi-directional RRT (BASELINE configs[3]).

The reference grows its two trees one candidate edge at a time (rrt.py:190-235,
planning/utils.py:139-164).  Here every rank draws a batch of samples per round, runs the
reference's extend loop on all of its lanes at once, and the new nodes of all ranks are exchanged
once per round with one all-gather of a padded slab; every rank appends the slabs in rank order,
so all ranks hold bit-identical trees and node ids and the first connection in (rank, lane) order
wins everywhere.

Two flavours of the SAME algorithm (DESIGN.md section 7 states it):

* :class:`DeviceBiRRT` -- the product: trees, lanes and candidates live in HBM
  (``mjpl_rrt_*`` in include/mjpl_hip.h), the exchange is ``ncclAllGather`` called from the library
  on the engine's stream (RCCL over xGMI); the host only ferries the 128-byte ncclUniqueId.
* :class:`ParallelBiRRT` -- a NumPy restatement with any :class:`EdgeValidator` behind it (the CPU
  oracle in the tests) and ``torch.distributed`` (gloo) for the exchange: what the GPU tests compare
  whole trees against, and what the world-size-2 CPU test runs.

Per-step rules are those of ``_constrained_extend``: a step of at most ``epsilon`` towards the
target (a step that gets within ``epsilon`` lands on the target), projected first if a
PoseConstraint is present (constraint order of examples/franka_constrained_move_to_pose.py:60-64),
then joint limits; the lane stops when the step is invalid, moves less than 1e-8, does not
approach the target, or fails the collision check (endpoint + interval waypoints).  Goals are the
roots of the goal tree (the reference's sink node, rrt.py:179-188, is implicit), so
``plan_to_configs`` / ``plan_to_poses`` work as in the reference.

Two rules of the batched planner that the reference's one-edge-at-a-time loop has no need of (round 6;
DESIGN.md section 7), stated here and implemented in the device planner, trees equal node for node:

* ``max_steps_per_round`` (64): a lane adds at most that many nodes per extension; a lane of the growing
  tree still under way is CARRIED -- no connect phase this round, and the next time its tree grows it goes on
  from the node it reached towards the same target instead of drawing a new one (:class:`Carry`).
* of the lanes whose first extension added nothing and that stand on the same node of the tree, only the
  lowest runs a connect phase (all of them would add the same chain, node for node).
"""
from __future__ import annotations

import time

import numpy as np

from .. import utils as _utils

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
INT_MAX = 0x7FFFFFFF


def _sm64(z):
    """splitmix64 finaliser on uint64 arrays (wraps modulo 2^64), as mjpl_rrt.h:sm64."""
    with np.errstate(over="ignore"):
        z = (np.asarray(z, dtype=np.uint64) + _GOLD)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rrt_key(seed: int, rank: int, rnd: int) -> np.uint64:
    s = _sm64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF))
    return _sm64(s ^ _sm64(np.uint64(((rank << 40) ^ rnd) & 0xFFFFFFFFFFFFFFFF)))


def rrt_u01(key, ctr) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = _sm64(np.uint64(key) + np.asarray(ctr, dtype=np.uint64) * _GOLD)
    return (x >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def sample_targets(key, lanes: int, lo, hi, p_goal: float, grow: int, q_init, goals, carry=None):
    """Targets of one round: (T [L, n], on bool[L]).  Lane l uses counters l * (n + 2) + {0..n-1}
    for its columns, + n for the goal-bias draw (biased iff u <= p, rrt.py:197) and + n + 1 for the
    goal pick (rrt.py:201-203).  Of the biased lanes that share a target only the lowest is on.
    `carry` (or None): a :class:`Carry` of the tree that grows -- a carried lane keeps the target (and
    the goal pick) of the round it was capped in instead of this round's draws; among the lanes of one
    goal it competes like any other."""
    n = len(lo)
    base = np.arange(lanes, dtype=np.uint64) * np.uint64(n + 2)
    cols = rrt_u01(key, base[:, None] + np.arange(n, dtype=np.uint64)[None, :])
    T = lo[None, :] + cols * (hi - lo)[None, :]
    biased = rrt_u01(key, base + np.uint64(n)) <= p_goal
    if grow == 0:
        g = np.minimum((rrt_u01(key, base + np.uint64(n + 1)) * len(goals)).astype(np.int64), len(goals) - 1)
        T[biased] = goals[g[biased]]
    else:
        g = np.zeros(lanes, np.int64)
        T[biased] = q_init
    g = np.where(biased, g, -1)
    if carry is not None and carry.flag.any():
        c = carry.flag
        T[c] = carry.T[c]
        g[c] = carry.goal[c]
    on = np.ones(lanes, bool)
    idx = np.flatnonzero(g >= 0)
    if len(idx):
        _, first = np.unique(g[idx], return_index=True)
        on[idx] = False
        on[idx[first]] = True
    if carry is not None:
        return T, on, g
    return T, on


class Carry:
    """Lanes of one tree whose chain was capped (`max_steps_per_round`): they go on, the next time their
    tree grows, towards the same target from the node they had reached."""

    def __init__(self, lanes: int, n: int):
        self.flag = np.zeros(lanes, bool)
        self.T = np.zeros((lanes, n))
        self.goal = np.full(lanes, -1, np.int64)
        self.node = np.zeros(lanes, np.int64)  # node id in the tree (global: the same on every rank)


def row_norm(d: np.ndarray) -> np.ndarray:
    """Row 2-norms with the sequential left-to-right sum the kernels use (mjpl_rrt.h:seqnorm)."""
    s = np.zeros(len(d))
    for c in range(d.shape[1]):
        s = s + d[:, c] * d[:, c]
    return np.sqrt(s)


def nearest(nodes: np.ndarray, targets: np.ndarray) -> np.ndarray:
    """Index of the node nearest to each target: squared distance summed column by column, ties to
    the lowest index (k_nearest_part)."""
    out = np.empty(len(targets), np.int64)
    for s in range(0, len(targets), 512):
        t = targets[s:s + 512]
        d2 = np.zeros((len(t), len(nodes)))
        for c in range(nodes.shape[1]):
            diff = nodes[None, :, c] - t[:, None, c]
            d2 = d2 + diff * diff
        out[s:s + 512] = np.argmin(d2, axis=1)
    return out


class EdgeValidator:
    """What the host planner needs from a collision backend: ``valid_edges(QA, QB, step)`` over
    planning columns -> bool[N] (endpoint QB + interior waypoints; ``step=None`` = endpoint only).
    An optional ``project(Q_old, Q) -> (Q_projected, ok)`` makes the planner project every step."""

    def valid_edges(self, QA: np.ndarray, QB: np.ndarray, step: float | None) -> np.ndarray:
        raise NotImplementedError


class HipEdgeValidator(EdgeValidator):
    """mjpl_amd.CollisionConstraint (and optionally a PoseConstraint) behind the EdgeValidator
    interface, for running the host flavour against the GPU kernels."""

    def __init__(self, constraint, qidx, qpos_base, pose_constraint=None):
        self.c = constraint
        self.c.set_planning(qidx, qpos_base)
        self.qidx = np.asarray(qidx, dtype=np.int64)
        self.qbase = np.asarray(qpos_base, dtype=np.float64).copy()
        self.pose = pose_constraint
        if pose_constraint is not None:
            self.project = self._project  # the planner projects only if the attribute exists

    def valid_edges(self, QA, QB, step):
        if step is None:
            ok = self.c.valid_configs_planning(QB)
            if self.pose is not None and len(QB):
                ok = ok & self.pose.valid_configs(self._full_rows(QB))
            return ok
        return self.c.valid_edges_planning(QA, QB, step)

    def _full_rows(self, Qp):
        full = np.repeat(self.qbase[None, :], len(Qp), axis=0)
        full[:, self.qidx] = Qp
        return full

    def _project(self, Q_old, Q):
        """Batched PoseConstraint.apply over planning columns.  A projection that would move a
        joint outside the planning set is rejected."""
        out, ok, _ = self.pose.apply_batch(self._full_rows(Q_old), self._full_rows(Q))
        fixed = np.ones(out.shape[1], bool)
        fixed[self.qidx] = False
        ok = ok & np.all(out[:, fixed] == self.qbase[None, fixed], axis=1)
        return out[:, self.qidx], ok


class ThreadGroup:
    """An in-process stand-in for a process group: `world` planners, one thread each, meet in
    ``allgather`` at a barrier.  Lets one process run the multi-rank algorithm in lockstep -- what the
    GPU tests compare a world of device planners with (the gloo flavour of the same exchange is
    tested on CPU processes, tests/test_parallel_rrt.py)."""

    def __init__(self, world: int):
        import threading
        self.world = int(world)
        self._barrier = threading.Barrier(self.world)
        self._slots: list = [None] * self.world

    def member(self, rank: int) -> "_ThreadMember":
        return _ThreadMember(self, int(rank))

    def run(self, fn):
        """fn(rank, member) on `world` threads -> list of results in rank order (re-raises the first error)."""
        import threading
        out, err = [None] * self.world, [None] * self.world

        def work(k):
            try:
                out[k] = fn(k, self.member(k))
            except BaseException as ex:  # noqa: BLE001 -- reported to the caller below
                err[k] = ex
                self._barrier.abort()

        threads = [threading.Thread(target=work, args=(k,)) for k in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for ex in err:
            if ex is not None and not isinstance(ex, __import__("threading").BrokenBarrierError):
                raise ex
        for ex in err:
            if ex is not None:
                raise ex
        return out


class _ThreadMember:
    def __init__(self, group: ThreadGroup, rank: int):
        self.group, self.rank, self.world = group, rank, group.world

    def allgather(self, arr: np.ndarray) -> np.ndarray:
        g = self.group
        g._slots[self.rank] = np.array(arr, copy=True)
        g._barrier.wait()
        out = np.stack(g._slots)
        g._barrier.wait()  # nobody overwrites a slot before everyone has read it
        return out


class _TorchMember:
    """torch.distributed process group (gloo on CPU) behind the same three members."""

    def __init__(self, group):
        import torch.distributed as dist
        self._dist, self._group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def allgather(self, arr: np.ndarray) -> np.ndarray:
        import torch
        mine = torch.from_numpy(np.ascontiguousarray(arr))
        out = torch.empty((self.world,) + arr.shape, dtype=mine.dtype)
        self._dist.all_gather_into_tensor(out.view(-1, *arr.shape[1:]) if arr.ndim > 1 else out.view(-1), mine,
                                          group=self._group)
        return out.numpy()


def lockstep_round(rrts, request_stop=False):
    """One round of `world` device planners (:class:`mjpl_amd.engine.DeviceRRT`, rank k = rrts[k],
    each given its identity with ``set_world(k, world)``) with the exchange done HERE instead of by
    RCCL: every rank's header and slabs are read back, laid out as an all-gather would leave them
    (rank k at row k * stride) and handed to every rank's ``round_finish``.  This is the seam
    ``mjpl_rrt_round`` itself is built on (round_begin -> all-gather -> round_finish), so it runs
    the multi-rank branch of the library on any number of GPUs, one included.  -> list of infos."""
    import ctypes as C
    world = len(rrts)
    stop = [request_stop] * world if isinstance(request_stop, bool) else list(request_stop)
    heads = np.stack([r.round_begin(stop[k]) for k, r in enumerate(rrts)])
    nplan = rrts[0].nplan
    stride = [int(heads[:, 0].max()), int(heads[:, 1].max())]
    failed = bool(np.any(heads[:, 6] != 0))
    gathered = []  # per pass: (rows [world, stride, nplan], parents [world, stride])
    for which in (0, 1):
        rows = np.zeros((world, stride[which], nplan))
        par = np.zeros((world, stride[which]), np.int32)
        if not failed:
            for k, r in enumerate(rrts):
                cnt = int(heads[k, which])
                if cnt == 0:
                    continue
                drows, dpar = r.round_slabs(which)
                lib, h = r.eng.lib, r.eng.h
                r.eng._ok(lib.mjpl_d2h(h, rows[k].ctypes.data, drows, cnt * nplan * 8))
                r.eng._ok(lib.mjpl_d2h(h, par[k].ctypes.data, dpar, cnt * 4))
                r.eng.sync()
        gathered.append((rows, par))
    infos, errors = [], []
    for r in rrts:
        bufs = []
        for which in (0, 1):
            rows, par = gathered[which]
            if stride[which] == 0 or failed:
                bufs.append((None, None))
                continue
            bufs.append((r.eng.alloc(rows.nbytes).upload(rows), r.eng.alloc(par.nbytes).upload(par)))
        try:
            infos.append(r.round_finish(heads, [b[0].ptr if b[0] else None for b in bufs],
                                        [b[1].ptr if b[1] else None for b in bufs], stride))
        except Exception as ex:  # noqa: BLE001 -- every rank must get its turn before the first error is raised
            errors.append(ex)
        r.eng.sync()
        for b in bufs:
            for x in b:
                if x is not None:
                    x.free()
    if errors:
        raise errors[0]
    return infos


class _Trees:
    """Start tree (0) and goal tree (1): rows of planning columns + parent ids (-1 = root)."""

    def __init__(self, n):
        self.Q = [np.empty((1024, n)), np.empty((1024, n))]
        self.parent = [np.empty(1024, np.int64), np.empty(1024, np.int64)]
        self.n = [0, 0]

    def append(self, t, rows, parents):
        k = len(rows)
        while self.n[t] + k > len(self.Q[t]):
            self.Q[t] = np.concatenate([self.Q[t], np.empty_like(self.Q[t])])
            self.parent[t] = np.concatenate([self.parent[t], np.empty_like(self.parent[t])])
        a = self.n[t]
        self.Q[t][a:a + k] = rows
        self.parent[t][a:a + k] = parents
        self.n[t] += k

    def nodes(self, t):
        return self.Q[t][: self.n[t]]


class _PlannerBase:
    """Argument checks, goal handling and path assembly shared by both flavours."""

    def __init__(self, model, planning_joints, q_template, epsilon, seed, goal_biasing_probability, batch,
                 max_rounds, max_planning_time):
        if not planning_joints:
            raise ValueError("`planning_joints` cannot be empty.")
        if epsilon <= 0.0:
            raise ValueError("`epsilon` must be > 0.0")
        if not 0.0 <= goal_biasing_probability <= 1.0:
            raise ValueError("`goal_biasing_probability` must be within [0.0, 1.0].")
        if batch <= 0 or max_rounds <= 0 or max_planning_time <= 0.0:
            raise ValueError("`batch`, `max_rounds` and `max_planning_time` must be > 0")
        self.model = model
        self.planning_joints = planning_joints
        self.qidx = np.asarray(_utils.qpos_idx(model, planning_joints), dtype=np.int64)
        self.q_template = np.asarray(q_template, dtype=np.float64).copy()
        self.eps = float(epsilon)
        self.seed, self.p_goal = int(seed), float(goal_biasing_probability)
        self.batch, self.max_rounds, self.max_time = int(batch), int(max_rounds), float(max_planning_time)
        self.lo = np.asarray(model.jnt_range[self.qidx, 0], dtype=np.float64)
        self.hi = np.asarray(model.jnt_range[self.qidx, 1], dtype=np.float64)
        self.rank, self.world = 0, 1
        self.stats: dict = {}

    # -- to be provided
    def _valid_ends(self, ends: np.ndarray) -> np.ndarray:
        raise NotImplementedError

    def _search(self, a: np.ndarray, goals: np.ndarray) -> np.ndarray | None:
        """-> path rows over planning columns (start ... goal) or None"""
        raise NotImplementedError

    # -- the reference's entry points (rrt.py:69-139)
    def plan_to_config(self, q_init: np.ndarray, q_goal: np.ndarray) -> list[np.ndarray]:
        return self.plan_to_configs(q_init, [q_goal])

    def plan_to_pose(self, q_init, pose, site: str, solver=None) -> list[np.ndarray]:
        return self.plan_to_poses(q_init, [pose], site, solver)

    def plan_to_poses(self, q_init, poses, site: str, solver=None) -> list[np.ndarray]:
        """IK for every pose, then plan to whatever configurations came back (rrt.py:106-139)."""
        if solver is None:
            solver = self._default_solver()
        configs = [q for p in poses for q in solver.solve_ik(p, site, q_init_guess=q_init)]
        if configs:  # one batched validity check of all the ends
            ok = self._valid_ends(np.asarray(configs, float)[:, self.qidx])
            configs = [q for q, v in zip(configs, ok) if v]
        return [] if not configs else self.plan_to_configs(q_init, configs)

    def _default_solver(self):
        raise ValueError("plan_to_poses needs an IK `solver` (e.g. mjpl_amd.HipIKSolver)")

    def plan_to_configs(self, q_init: np.ndarray, q_goals) -> list[np.ndarray]:
        q_init = np.asarray(q_init, float)
        q_goals = [np.asarray(q, float) for q in q_goals]
        if not q_goals:
            raise ValueError("`q_goals` cannot be empty")
        fixed = np.setdiff1d(np.arange(self.model.nq), self.qidx)
        # the validators were compiled with q_template holding the joints outside the planning set
        # (the reference keeps them at q_init, rrt.py:205-206): a q_init that disagrees would be
        # searched for another robot configuration than it starts in
        if not np.array_equal(q_init[fixed], self.q_template[fixed]):
            raise ValueError("q_init differs from the planner's `q_template` outside of the planning joints")
        G = np.stack(q_goals)  # (one call for all goals: an IK solver hands a planner a couple of hundred)
        if not np.isclose(G[:, fixed], q_init[fixed][None, :], rtol=0, atol=1e-12).all():
            raise ValueError("goal config differs from q_init outside of the planning joints")
        a = q_init[self.qidx]
        goals = G[:, self.qidx]
        ends = np.concatenate([a[None], goals])
        in_lim = np.all((ends >= self.lo) & (ends <= self.hi), axis=1)
        if not (in_lim.all() and self._valid_ends(ends).all()):
            raise ValueError("q_init or a goal config is not a valid configuration")
        near = np.flatnonzero(np.linalg.norm(G - q_init[None, :], axis=1) <= self.eps)  # a direct connection (rrt.py:174-176)
        if len(near):
            return [q_init, q_goals[int(near[0])]]
        rows = self._search(a, goals)
        if rows is None:
            return []
        out = []
        for r in rows:
            q = self.q_template.copy()
            q[self.qidx] = r
            out.append(q)
        out[0] = q_init
        hit = np.flatnonzero(np.all(goals == rows[-1][None, :], axis=1))
        out[-1] = q_goals[int(hit[0])]
        return out


class ParallelBiRRT(_PlannerBase):
    """Host (NumPy) flavour: any :class:`EdgeValidator`, torch.distributed for the exchange."""

    def __init__(self, model, planning_joints: list[str], validator: EdgeValidator, q_template: np.ndarray,
                 epsilon: float = 0.05, interval_step: float | None = None, seed: int = 0,
                 goal_biasing_probability: float = 0.05, batch: int = 256, max_rounds: int = 1000,
                 max_planning_time: float = 10.0, max_new_per_round: int = 1 << 20, group=None,
                 max_steps_per_round: int = 64):
        super().__init__(model, planning_joints, q_template, epsilon, seed, goal_biasing_probability, batch,
                         max_rounds, max_planning_time)
        if max_steps_per_round < 0:
            raise ValueError("`max_steps_per_round` must be >= 0 (0: no cap)")
        # A lane adds at most this many nodes to a tree per extension (0: as many as its chain has, the
        # reference's _constrained_extend).  A lane of the growing tree that is still under way then is
        # CARRIED: it sits out the connect phase, and the next time its tree grows it goes on from the node
        # it had reached towards the same target instead of drawing a new one.  A connect-phase lane that
        # is capped just stops.  No round waits for a chain of a thousand steps.
        self.max_steps = int(max_steps_per_round)
        self.validator = validator
        self.interval_step = interval_step
        self.slab_rows = int(max_new_per_round)
        # `group`: a torch.distributed process group (gloo), or anything with rank / world / allgather
        # (ThreadGroup.member: several ranks in one process)
        self.group = None
        if group is not None:
            self.group = group if hasattr(group, "world") else _TorchMember(group)
            self.rank, self.world = self.group.rank, self.group.world
        self.trees: _Trees | None = None

    def _valid_ends(self, ends):
        return np.asarray(self.validator.valid_edges(ends, ends, None), dtype=bool)

    # ------------------------------------------------------------------ one extension
    def _extend(self, t: int, targets: np.ndarray, on: np.ndarray, start=None):
        """Extend tree t towards `targets` on the lanes that are `on`.
        -> (reached [L, n], ref [L], new rows, new parents, capped bool[L]): a reference >= 0 is a node id
        of tree t, < 0 means pending entry -1 - ref of this extension; new nodes are ordered (lane, level).
        `start` (or None): (mask, node ids) -- those lanes start at the given node instead of the nearest.
        capped: the lane added `max_steps` nodes and was still under way."""
        nodes = self.trees.nodes(t)
        L = len(targets)
        near = nearest(nodes, targets)
        if start is not None:
            near = np.where(start[0], start[1], near)
        cur = nodes[near].copy()
        active = on & ~np.all(cur == targets, axis=1)
        project = getattr(self.validator, "project", None)
        acc_rows, acc_lane, acc_level = [], [], []
        cnt = np.zeros(L, np.int64)
        capped = np.zeros(L, bool)
        cap = self.max_steps if self.max_steps > 0 else None

        def gen(a, walk):
            d = targets[a] - walk
            dist = row_norm(d)
            q_new = walk + d / dist[:, None] * np.minimum(self.eps, dist)[:, None]
            reach = np.all(q_new == targets[a], axis=1) | (dist <= self.eps)
            q_new[reach] = targets[a][reach]  # a step of at most eps lands on the target
            return q_new, dist, reach

        def rules(a, walk, q_new, dist):
            ok = np.all((q_new >= self.lo) & (q_new <= self.hi), axis=1)
            ok &= ~(row_norm(q_new - walk) < 1e-8)
            ok &= ~(row_norm(targets[a] - q_new) > dist)
            return ok

        if project is None:
            # nothing projects: a lane's candidates do not depend on the verdicts.  Generate every
            # lane's whole chain (its first `max_steps` steps), validate all edges in one launch, keep
            # each lane's valid prefix.
            walk = cur.copy()
            alive = active.copy()
            QA, QB, owner, level = [], [], [], []
            lv = 0
            while alive.any() and (cap is None or lv < cap):
                a = np.flatnonzero(alive)
                q_new, dist, reach = gen(a, walk[a])
                ok = rules(a, walk[a], q_new, dist)
                good = a[ok]
                QA.append(walk[good].copy())
                QB.append(q_new[ok])
                owner.append(good)
                level.append(np.full(len(good), lv))
                walk[good] = q_new[ok]
                alive[a[~ok | reach]] = False
                lv += 1
            if owner and sum(len(o) for o in owner):
                QA, QB = np.concatenate(QA), np.concatenate(QB)
                owner, level = np.concatenate(owner), np.concatenate(level)
                valid = np.asarray(self.validator.valid_edges(QA, QB, self.interval_step), dtype=bool)
                first_fail = np.full(L, lv)
                np.minimum.at(first_fail, owner[~valid], level[~valid])
                keep = level < first_fail[owner]
                acc_rows, acc_lane, acc_level = [QB[keep]], [owner[keep]], [level[keep]]
                if cap is not None:  # still walking after `cap` steps, every one of them valid
                    capped = alive & (first_fail >= cap)
        else:
            while active.any():
                a = np.flatnonzero(active)
                q_new, dist, _ = gen(a, cur[a])
                q_new, ok = project(cur[a], q_new)  # constraints that project come first
                reach = np.all(q_new == targets[a], axis=1)
                ok = np.asarray(ok, bool) & rules(a, cur[a], q_new, dist)
                # the device validates every candidate (a rejected one as a zero-length edge)
                sel = np.flatnonzero(ok)
                if len(sel):
                    ok[sel] = self.validator.valid_edges(cur[a][sel], q_new[sel], self.interval_step)
                sel = np.flatnonzero(ok)
                acc_rows.append(q_new[sel])
                acc_lane.append(a[sel])
                acc_level.append(cnt[a[sel]].copy())
                cur[a[sel]] = q_new[sel]
                cnt[a[sel]] += 1
                active[a[~ok | reach]] = False
                if cap is not None:
                    full = active & (cnt >= cap)
                    capped |= full
                    active &= ~full

        if acc_rows and sum(len(r) for r in acc_rows):
            rows, lane, level = np.concatenate(acc_rows), np.concatenate(acc_lane), np.concatenate(acc_level)
            order = np.lexsort((level, lane))  # lanes ascending, levels ascending within a lane
            rows, lane, level = rows[order], lane[order], level[order]
            self.longest_chain = max(self.longest_chain, int(level.max()) + 1)  # (statistics: most nodes a lane added in one extension)
            pos = np.arange(len(rows))
            parents = np.where(level == 0, near[lane], -pos)  # -1 - (pos - 1)
            reached = cur.copy()
            ref = near.copy()
            last = np.flatnonzero(np.r_[lane[1:] != lane[:-1], True])
            reached[lane[last]] = rows[last]
            ref[lane[last]] = -1 - pos[last]
            return reached, ref, rows, parents, capped
        return cur, near.copy(), np.empty((0, targets.shape[1])), np.empty(0, np.int64), capped

    # ------------------------------------------------------------------ exchange
    def _allgather(self, arr: np.ndarray) -> np.ndarray:
        if self.world == 1:
            return arr[None]
        return self.group.allgather(arr)

    def _search(self, a, goals):
        n = len(self.qidx)
        self.trees = _Trees(n)
        self.trees.append(0, a[None], [-1])
        self.trees.append(1, goals, [-1] * len(goals))
        t0 = time.time()
        winner = None
        rounds = 0
        carry = [Carry(self.batch, n), Carry(self.batch, n)]  # per tree: the lanes whose chains go on
        self.carried = []  # (per round: how many lanes went on from a carried chain -- tests, statistics)
        self.longest_chain = 0
        for rounds in range(1, self.max_rounds + 1):
            grow = (rounds - 1) % 2  # tree swap every round (rrt.py:234-235)
            other = 1 - grow
            key = rrt_key(self.seed, self.rank, rounds)
            cg = carry[grow]
            T, on, goal = sample_targets(key, self.batch, self.lo, self.hi, self.p_goal, grow, a, goals, carry=cg)
            took = cg.flag & on  # (a carried lane that lost its goal to a lower lane is dropped)
            self.carried.append(int(took.sum()))
            RA, refA, rowsA, parA, capped = self._extend(grow, T, on, start=(took, cg.node))
            cg.flag = capped
            cg.T[capped], cg.goal[capped] = T[capped], goal[capped]
            cg.node[capped] = refA[capped]  # (pending: made a node id below, once this rank's block has its base)
            on = on & ~capped               # carried lanes sit the connect phase out
            # Lanes whose first extension added nothing stand on an old node of the tree; those on the SAME node would all
            # run the same connect phase -- the same chain towards the same configuration, node for node (a sixth of
            # a tree's rows were such copies): of them only the lowest lane takes part.
            stuck = np.flatnonzero(on & (refA >= 0))
            if len(stuck):
                _, first = np.unique(refA[stuck], return_index=True)
                on[stuck] = False
                on[stuck[first]] = True
            RB, refB, rowsB, parB, _ = self._extend(other, RA, on)
            hit = np.flatnonzero(on & np.all(RA == RB, axis=1))
            head = np.zeros(8, np.int64)
            head[:3] = len(rowsA), len(rowsB), INT_MAX
            if len(hit):
                k = int(hit[0])  # [3]: node in the start tree, [4]: in the goal tree
                head[2:5] = k, (refA[k] if grow == 0 else refB[k]), (refB[k] if grow == 0 else refA[k])
            head[5] = int(time.time() - t0 >= self.max_time)
            heads = self._allgather(head)
            if max(int(heads[:, 0].max()), int(heads[:, 1].max())) > self.slab_rows:
                raise RuntimeError("more new nodes in one round than `max_new_per_round`")
            win_rank = next((k for k in range(self.world) if heads[k, 2] != INT_MAX), -1)
            bases = {}
            for which, t, rows, par in ((0, grow, rowsA, parA), (1, other, rowsB, parB)):
                mx = int(heads[:, which].max())
                if mx == 0:
                    bases[t] = self.trees.n[t]
                    continue
                slab = np.zeros((mx, n + 1))
                slab[: len(rows), :n] = rows
                slab[: len(rows), n] = par
                slabs = self._allgather(slab)
                for k in range(self.world):
                    cnt = int(heads[k, which])
                    base = self.trees.n[t]
                    if k == win_rank:
                        bases[t] = base
                    p = slabs[k, :cnt, n].astype(np.int64)
                    self.trees.append(t, slabs[k, :cnt, :n], np.where(p >= 0, p, base + (-1 - p)))
                    if k == self.rank and which == 0:  # this rank's capped lanes: their last nodes' ids
                        c = carry[grow].flag
                        carry[grow].node[c] = base + (-1 - carry[grow].node[c])
                bases.setdefault(t, self.trees.n[t])
            if win_rank >= 0:
                ra, rb = int(heads[win_rank, 3]), int(heads[win_rank, 4])
                winner = (ra if ra >= 0 else bases[0] + (-1 - ra), rb if rb >= 0 else bases[1] + (-1 - rb))
                break
            if heads[:, 5].any():
                break
        self.stats = dict(rounds=rounds, nodes=tuple(self.trees.n), world=self.world, seconds=time.time() - t0,
                          win_rank=win_rank, last_heads=np.array(heads))
        if winner is None:
            return None
        ia, ib = winner
        head_ids, tail_ids = [], []
        while ia >= 0:
            head_ids.append(ia)
            ia = int(self.trees.parent[0][ia])
        ib = int(self.trees.parent[1][ib])  # the junction configuration appears once
        while ib >= 0:
            tail_ids.append(ib)
            ib = int(self.trees.parent[1][ib])
        if not tail_ids:  # the junction IS a goal root
            return np.array([self.trees.Q[0][i] for i in reversed(head_ids)])
        return np.array([self.trees.Q[0][i] for i in reversed(head_ids)] + [self.trees.Q[1][i] for i in tail_ids])


class DeviceBiRRT(_PlannerBase):
    """The product flavour: everything of a round stays on the GPU (``mjpl_rrt_*``).

    ``collision`` is an :class:`mjpl_amd.CollisionConstraint`; ``pose`` an optional
    :class:`mjpl_amd.PoseConstraint` built on the same engine.  ``comm`` = (unique_id bytes, rank,
    world) attaches an RCCL communicator to the engine (or pass ``group``, a torch.distributed
    group used ONLY to broadcast the 128-byte id from rank 0)."""

    def __init__(self, model, planning_joints: list[str], collision, q_template: np.ndarray, epsilon: float = 0.05,
                 interval_step: float | None = None, seed: int = 0, goal_biasing_probability: float = 0.05,
                 batch: int = 4096, max_rounds: int = 1000, max_planning_time: float = 10.0, capacity: int = 1 << 20,
                 pose=None, comm=None, group=None, max_new_per_round: int = 0, max_steps_per_round: int = 64):
        super().__init__(model, planning_joints, q_template, epsilon, seed, goal_biasing_probability, batch,
                         max_rounds, max_planning_time)
        from .. import engine as _engine
        self.collision, self.pose = collision, pose
        self.interval_step = interval_step
        collision.set_planning(self.qidx.astype(np.int32), self.q_template)
        collision._ensure_planning()
        eng = collision.engine
        if pose is not None and pose.engine is not eng:
            raise ValueError("the PoseConstraint must be built on the collision constraint's engine")
        if group is not None and comm is None:
            import torch.distributed as dist
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            box = [_engine.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            comm = (box[0], rank, world)
        if comm is not None:
            uid, self.rank, self.world = comm
            eng.comm_init(uid, self.rank, self.world)
        self.rrt = _engine.DeviceRRT(eng, self.batch, capacity, self.lo, self.hi, epsilon=self.eps,
                                     interval_step=interval_step, goal_bias=self.p_goal, seed=self.seed,
                                     pose=None if pose is None else pose._proj, max_new_per_round=max_new_per_round,
                                     max_steps_per_round=max_steps_per_round)

    def _valid_ends(self, ends):
        self.collision._ensure_planning()
        ok = self.collision.valid_configs_planning(ends)
        if self.pose is not None and len(ends):
            full = np.repeat(self.q_template[None, :], len(ends), axis=0)
            full[:, self.qidx] = ends
            ok = ok & self.pose.valid_configs(full)
        return ok

    def _default_solver(self):
        from ..constraint.joint_limit_constraint import JointLimitConstraint
        from ..inverse_kinematics.hip_ik_solver import HipIKSolver
        cons = [JointLimitConstraint(self.model), self.collision]
        return HipIKSolver(self.model, self.planning_joints, cons, seed=self.seed, max_attempts=5,
                           engine=self.collision.engine)

    def _search(self, a, goals):
        self.collision._ensure_planning()
        self.rrt.reset(a, goals, self.seed)
        t0 = time.time()
        info = None
        found = False
        rounds = 0
        for rounds in range(1, self.max_rounds + 1):
            info = self.rrt.round(request_stop=time.time() - t0 >= self.max_time)
            if info.connected:
                found = True
                break
            if info.stop_requested:
                break
        self.stats = dict(rounds=rounds, nodes=(info.nodes[0], info.nodes[1]) if info else (1, len(goals)),
                          world=self.world, seconds=time.time() - t0)
        return self.rrt.path() if found else None
