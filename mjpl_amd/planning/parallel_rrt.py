"""Frontier-sharded bi-directional RRT (BASELINE config 4).

The reference grows its two trees one candidate edge at a time (rrt.py:195-235,
planning/utils.py:139-164).  Here every rank draws its own batch of samples per round, runs
the reference's extend loop on all of its lanes at once -- each step of every lane is one
row of a single batched edge validation -- and the new nodes of all ranks are exchanged once
per round with ONE all-gather of a fixed-size slab (RCCL over xGMI on GPUs, gloo in the CPU
tests).  Every rank appends the slabs in rank order, so all ranks hold bit-identical trees
and node ids; the first connection in (rank, lane) order wins on every rank.

Per-edge accept/stop rules are those of ``_constrained_extend``: the stepped configuration is
first projected if the validator offers ``project`` (PoseConstraint, batched on the GPU), then
the lane stops when the step is invalid, when it moves less than 1e-8, when it does not
approach the target, or when the interval check fails.
"""
from __future__ import annotations

import time

import numpy as np

from .. import utils as _utils


class EdgeValidator:
    """What the planner needs from a collision backend: ``valid_edges(QA, QB, step)`` over
    planning columns -> bool[N] (endpoint QB + interior waypoints; ``step=None`` = endpoint only)."""

    def valid_edges(self, QA: np.ndarray, QB: np.ndarray, step: float | None) -> np.ndarray:
        raise NotImplementedError


class HipEdgeValidator(EdgeValidator):
    """mjpl_amd.CollisionConstraint behind the EdgeValidator interface; also offers the
    brute-force nearest-neighbour kernel (Tree.nearest_neighbor, planning/tree.py:57-66)."""

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
        """Batched PoseConstraint.apply over planning columns (the first constraint of
        apply_constraints in the reference's constrained planning example,
        franka_constrained_move_to_pose.py:60-64).  A projection that would move a joint
        outside the planning set is rejected."""
        out, ok, _ = self.pose.apply_batch(self._full_rows(Q_old), self._full_rows(Q))
        fixed = np.ones(out.shape[1], bool)
        fixed[self.qidx] = False
        ok = ok & np.all(out[:, fixed] == self.qbase[None, fixed], axis=1)
        return out[:, self.qidx], ok

    def nearest(self, nodes: np.ndarray, queries: np.ndarray) -> np.ndarray:
        """Index of the node nearest to each query (squared Euclidean distance in float64,
        ties to the lowest index), computed by ``mjpl_nearest_dev``."""
        eng = self.c.engine
        self.c._ensure_planning()
        n, m = len(nodes), len(queries)
        hn = np.ascontiguousarray(nodes.T)
        hq = np.ascontiguousarray(queries.T)
        dn, dq = eng.alloc(hn.nbytes).upload(hn), eng.alloc(hq.nbytes).upload(hq)
        di = eng.alloc(4 * m)
        try:
            eng.nearest_dev(dn.ptr, n, n, dq.ptr, m, di.ptr)
            return di.download(np.int32, m).astype(np.int64)
        finally:
            for b in (dn, dq, di):
                b.free()


def _row_norm(d: np.ndarray) -> np.ndarray:
    return np.sqrt(np.einsum("ij,ij->i", d, d))


class _Pending:
    """This rank's new nodes of the current round, in creation order.  Entry k is referred to
    as ``-(k + 1)`` until the exchange has given it a global id."""

    def __init__(self):
        self.rows: list[np.ndarray] = []   # chunks [m, n]
        self.parents: list[int] = []
        self.trees: list[int] = []
        self.keys: list[bytes] = []

    def __len__(self) -> int:
        return len(self.parents)

    def add(self, rows: np.ndarray, parents: list, tree: int, keys: list) -> None:
        self.rows.append(rows)
        self.parents.extend(parents)
        self.trees.extend([tree] * len(parents))
        self.keys.extend(keys)

    def slab(self, nrows: int, ncols: int) -> np.ndarray:
        out = np.zeros((max(nrows, 1), ncols + 2))
        cnt = len(self)
        if cnt:
            out[:cnt, :ncols] = np.concatenate(self.rows)
            out[:cnt, ncols] = self.parents
            out[:cnt, ncols + 1] = self.trees
        return out


class ParallelBiRRT:
    def __init__(self, model, planning_joints: list[str], validator: EdgeValidator, q_template: np.ndarray,
                 epsilon: float = 0.05, interval_step: float | None = None, seed: int = 0,
                 goal_biasing_probability: float = 0.05, batch: int = 256, max_rounds: int = 1000,
                 max_planning_time: float = 10.0, max_new_per_round: int = 1 << 20, group=None):
        if not planning_joints:
            raise ValueError("`planning_joints` cannot be empty.")
        if epsilon <= 0.0:
            raise ValueError("`epsilon` must be > 0.0")
        if not 0.0 <= goal_biasing_probability <= 1.0:
            raise ValueError("`goal_biasing_probability` must be within [0.0, 1.0].")
        if batch <= 0 or max_rounds <= 0 or max_planning_time <= 0.0:
            raise ValueError("`batch`, `max_rounds` and `max_planning_time` must be > 0")
        self.model = model
        self.qidx = np.asarray(_utils.qpos_idx(model, planning_joints), dtype=np.int64)
        self.validator = validator
        self.q_template = np.asarray(q_template, dtype=np.float64).copy()
        self.eps, self.interval_step = float(epsilon), interval_step
        self.seed, self.p_goal = int(seed), float(goal_biasing_probability)
        self.batch, self.max_rounds, self.max_time = int(batch), int(max_rounds), float(max_planning_time)
        self.slab_rows = int(max_new_per_round)
        self.lo = np.asarray(model.jnt_range[self.qidx, 0], dtype=np.float64)
        self.hi = np.asarray(model.jnt_range[self.qidx, 1], dtype=np.float64)
        self.group = group
        self.gpu_nn_min_nodes = 2048  # below this the host reduction is faster than a launch
        self.rank, self.world = 0, 1
        if group is not None:
            import torch.distributed as dist
            self._dist = dist
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.stats = {}

    # ------------------------------------------------------------------ replicated tree store
    def _reset(self, q_init_p, q_goal_p):
        n = len(self.qidx)
        self.Q = np.empty((1024, n))
        self.parent = np.empty(1024, np.int64)
        self.tree = np.empty(1024, np.int8)
        self.n = 0
        self.index: dict[bytes, int] = {}
        self._append(q_init_p, -1, 0)
        self._append(q_goal_p, -1, 1)

    def _reserve(self, extra: int) -> None:
        while self.n + extra > len(self.Q):
            self.Q = np.concatenate([self.Q, np.empty_like(self.Q)])
            self.parent = np.concatenate([self.parent, np.empty_like(self.parent)])
            self.tree = np.concatenate([self.tree, np.empty_like(self.tree)])

    def _append(self, q, parent, tree) -> int:
        key = q.tobytes() + bytes([tree])
        hit = self.index.get(key)
        if hit is not None:
            return hit
        self._reserve(1)
        i = self.n
        self.Q[i], self.parent[i], self.tree[i] = q, parent, tree
        self.index[key] = i
        self.n += 1
        return i

    def _nearest(self, targets, tree):
        ids = np.flatnonzero(self.tree[: self.n] == tree)
        nodes = self.Q[ids]
        gpu_nn = getattr(self.validator, "nearest", None)
        if gpu_nn is not None and len(nodes) >= self.gpu_nn_min_nodes:
            return ids[gpu_nn(nodes, targets)]
        out = np.empty(len(targets), np.int64)
        for s in range(0, len(targets), 256):  # chunked [chunk, n_tree] distance matrix
            t = targets[s:s + 256]
            d2 = (np.einsum("ij,ij->i", t, t)[:, None] - 2.0 * t @ nodes.T
                  + np.einsum("ij,ij->i", nodes, nodes)[None, :])
            out[s:s + 256] = ids[np.argmin(d2, axis=1)]
        return out

    # ------------------------------------------------------------------ batched extend
    def _extend(self, targets, tree, pending):
        """Reference extend loop on every lane at once.  ``pending`` (:class:`_Pending`) collects
        this rank's new nodes: a parent reference >= 0 is a global node id, < 0 refers to pending
        entry ``-1 - ref``.  Returns (q_reached, ref_reached) per lane."""
        B = len(targets)
        near = self._nearest(targets, tree)
        cur = self.Q[near].copy()
        ref = near.copy()
        active = ~np.all(cur == targets, axis=1)
        project = getattr(self.validator, "project", None)
        pend_index: dict[bytes, int] = {}

        tb = bytes([tree])

        def accept_rows(lanes, rows):
            """Lanes move to the configurations `rows`, taken in order (a lane may appear several
            times: its later rows hang off its earlier ones).  A row is an existing node of this
            tree, a node already pending this round, or a new pending node."""
            keys = [r.tobytes() + tb for r in rows]
            new_at, new_parents, new_keys = [], [], []
            ref_l = ref.tolist()
            index_get, pend_get = self.index.get, pend_index.get
            base = len(pending)
            for k, (lane, key) in enumerate(zip(lanes.tolist(), keys)):
                r = index_get(key)
                if r is None:
                    r = pend_get(key)
                    if r is None:
                        new_at.append(k)
                        new_parents.append(ref_l[lane])
                        new_keys.append(key)
                        r = -(base + len(new_at))
                        pend_index[key] = r
                ref_l[lane] = r
            if new_at:
                pending.add(rows[new_at], new_parents, tree, new_keys)
            ref[:] = ref_l
            # every lane ends on its last row
            last = np.full(B, -1)
            last[lanes] = np.arange(len(lanes))
            moved = np.flatnonzero(last >= 0)
            cur[moved] = rows[last[moved]]

        if project is None:
            # Nothing projects: a lane's candidates do not depend on the verdicts.  Generate every
            # lane's whole chain of steps first, validate all their edges in ONE launch, then let each
            # lane keep the steps before its first failure.
            walk = cur.copy()
            chains_a, chains_b, owner = [], [], []
            alive = active.copy()
            while alive.any():
                a = np.flatnonzero(alive)
                d = targets[a] - walk[a]
                dist = _row_norm(d)
                q_new = walk[a] + d / dist[:, None] * np.minimum(self.eps, dist)[:, None]
                reach = np.all(q_new == targets[a], axis=1) | (dist <= self.eps)
                q_new[reach] = targets[a][reach]
                ok = np.all((q_new >= self.lo) & (q_new <= self.hi), axis=1)
                ok &= ~(_row_norm(q_new - walk[a]) < 1e-8)
                ok &= ~(_row_norm(targets[a] - q_new) > dist)
                good = a[ok]
                chains_a.append(walk[good].copy())
                chains_b.append(q_new[ok])
                owner.append(good)
                walk[good] = q_new[ok]
                alive[a[~ok | reach]] = False
            if owner and sum(len(o) for o in owner):
                QA, QB = np.concatenate(chains_a), np.concatenate(chains_b)
                lanes_f = np.concatenate(owner)
                level_f = np.concatenate([np.full(len(o), s) for s, o in enumerate(owner)])
                valid = np.asarray(self.validator.valid_edges(QA, QB, self.interval_step), dtype=bool)
                first_fail = np.full(B, len(owner))
                np.minimum.at(first_fail, lanes_f[~valid], level_f[~valid])
                keep = level_f < first_fail[lanes_f]  # step by step (levels), lanes ascending within a level
                if keep.any():
                    accept_rows(lanes_f[keep], QB[keep])
            return cur, ref

        while active.any():
            a = np.flatnonzero(active)
            d = targets[a] - cur[a]
            dist = _row_norm(d)
            q_new = cur[a] + d / dist[:, None] * np.minimum(self.eps, dist)[:, None]
            reach = np.all(q_new == targets[a], axis=1) | (dist <= self.eps)
            q_new[reach] = targets[a][reach]  # `_step` lands on the target within one step
            # constraints that project come first (constraint/utils.py:30-31)
            q_new, ok = project(cur[a], q_new)
            reach = np.all(q_new == targets[a], axis=1)
            ok &= np.all((q_new >= self.lo) & (q_new <= self.hi), axis=1)
            moved = _row_norm(q_new - cur[a])
            ok &= ~(moved < 1e-8)
            ok &= ~(_row_norm(targets[a] - q_new) > dist)
            if ok.any():
                sel = np.flatnonzero(ok)
                ok[sel] = self.validator.valid_edges(cur[a][sel], q_new[sel], self.interval_step)
            sel = np.flatnonzero(ok)
            if len(sel):
                accept_rows(a[sel], q_new[sel])
            done = ~ok | reach
            active[a[done]] = False
        return cur, ref

    # ------------------------------------------------------------------ exchange
    def _allgather(self, arr: np.ndarray) -> np.ndarray:
        """[rows, cols] float64 per rank -> [world, rows, cols] (rank order)."""
        if self.world == 1:
            return arr[None]
        import torch
        dev = "cuda" if self._dist.get_backend(self.group) == "nccl" else "cpu"
        mine = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        out = torch.empty((self.world * arr.shape[0], arr.shape[1]), dtype=mine.dtype, device=dev)
        self._dist.all_gather_into_tensor(out, mine, group=self.group)
        return out.cpu().numpy().reshape((self.world,) + arr.shape)

    def _exchange(self, pending, connection):
        """One exchange step per round: a 4-double header per rank (count, connection refs, stop
        flag), then ONE all-gather of the new-node slabs, padded to the round's largest count
        rounded up to a power of two (few distinct message sizes).  Slabs are merged in rank
        order on every rank, so node ids are global and identical everywhere."""
        n = len(self.qidx)
        head = np.array([[len(pending), np.nan, np.nan, float(time.time() - self._t0 >= self.max_time)]])
        if connection is not None:
            head[0, 1:3] = connection
        heads = self._allgather(head)[:, 0, :]
        most = int(heads[:, 0].max())
        if most > self.slab_rows:
            raise RuntimeError("more new nodes in one round than `max_new_per_round`")
        rows = 1 << max(most - 1, 0).bit_length() if most else 0
        slab = pending.slab(rows, n)
        slabs = self._allgather(slab) if rows else np.zeros((self.world, 1, n + 2))
        winner, stop = None, False
        for r in range(self.world):
            cnt = int(heads[r, 0])
            block = slabs[r, :cnt]
            qrows = np.ascontiguousarray(block[:, :n])
            prefs = block[:, n].astype(np.int64)
            trees = block[:, n + 1].astype(np.int64)
            if r == self.rank and len(pending.keys) == cnt:
                keys = pending.keys
            else:
                keys = [q.tobytes() + bytes([t]) for q, t in zip(qrows, trees.tolist())]
            fresh = cnt > 0 and len(set(keys)) == cnt and not any(k in self.index for k in keys)
            if fresh:
                # the usual case: none of these nodes exists yet -> append them in one go
                base = self.n
                self._reserve(cnt)
                local_to_global = base + np.arange(cnt, dtype=np.int64)
                self.Q[base:base + cnt] = qrows
                self.tree[base:base + cnt] = trees
                self.parent[base:base + cnt] = np.where(prefs >= 0, prefs, base + (-1 - prefs))
                self.index.update(zip(keys, range(base, base + cnt)))
                self.n += cnt
            else:
                local_to_global = np.empty(cnt, np.int64)
                for k in range(cnt):
                    pref = int(prefs[k])
                    par = pref if pref >= 0 else int(local_to_global[-1 - pref])
                    local_to_global[k] = self._append(qrows[k].copy(), par, int(trees[k]))
            if winner is None and not np.isnan(heads[r, 1]):
                refs = [int(heads[r, 1]), int(heads[r, 2])]
                winner = tuple(x if x >= 0 else int(local_to_global[-1 - x]) for x in refs)
            stop = stop or bool(heads[r, 3])
        return winner, stop

    # ------------------------------------------------------------------ driver
    def plan_to_config(self, q_init: np.ndarray, q_goal: np.ndarray) -> list[np.ndarray]:
        q_init, q_goal = np.asarray(q_init, float), np.asarray(q_goal, float)
        fixed = np.setdiff1d(np.arange(self.model.nq), self.qidx)
        if not np.allclose(q_init[fixed], q_goal[fixed], rtol=0, atol=1e-12):
            raise ValueError("goal config differs from q_init outside of the planning joints")
        a, b = q_init[self.qidx], q_goal[self.qidx]
        ends = np.stack([a, b])
        in_lim = np.all((ends >= self.lo) & (ends <= self.hi), axis=1)
        if not (in_lim.all() and self.validator.valid_edges(ends, ends, None).all()):
            raise ValueError("q_init or q_goal is not a valid configuration")
        if np.linalg.norm(b - a) <= self.eps:
            return [q_init, q_goal]

        self._reset(a, b)
        rng = np.random.default_rng(self.seed + 1000003 * self.rank)
        self._t0 = time.time()
        winner = None
        rounds = checks = 0
        for rounds in range(1, self.max_rounds + 1):
            grow = (rounds - 1) % 2          # tree swap every round (rrt.py:234-235)
            other = 1 - grow
            targets = rng.uniform(self.lo, self.hi, size=(self.batch, len(self.qidx)))
            bias = rng.random(self.batch) <= self.p_goal
            targets[bias] = b if grow == 0 else a  # goal bias: the other tree's root
            pending = _Pending()
            reached_a, ref_a = self._extend(targets, grow, pending)
            reached_b, ref_b = self._extend(reached_a, other, pending)
            checks += 2 * self.batch
            hit = np.flatnonzero(np.all(reached_a == reached_b, axis=1))
            conn = None
            if len(hit):
                k = hit[0]
                conn = (ref_a[k], ref_b[k]) if grow == 0 else (ref_b[k], ref_a[k])
            winner, stop = self._exchange(pending, conn)
            if winner is not None or stop:
                break
        self.stats = dict(rounds=rounds, nodes=self.n, world=self.world,
                          seconds=time.time() - self._t0)
        if winner is None:
            return []
        ia, ib = winner  # node in the start tree, node in the goal tree (same configuration)
        head = []
        while ia >= 0:
            head.append(ia)
            ia = self.parent[ia]
        tail = []
        ib = self.parent[ib]  # the junction configuration appears once
        while ib >= 0:
            tail.append(ib)
            ib = self.parent[ib]
        out = []
        for i in list(reversed(head)) + tail:
            q = self.q_template.copy()
            q[self.qidx] = self.Q[i]
            out.append(q)
        out[0], out[-1] = q_init, q_goal
        return out
