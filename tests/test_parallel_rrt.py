"""Frontier-sharded bi-RRT: single process and world_size-2 over gloo on CPU.
The collision backend here is the CPU oracle behind the EdgeValidator interface (tests only)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(world_group=None, seed=1, batch=48):
    from mjpl_amd import scenes
    from mjpl_amd.planning.parallel_rrt import EdgeValidator, ParallelBiRRT
    from oracle import pyoracle

    class OracleValidator(EdgeValidator):
        def __init__(self, model, qidx, base):
            self.o = pyoracle.Oracle(model, planning_qidx=qidx, qpos_base=base)

        def valid_edges(self, QA, QB, step):
            if step is None:
                return self.o.valid_configs(QB, nthreads=2).astype(bool)
            return self.o.valid_edges(QA, QB, step, nthreads=2).astype(bool)

    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    v = OracleValidator(m, qidx, q_init)
    rng = np.random.default_rng(5)
    while True:
        g = q_init.copy()
        g[qidx] = rng.uniform(m.jnt_range[qidx, 0], m.jnt_range[qidx, 1])
        if v.valid_edges(g[qidx][None], g[qidx][None], None)[0]:
            break
    p = ParallelBiRRT(m, joints, v, q_init, epsilon=0.05, interval_step=0.01, seed=seed, batch=batch,
                      goal_biasing_probability=0.1, max_planning_time=120.0, group=world_group)
    return m, qidx, v, p, q_init, g


def _check_path(m, qidx, v, path, q_init, q_goal, eps=0.05):
    assert len(path) >= 2
    np.testing.assert_array_equal(path[0], q_init)
    np.testing.assert_array_equal(path[-1], q_goal)
    P = np.array(path)
    assert np.all(np.linalg.norm(P[1:] - P[:-1], axis=1) <= eps + 1e-12)
    fixed = np.setdiff1d(np.arange(m.nq), qidx)
    assert np.all(P[:, fixed] == q_init[fixed])
    assert v.valid_edges(P[:-1, qidx], P[1:, qidx], 0.01).all()


def test_single_process_plan_is_valid_and_deterministic():
    m, qidx, v, p, q_init, g = _make()
    path = p.plan_to_config(q_init, g)
    _check_path(m, qidx, v, path, q_init, g)
    _, _, _, p2, _, _ = _make()
    again = p2.plan_to_config(q_init, g)
    assert len(again) == len(path) and all(np.array_equal(a, b) for a, b in zip(path, again))
    # every node's parent precedes it (roots: -1); edges are no longer than epsilon
    for t in (0, 1):
        n, par, Q = p.trees.n[t], p.trees.parent[t], p.trees.Q[t]
        assert par[0] == -1 and np.all(par[1:n] < np.arange(1, n)) and np.all(par[1:n] >= 0)
        assert np.all(np.linalg.norm(Q[1:n] - Q[par[1:n]], axis=1) <= 0.05 + 1e-12)


def test_multiple_goals_and_the_sink_semantics():
    """plan_to_configs (rrt.py:141-237): the goal tree's roots are the goals; the path ends on one
    of them; a goal within epsilon of q_init short-circuits; an invalid goal raises."""
    m, qidx, v, p, q_init, g = _make(seed=4)
    rng = np.random.default_rng(11)
    goals = [g]
    while len(goals) < 3:
        c = q_init.copy()
        c[qidx] = rng.uniform(m.jnt_range[qidx, 0], m.jnt_range[qidx, 1])
        if v.valid_edges(c[qidx][None], c[qidx][None], None)[0]:
            goals.append(c)
    path = p.plan_to_configs(q_init, goals)
    assert len(path) > 2
    assert any(np.array_equal(path[-1], q) for q in goals)
    k = [np.array_equal(path[-1], q) for q in goals].index(True)
    _check_path(m, qidx, v, path, q_init, goals[k])
    assert p.trees.n[1] >= 3 and np.all(p.trees.parent[1][:3] == -1)
    near = q_init.copy()
    near[qidx[0]] += 0.01
    assert len(p.plan_to_configs(q_init, [g, near])) == 2
    bad = q_init.copy()
    bad[qidx[3]] = 0.5  # joint4's range is [-3.07, -0.07]
    with pytest.raises(ValueError, match="not a valid configuration"):
        p.plan_to_configs(q_init, [g, bad])


def test_counter_based_sampler_known_answers():
    """The sampler both flavours share (mjpl_rrt.h: sm64 / rrt_key / rrt_u01): splitmix64 outputs
    for seed 0 are the published test vector; lanes, goal bias and duplicate suppression."""
    from mjpl_amd.planning import parallel_rrt as pr
    z = np.uint64(0)
    outs = []
    for _ in range(3):  # splitmix64 stream with state += GOLD: sm64(state) is the published generator
        outs.append(int(pr._sm64(z)))
        z = np.uint64((int(z) + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    assert outs == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    key = pr.rrt_key(7, 1, 3)
    u = pr.rrt_u01(key, np.arange(1000))
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.05
    lo, hi = np.full(7, -1.0), np.full(7, 2.0)
    goals = np.arange(21.0).reshape(3, 7)
    T, on = pr.sample_targets(key, 4096, lo, hi, 0.25, 0, np.zeros(7), goals)
    biased = np.any(T[:, None, :] == goals[None, :, :], axis=2).all(axis=1) if False else np.array(
        [any(np.array_equal(t, g) for g in goals) for t in T])
    assert 0.2 < biased.mean() < 0.3
    assert on[~biased].all() and on[biased].sum() == 3  # one lane per goal
    assert np.all((T[~biased] >= lo) & (T[~biased] < hi))
    T1, on1 = pr.sample_targets(key, 4096, lo, hi, 0.25, 1, np.zeros(7), goals)
    assert on1[biased].sum() == 1 and np.array_equal(T1[~biased], T[~biased])
    T2, _ = pr.sample_targets(pr.rrt_key(7, 0, 3), 4096, lo, hi, 0.25, 0, np.zeros(7), goals)
    assert not np.array_equal(T2, T)  # another rank draws other targets


def test_argument_validation_and_trivial_goal():
    m, qidx, v, p, q_init, g = _make()
    near = q_init.copy()
    near[qidx[0]] += 0.01
    assert len(p.plan_to_config(q_init, near)) == 2
    bad = g.copy()
    bad[8] += 0.01  # a non-planning joint differs
    with pytest.raises(ValueError, match="outside of the planning joints"):
        p.plan_to_config(q_init, bad)
    out = q_init.copy()
    out[qidx[1]] = 5.0  # beyond the joint range
    with pytest.raises(ValueError, match="not a valid configuration"):
        p.plan_to_config(q_init, out)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, qidx, v, p, q_init, g = _make(world_group=dist.group.WORLD, batch=24)
        path = p.plan_to_config(q_init, g)
        _check_path(m, qidx, v, path, q_init, g)
        q.put((rank, np.array(path).tobytes(), tuple(p.trees.n),
               p.trees.nodes(0).tobytes() + p.trees.nodes(1).tobytes(),
               p.trees.parent[0][: p.trees.n[0]].tobytes() + p.trees.parent[1][: p.trees.n[1]].tobytes(), p.stats["world"]))
    finally:
        dist.destroy_process_group()


def test_world_size_two_gloo_ranks_agree():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, path0, n0, Q0, par0, w0), (r1, path1, n1, Q1, par1, w1) = res
    assert (r0, r1, w0, w1) == (0, 1, 2, 2)
    assert path0 == path1, "both ranks must return the same path"
    assert n0 == n1 and Q0 == Q1 and par0 == par1, "replicated trees must be bit-identical"
    # the in-process ThreadGroup (what the GPU tests compare a world of device planners with) is the
    # same exchange: same trees, same path as the two gloo processes
    from mjpl_amd.planning.parallel_rrt import ThreadGroup

    def one(rank, member):
        m, qidx, v, p, q_init, g = _make(world_group=member, batch=24)
        path = p.plan_to_config(q_init, g)
        return (np.array(path).tobytes(), tuple(p.trees.n), p.trees.nodes(0).tobytes() + p.trees.nodes(1).tobytes(),
                p.trees.parent[0][: p.trees.n[0]].tobytes() + p.trees.parent[1][: p.trees.n[1]].tobytes())

    for got in ThreadGroup(2).run(one):
        assert got == (path0, n0, Q0, par0)


def test_thread_group_multi_round_and_rank_one_winner():
    """Lanes few enough that the search takes several rounds; seeds chosen so that rank 1 holds the
    connecting lane and that a rank contributes an empty slab to a round."""
    from mjpl_amd.planning.parallel_rrt import INT_MAX, ThreadGroup
    seen, carried = [], []
    for seed, batch in ((26, 4), (26, 2)):
        def one(rank, member):
            m, qidx, v, p, q_init, g = _make(world_group=member, seed=seed, batch=batch)
            path = p.plan_to_config(q_init, g)
            _check_path(m, qidx, v, path, q_init, g)
            return (np.array(path).tobytes(), tuple(p.trees.n), p.stats["rounds"], p.stats["win_rank"], p.stats["last_heads"],
                    sum(p.carried))

        a, b = ThreadGroup(2).run(one)
        assert a[:4] == b[:4] and np.array_equal(a[4], b[4])
        assert a[2] > 1 and a[3] == 1 and a[4][0, 2] == INT_MAX
        seen.append(a[4])
        carried.append(a[5] + b[5])
    assert (seen[1][:, :2] == 0).any(), "a rank with an empty slab"
    assert carried[0] > 0, "a search of ten rounds in which some lane's chain went on from the round before last"


def test_capped_chains_are_carried_and_the_cap_binds():
    """`max_steps_per_round`: a lane adds at most that many nodes per extension; a lane of
    the growing tree still under way is carried -- no connect phase this round, the same target and the node it
    reached the next time its tree grows.  With a cap of 6 no lane adds more than 6 nodes in
    an extension; carried lanes appear; the path is valid; and a cap no chain reaches changes nothing."""
    from mjpl_amd.planning.parallel_rrt import ParallelBiRRT
    m, qidx, v, p, q_init, g = _make(seed=3, batch=16)
    kw = dict(epsilon=0.05, interval_step=0.01, seed=3, batch=16, goal_biasing_probability=0.1, max_planning_time=120.0)
    capped = ParallelBiRRT(m, p.planning_joints, v, q_init, max_steps_per_round=6, **kw)
    path = capped.plan_to_config(q_init, g)
    _check_path(m, qidx, v, path, q_init, g)
    assert sum(capped.carried) > 0 and capped.carried[0] == 0 and capped.carried[1] == 0  # (round 3 is the first to take over lanes)
    assert capped.longest_chain == 6  # (the cap binds: some chain was cut, none is longer)
    free = ParallelBiRRT(m, p.planning_joints, v, q_init, max_steps_per_round=0, **kw)
    loose = ParallelBiRRT(m, p.planning_joints, v, q_init, max_steps_per_round=100000, **kw)
    pa, pb = free.plan_to_config(q_init, g), loose.plan_to_config(q_init, g)
    assert sum(loose.carried) == 0 and free.longest_chain == loose.longest_chain > 6 and len(pa) == len(pb) and all(np.array_equal(x, y) for x, y in zip(pa, pb))
    assert free.trees.n == loose.trees.n
    with pytest.raises(ValueError, match="max_steps_per_round"):
        ParallelBiRRT(m, p.planning_joints, v, q_init, max_steps_per_round=-1, **kw)


def test_projecting_validator_on_the_cpu(oracle_mod):
    """The planner's `project` hook (PoseConstraint first, constraint/utils.py:30-31) with the CPU
    oracle's projection behind it: every node of the path satisfies the pose constraint."""
    from mjpl_amd import scenes
    from mjpl_amd.lie import SE3, SO3
    from mjpl_amd.planning.parallel_rrt import EdgeValidator, ParallelBiRRT

    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = np.asarray(scenes.planning_index(m, joints))
    q_init = m.keyframe("home").qpos.copy()
    ident = oracle_mod.PoseOracle(m, "ee_site", (np.array([1.0, 0, 0, 0]), np.zeros(3)), [(-np.inf, np.inf)] * 6)
    pos, mat = ident.site_pose(q_init)
    inv = SE3.from_rotation_and_translation(SO3.from_matrix(mat), pos).inverse()
    inf = (-np.inf, np.inf)
    po = oracle_mod.PoseOracle(m, "ee_site", (inv.wxyz_xyz[:4], inv.wxyz_xyz[4:]),
                               [inf, inf, inf, (-0.1, 0.1), (-0.1, 0.1), inf], q_step=0.05)

    class V(EdgeValidator):
        def __init__(self):
            self.o = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=q_init)

        def full(self, Qp):
            F = np.repeat(q_init[None], len(Qp), axis=0)
            F[:, qidx] = Qp
            return F

        def valid_edges(self, QA, QB, step):
            if step is None:
                ok = self.o.valid_configs(QB, nthreads=2).astype(bool)
                return ok & np.array([po.valid_config(r) for r in self.full(QB)], dtype=bool)
            return self.o.valid_edges(QA, QB, step, nthreads=2).astype(bool)

        def project(self, Q_old, Q):
            out, ok, _ = po.apply_batch(self.full(Q_old), self.full(Q), nthreads=2)
            return out[:, qidx], ok

    v = V()
    rng = np.random.default_rng(3)
    while True:  # a goal on the constraint manifold: project a random configuration
        g = q_init.copy()
        g[qidx] = q_init[qidx] + rng.normal(scale=0.4, size=len(qidx))
        po.set_q_step(np.inf)
        gp = po.apply(q_init, g)
        po.set_q_step(0.05)
        if gp is not None and v.valid_edges(gp[qidx][None], gp[qidx][None], None)[0]:
            break
    p = ParallelBiRRT(m, joints, v, q_init, epsilon=0.05, interval_step=0.01, seed=2, batch=32,
                      goal_biasing_probability=0.2, max_planning_time=120.0)
    path = p.plan_to_config(q_init, gp)
    assert len(path) > 2, p.stats
    for q in path:
        assert po.valid_config(q)
    P = np.array(path)
    assert v.o.valid_configs(P[:, qidx], nthreads=2).all()
