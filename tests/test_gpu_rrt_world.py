"""The multi-rank branch of the device planner (BASELINE configs[3]) on ONE GPU: a world
of `mjpl_rrt` handles -- one engine, one stream and one rank identity each -- whose rounds are split at
the exchange (mjpl_rrt_round_begin / _slabs / _finish, the seam mjpl_rrt_round itself is built on) and
whose headers and slabs are laid out as an all-gather leaves them by the test instead of by RCCL.
Reference: the NumPy planner (ParallelBiRRT) run as the same world in lockstep on the CPU oracle,
which tests/test_parallel_rrt.py ties to the gloo flavour.  Required: every rank's two trees, parents
and path equal the host world's node for node -- with unequal slab counts, empty slabs, several
rounds, and rank 1 holding the connecting lane."""
import numpy as np
import pytest

import mjpl_amd as mjpl
from mjpl_amd import engine as eng_mod
from mjpl_amd import scenes
from mjpl_amd.planning import parallel_rrt as pr

pytestmark = pytest.mark.gpu


class OracleValidator(pr.EdgeValidator):
    def __init__(self, oracle_mod, model, qidx, base):
        self.o = oracle_mod.Oracle(model, planning_qidx=qidx, qpos_base=base)

    def valid_edges(self, QA, QB, step):
        if step is None:
            return self.o.valid_configs(QB, nthreads=2).astype(bool)
        return self.o.valid_edges(QA, QB, step, nthreads=2).astype(bool)


def _scene(oracle_mod):
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    v = OracleValidator(oracle_mod, m, qidx, q_init)
    rng = np.random.default_rng(5)
    while True:
        g = q_init.copy()
        g[qidx] = rng.uniform(m.jnt_range[qidx, 0], m.jnt_range[qidx, 1])
        if v.valid_edges(g[qidx][None], g[qidx][None], None)[0]:
            return m, joints, qidx, q_init, g


KW = dict(epsilon=0.05, interval_step=0.01, goal_biasing_probability=0.1)


def _host_world(oracle_mod, world, seed, batch):
    m, joints, qidx, q_init, g = _scene(oracle_mod)

    def one(rank, member):
        v = OracleValidator(oracle_mod, m, qidx, q_init)
        p = pr.ParallelBiRRT(m, joints, v, q_init, seed=seed, batch=batch, max_planning_time=600.0, group=member, **KW)
        path = p.plan_to_config(q_init, g)
        return (np.array(path)[:, qidx], [p.trees.nodes(t).copy() for t in (0, 1)],
                [p.trees.parent[t][: p.trees.n[t]].copy() for t in (0, 1)], p.stats)

    return pr.ThreadGroup(world).run(one)


def _device_world(world, seed, batch, m, joints, qidx, q_init, g, desc=None):
    ccs = [mjpl.CollisionConstraint(m) for _ in range(world)]
    rrts = []
    for k, cc in enumerate(ccs):
        cc.set_planning(qidx, q_init)
        cc._ensure_planning()
        r = eng_mod.DeviceRRT(cc.engine, batch, 1 << 18, m.jnt_range[qidx, 0], m.jnt_range[qidx, 1], epsilon=KW["epsilon"],
                              interval_step=KW["interval_step"], goal_bias=KW["goal_biasing_probability"], seed=seed,
                              max_steps_per_round=64,  # (ParallelBiRRT's default: capped chains are carried)
                              **((desc or {}).get(k, {})))
        r.set_world(k, world)
        r.reset(q_init[qidx], g[qidx][None], seed)
        rrts.append(r)
    return ccs, rrts


@pytest.mark.parametrize("world,seed,batch", [(2, 1, 4), (2, 15, 3), (2, 11, 3), (2, 11, 4), (2, 5, 24), (3, 13, 3)])
def test_device_world_on_one_gpu_equals_the_host_world(oracle_mod, world, seed, batch):
    host = _host_world(oracle_mod, world, seed, batch)
    stats = host[0][3]
    m, joints, qidx, q_init, g = _scene(oracle_mod)
    ccs, rrts = _device_world(world, seed, batch, m, joints, qidx, q_init, g)
    infos = None
    heads_seen = []
    for rnd in range(1, stats["rounds"] + 1):
        infos = pr.lockstep_round(rrts)
        heads_seen.append([(int(i.new_nodes[0]), int(i.new_nodes[1])) for i in infos])
        assert all(i.round == rnd for i in infos)
        if rnd < stats["rounds"]:
            assert not any(i.connected for i in infos), (rnd, stats["rounds"])
    assert all(i.connected for i in infos), "every rank learns of the connection in the same round"
    assert {i.conn_rank for i in infos} == {stats["win_rank"]}
    assert len({(i.conn_start, i.conn_goal) for i in infos}) == 1
    for k in range(world):
        want_path, want_Q, want_par, _ = host[k]
        for t in (0, 1):
            Q, par = rrts[k].tree(t)
            assert (infos[k].nodes[t]) == len(want_Q[t])
            np.testing.assert_array_equal(Q, want_Q[t])
            np.testing.assert_array_equal(par, want_par[t])
        np.testing.assert_array_equal(rrts[k].path(), want_path)
    for r in rrts:
        r.close()


def test_cases_cover_rank_one_winner_empty_and_unequal_slabs(oracle_mod):
    """What the parametrised cases above are there for, read off the host world's last headers."""
    want = {(2, 1, 4): dict(win=1, rounds_gt=1, empty=True), (2, 15, 3): dict(win=1, rounds_gt=1, empty=True),
            (2, 11, 3): dict(win=0, rounds_gt=1), (2, 11, 4): dict(win=0, rounds_gt=1)}
    for (world, seed, batch), w in want.items():
        st = _host_world(oracle_mod, world, seed, batch)[0][3]
        heads = st["last_heads"]
        assert st["win_rank"] == w["win"] and st["rounds"] > 1
        assert heads[0, 0] != heads[1, 0] or heads[0, 1] != heads[1, 1]
        if w.get("empty"):
            assert (heads[:, :2] == 0).any()


def test_a_rank_that_fails_takes_every_rank_out_of_the_same_round(oracle_mod):
    """Rank 0's slab of new nodes holds 4 rows: its extension overflows in round 1.  It must still
    deliver a header -- the other ranks are about to enter the exchange -- and every rank returns the
    error from round_finish of that round (with RCCL: nobody is left waiting in a collective)."""
    m, joints, qidx, q_init, g = _scene(oracle_mod)
    ccs, rrts = _device_world(2, 1, 64, m, joints, qidx, q_init, g, desc={0: dict(max_new_per_round=4)})
    heads = [r.round_begin() for r in rrts]
    assert heads[0][6] == -4 and heads[0][0] == 0 and heads[0][1] == 0 and heads[0][2] == pr.INT_MAX
    assert heads[1][6] == 0 and heads[1][0] > 0
    for k, r in enumerate(rrts):
        with pytest.raises(eng_mod.MjplError, match="rank 0 failed in round 1") as ei:
            r.round_finish(np.stack(heads), [None, None], [None, None], [0, 0])
        assert ei.value.code == -4
        if k == 0:
            assert "pending slab" in str(ei.value) or "new nodes" in str(ei.value)
    for r in rrts:
        r.close()


def test_seam_argument_errors(oracle_mod):
    m, joints, qidx, q_init, g = _scene(oracle_mod)
    ccs, rrts = _device_world(2, 3, 8, m, joints, qidx, q_init, g)
    r = rrts[1]
    with pytest.raises(eng_mod.MjplError, match="without a communicator"):
        r.round()  # rank 1 of 2 and no RCCL communicator: only the split round can run
    with pytest.raises(eng_mod.MjplError, match="no round in flight"):
        r.round_slabs(0)
    with pytest.raises(eng_mod.MjplError, match="bad rank"):
        r.set_world(2, 2)
    r.round_begin()
    with pytest.raises(eng_mod.MjplError, match="has not been finished"):
        r.round_begin()
    with pytest.raises(eng_mod.MjplError, match="a round is in flight"):
        r.set_world(0, 1)
    for x in rrts:
        x.close()
