"""The device-resident frontier bi-RRT (mjpl_rrt_* through the C ABI) against its NumPy
restatement running on the CPU oracle: same seed -> the same targets, the same trees node for node,
the same path; multi-goal / pose goals; the RCCL exchange with a one-rank communicator; and one
GPU's share of BASELINE configs[3] (131 072 lanes per round under [PoseConstraint, JointLimit,
Collision]) checked through properties every accepted node and edge must have."""
import os

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
            return self.o.valid_configs(QB, nthreads=8).astype(bool)
        return self.o.valid_edges(QA, QB, step, nthreads=8).astype(bool)


def _scene(oracle_mod, seed=5):
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    v = OracleValidator(oracle_mod, m, qidx, q_init)
    rng = np.random.default_rng(seed)
    goals = []
    while len(goals) < 3:
        g = q_init.copy()
        g[qidx] = rng.uniform(m.jnt_range[qidx, 0], m.jnt_range[qidx, 1])
        if v.valid_edges(g[qidx][None], g[qidx][None], None)[0]:
            goals.append(g)
    return m, joints, qidx, q_init, v, goals


def test_device_sampler_is_the_host_sampler(oracle_mod):
    m, joints, qidx, q_init, v, goals = _scene(oracle_mod)
    cc = mjpl.CollisionConstraint(m)
    cc.set_planning(qidx, q_init)
    lo, hi = m.jnt_range[qidx, 0], m.jnt_range[qidx, 1]
    G = np.stack([g[qidx] for g in goals])
    r = eng_mod.DeviceRRT(cc.engine, 4096, 1 << 20, lo, hi, epsilon=0.05, interval_step=0.01, goal_bias=0.2, seed=9)
    r.reset(q_init[qidx], G, 9)
    for rnd in (1, 2, 3):
        r.round()
        T, on = r.lanes_state()
        Th, onh = pr.sample_targets(pr.rrt_key(9, 0, rnd), 4096, lo, hi, 0.2, (rnd - 1) % 2, q_init[qidx], G)
        np.testing.assert_array_equal(T, Th)
        np.testing.assert_array_equal(on, onh)
    r.close()


@pytest.mark.parametrize("seed,batch,ngoal", [(1, 48, 1), (2, 256, 3), (3, 64, 2)])
def test_device_trees_equal_the_oracle_validated_planner(oracle_mod, seed, batch, ngoal):
    """Same algorithm, same seed: the GPU planner (float32 filter + float64 kernels, device RNG,
    device nearest neighbour, chunked extension) and the NumPy planner on the CPU oracle must build
    the same two trees node for node and return the same path."""
    m, joints, qidx, q_init, v, goals = _scene(oracle_mod, seed=seed + 10)
    goals = goals[:ngoal]
    kw = dict(epsilon=0.05, interval_step=0.01, seed=seed, goal_biasing_probability=0.1, batch=batch,
              max_planning_time=300.0)
    host = pr.ParallelBiRRT(m, joints, v, q_init, **kw)
    want = host.plan_to_configs(q_init, goals)
    assert len(want) > 2
    cc = mjpl.CollisionConstraint(m)
    dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, capacity=1 << 20, **kw)
    got = dev.plan_to_configs(q_init, goals)
    assert dev.stats["rounds"] == host.stats["rounds"] and dev.stats["nodes"] == host.stats["nodes"]
    for t in (0, 1):
        Q, par = dev.rrt.tree(t)
        np.testing.assert_array_equal(Q, host.trees.nodes(t))
        np.testing.assert_array_equal(par, host.trees.parent[t][: host.trees.n[t]])
    assert len(got) == len(want)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    assert any(np.array_equal(got[-1], g) for g in goals)


@pytest.mark.parametrize("cap,seed,batch", [(5, 2, 64), (12, 6, 200)])
def test_capped_chains_carried_into_later_rounds_equal_the_host_planner(oracle_mod, cap, seed, batch):
    """mjpl_rrt_desc.max_steps_per_round: a lane adds at most `cap` nodes per extension; a lane of
    the growing tree still under way is carried -- it sits the connect phase out and, the next time its tree grows,
    goes on from the node it reached towards the same target.  With caps this small most chains are cut and many lanes
    are carried, over a dozen rounds: the device's trees must still be the NumPy statement's node for node, and the
    statement must report carried lanes (else the test tests nothing)."""
    m, joints, qidx, q_init, v, goals = _scene(oracle_mod, seed=seed + 40)
    kw = dict(epsilon=0.05, interval_step=0.01, seed=seed, goal_biasing_probability=0.1, batch=batch, max_planning_time=300.0,
              max_steps_per_round=cap)
    host = pr.ParallelBiRRT(m, joints, v, q_init, **kw)
    want = host.plan_to_configs(q_init, goals[:2])
    assert len(want) > 2 and host.stats["rounds"] >= 3 and sum(host.carried) > 0 and host.longest_chain == cap
    cc = mjpl.CollisionConstraint(m)
    dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, capacity=1 << 20, **kw)
    got = dev.plan_to_configs(q_init, goals[:2])
    assert dev.stats["rounds"] == host.stats["rounds"] and dev.stats["nodes"] == host.stats["nodes"]
    for t in (0, 1):
        Q, par = dev.rrt.tree(t)
        np.testing.assert_array_equal(Q, host.trees.nodes(t))
        np.testing.assert_array_equal(par, host.trees.parent[t][: host.trees.n[t]])
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))


def test_endpoint_only_validation_and_argument_errors(oracle_mod):
    m, joints, qidx, q_init, v, goals = _scene(oracle_mod, seed=21)
    kw = dict(epsilon=0.07, interval_step=None, seed=4, goal_biasing_probability=0.05, batch=96, max_planning_time=300.0)
    host = pr.ParallelBiRRT(m, joints, v, q_init, **kw)
    want = host.plan_to_config(q_init, goals[0])
    cc = mjpl.CollisionConstraint(m)
    dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, capacity=1 << 20, **kw)
    got = dev.plan_to_config(q_init, goals[0])
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    near = q_init.copy()
    near[qidx[0]] += 0.01
    assert len(dev.plan_to_configs(q_init, [goals[0], near])) == 2
    bad = q_init.copy()
    bad[qidx[3]] = 0.5
    with pytest.raises(ValueError, match="not a valid configuration"):
        dev.plan_to_config(q_init, bad)
    other = goals[0].copy()
    other[8] += 0.01
    with pytest.raises(ValueError, match="outside of the planning joints"):
        dev.plan_to_config(q_init, other)
    with pytest.raises(eng_mod.MjplError, match="status -4"):
        tiny = mjpl.DeviceBiRRT(m, joints, mjpl.CollisionConstraint(m), q_init, capacity=64, **kw)
        tiny.plan_to_config(q_init, goals[0])


def test_allgather_through_a_one_rank_rccl_communicator():
    m = scenes.one_dof_ball()
    e = eng_mod.Engine(m)
    e.comm_init(eng_mod.comm_unique_id(), 0, 1)
    src = np.arange(1000, dtype=np.float64)
    a, b = e.alloc(src.nbytes).upload(src), e.alloc(src.nbytes)
    e.allgather_dev(a.ptr, b.ptr, src.nbytes)
    np.testing.assert_array_equal(b.download(np.float64, 1000), src)
    e.allgather_dev(a.ptr, a.ptr, src.nbytes)  # in place
    np.testing.assert_array_equal(a.download(np.float64, 1000), src)
    with pytest.raises(eng_mod.MjplError, match="already has a communicator"):
        e.comm_init(eng_mod.comm_unique_id(), 0, 1)
    e.comm_destroy()
    e.close()


def _constrained(m, q_init, seed):
    cc = mjpl.CollisionConstraint(m)
    frame = mjpl.site_pose(m, q_init, "ee_site", engine=cc.engine)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), engine=cc.engine)
    cons = [pc, mjpl.JointLimitConstraint(m), cc]
    pc.q_step = np.inf
    q_goal = mjpl.random_config(m, q_init, scenes.FRANKA_ARM_JOINTS, seed, cons)
    pc.q_step = 0.05
    return cc, pc, q_goal


def test_one_gpu_share_of_configs3_through_rccl(oracle_mod):
    """BASELINE configs[3] on one GPU: 131 072 samples per round, constraints [PoseConstraint(roll,
    pitch +-0.1), JointLimit, Collision] (examples/franka_constrained_move_to_pose.py:51-64), the
    exchange running through a one-rank RCCL communicator.  Size-independent properties of every
    node the rounds added: on the constraint manifold, inside the joint limits, collision-free, a
    valid edge away from its parent (oracle on a sample), parents preceding children."""
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    cc, pc, q_goal = _constrained(m, q_init, 7)
    L = 131072
    dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=3, goal_biasing_probability=0.05,
                           batch=L, capacity=1 << 22, pose=pc, comm=(eng_mod.comm_unique_id(), 0, 1))
    dev.rrt.reset(q_init[qidx], q_goal[qidx][None], 3)
    infos = [dev.rrt.round(), dev.rrt.round()]
    assert infos[0].new_nodes[0] > 1000 and infos[1].new_nodes[1] > 1000, [(i.new_nodes[0], i.new_nodes[1]) for i in infos]
    orc = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=q_init)
    rng = np.random.default_rng(0)
    for t in (0, 1):
        Q, par = dev.rrt.tree(t)
        n = len(Q)
        assert n == infos[1].nodes[t]
        assert par[0] == -1 and np.all(par[1:] < np.arange(1, n)) and np.all(par[1:] >= 0)
        assert np.all((Q >= m.jnt_range[qidx, 0]) & (Q <= m.jnt_range[qidx, 1]))
        full = np.repeat(q_init[None], n, axis=0)
        full[:, qidx] = Q
        assert pc.valid_configs(full).all()
        cc._ensure_planning()
        assert cc.valid_configs_planning(Q).all()
        step = np.linalg.norm(Q[1:] - Q[par[1:]], axis=1)
        assert step.max() <= 0.05 + 2 * 0.05 + 1e-9 and step.min() >= 1e-8
        pick = rng.choice(np.arange(1, n), size=min(3000, n - 1), replace=False)
        assert orc.valid_edges(Q[par[pick]], Q[pick], 0.01, nthreads=8).all()
    # and the search itself, at a size that finishes quickly: a path on the manifold
    small = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=7, goal_biasing_probability=0.1,
                             batch=2048, capacity=1 << 20, pose=pc, max_planning_time=120.0)
    path = small.plan_to_config(q_init, q_goal)
    assert len(path) > 2, small.stats
    P = np.stack(path)
    np.testing.assert_array_equal(P[0], q_init)
    np.testing.assert_array_equal(P[-1], q_goal)
    assert pc.valid_configs(P).all()
    assert orc.valid_edges(P[:-1, qidx], P[1:, qidx], 0.01, nthreads=8).all()


def test_plan_to_poses_through_batched_ik(oracle_mod):
    """plan_to_poses (rrt.py:106-139): IK seeds for each pose, the valid ones become goals."""
    m, joints, qidx, q_init, v, goals = _scene(oracle_mod, seed=31)
    cc = mjpl.CollisionConstraint(m)
    poses = [mjpl.site_pose(m, g, "ee_site", engine=cc.engine) for g in goals[:2]]
    dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=5, goal_biasing_probability=0.1,
                           batch=1024, capacity=1 << 18, max_planning_time=120.0)
    path = dev.plan_to_poses(q_init, poses, "ee_site")
    assert len(path) > 2
    end = mjpl.site_pose(m, path[-1], "ee_site", engine=cc.engine)
    err = min(np.linalg.norm(end.translation() - p.translation()) for p in poses)
    assert err < 2e-3
    P = np.stack(path)
    assert v.valid_edges(P[:-1, qidx], P[1:, qidx], 0.01).all()


@pytest.mark.parametrize("policy", [("1", "32768"), ("3", "100"), ("64", "65536"), ("7", "1000000")])
def test_projecting_extension_makes_the_same_trees_whatever_the_steps_per_launch(policy):
    """Under a PoseConstraint a lane's candidates depend on the projection of the step before, not on its collision
    verdict: the device walks up to S steps of every active lane in one launch (k_rrt_gen_project), validates them
    together and keeps each lane's leading valid ones.  The trees must be, bit for bit, those of S = 1 -- one step
    per launch, the level-synchronous loop -- for any S and any slot budget (MJPL_RRT_PROJ_STEPS / _SLOTS)."""
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    trees = []
    # (the S = 1 run also keeps its projection on the INTERPRETING kernel, MJPL_POSE_SPEC=0; the other run takes the
    #  chain as straight-line code from the model's library, mjpl_project.h: the same trees, bit for bit)
    for (steps, slots), pose_spec in zip((("1", "1"), policy), ("0", "1")):
        old = {k: os.environ.get(k) for k in ("MJPL_RRT_PROJ_STEPS", "MJPL_RRT_PROJ_SLOTS", "MJPL_POSE_SPEC")}
        os.environ.update(MJPL_RRT_PROJ_STEPS=steps, MJPL_RRT_PROJ_SLOTS=slots, MJPL_POSE_SPEC=pose_spec)
        try:
            cc, pc, q_goal = _constrained(m, q_init, 7)
            dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=11, goal_biasing_probability=0.05,
                                   batch=4096, capacity=1 << 21, pose=pc)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
        assert pc._proj.spec_loaded() == (pose_spec == "1")
        dev.rrt.reset(q_init[qidx], q_goal[qidx][None], 11)
        infos = [dev.rrt.round() for _ in range(3)]
        assert infos[-1].nodes[0] > 5000 and infos[-1].nodes[1] > 5000
        trees.append([dev.rrt.tree(t) for t in (0, 1)])
        dev.rrt.close()
    for t in (0, 1):
        np.testing.assert_array_equal(trees[0][t][0], trees[1][t][0])
        np.testing.assert_array_equal(trees[0][t][1], trees[1][t][1])


def test_first_round_of_131072_lanes_equals_the_host_planner_on_a_lane_prefix(oracle_mod):
    """configs[3]'s per-GPU lane count, round 1: lanes are independent (every lane extends from q_init; the
    sampler is counter-based, keyed by seed, rank and round, lane l drawing from its own counters), so the new
    nodes of lanes 0 .. 4095 -- a prefix of the round's block, nodes being ordered (lane, level) -- must be, bit
    for bit, what the NumPy planner on the CPU oracle adds in ITS first round with 4 096 lanes: same rows, same
    parents, for the extension towards the samples and for the other tree's extension towards what was reached.
    [JointLimit, Collision] with interval checks (the projecting set is compared through properties in
    test_one_gpu_share_of_configs3_through_rccl: the CPU projection differs from the device's in the last bits)."""
    m, joints, qidx, q_init, v, goals = _scene(oracle_mod, seed=77)
    kw = dict(epsilon=0.05, interval_step=0.01, seed=21, goal_biasing_probability=0.05, max_planning_time=600.0)
    host = pr.ParallelBiRRT(m, joints, v, q_init, batch=4096, max_rounds=1, **kw)
    host.plan_to_configs(q_init, goals[:1])
    assert host.stats["rounds"] == 1
    cc = mjpl.CollisionConstraint(m)
    dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, batch=131072, capacity=1 << 25, **kw)
    dev.rrt.reset(q_init[qidx], goals[0][qidx][None], 21)
    info = dev.rrt.round()
    T, on = dev.rrt.lanes_state()
    Th, onh = pr.sample_targets(pr.rrt_key(21, 0, 1), 4096, m.jnt_range[qidx, 0], m.jnt_range[qidx, 1], 0.05, 0, q_init[qidx],
                                goals[0][qidx][None])
    np.testing.assert_array_equal(T[:4096], Th)
    # (of the biased lanes that share a target the lowest takes part: the lowest of all 131 072 is the lowest of the prefix)
    np.testing.assert_array_equal(on[:4096], onh)
    for t in (0, 1):
        Q, par = dev.rrt.tree(t)
        hq, hp = host.trees.nodes(t), host.trees.parent[t][: host.trees.n[t]]
        k = len(hq)
        assert info.nodes[t] > 20 * k > 1000  # the prefix is a small part of a big round
        np.testing.assert_array_equal(Q[:k], hq)
        np.testing.assert_array_equal(par[:k], hp)
        assert par[k] == 0  # the next chain starts at the root, too: lane 4096 (or later) begins


@pytest.mark.parametrize("lanes_per_row", ["1", "4", "8", "16", "64"])
def test_projecting_extension_makes_the_same_trees_whatever_the_lanes_per_row(lanes_per_row):
    """The generated chunk kernel (mjpl_rows.h: k_rrt_gen_project_rows) gives a row -- an active lane's chain of steps --
    one lane, four or eight -- or sixteen, two halves of eight of which one runs the next step's first Newton pass beside
    this step's closing evaluation (k_rrt_gen_project_ahead); by default the host picks by the number of active lanes.  Forced to one value
    for every chunk (MJPL_RRT_PROJ_G), the trees of three rounds must be, bit for bit, those of the interpreting kernel
    at one step per launch: the lanes of a row only share out the statements."""
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    trees = []
    keys = ("MJPL_RRT_PROJ_STEPS", "MJPL_RRT_PROJ_SLOTS", "MJPL_POSE_SPEC", "MJPL_RRT_PROJ_G")
    for env in (dict(MJPL_RRT_PROJ_STEPS="1", MJPL_RRT_PROJ_SLOTS="1", MJPL_POSE_SPEC="0"), dict(MJPL_RRT_PROJ_G=lanes_per_row)):
        old = {k: os.environ.pop(k, None) for k in keys}
        os.environ.update(env)
        try:
            cc, pc, q_goal = _constrained(m, q_init, 7)
            dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=13, goal_biasing_probability=0.05,
                                   batch=2048, capacity=1 << 21, pose=pc)
        finally:
            for k in keys:
                os.environ.pop(k, None)
                if old[k] is not None:
                    os.environ[k] = old[k]
        dev.rrt.reset(q_init[qidx], q_goal[qidx][None], 13)
        infos = [dev.rrt.round() for _ in range(3)]
        assert infos[-1].nodes[0] > 2000 and infos[-1].nodes[1] > 2000
        trees.append([dev.rrt.tree(t) for t in (0, 1)])
        dev.rrt.close()
    for t in (0, 1):
        np.testing.assert_array_equal(trees[0][t][0], trees[1][t][0])
        np.testing.assert_array_equal(trees[0][t][1], trees[1][t][1])


@pytest.mark.parametrize("early_lanes,early_next", [("64", "1"), ("3000", "1"), ("3000", "0")])
def test_connect_phase_neighbours_looked_up_early_make_the_same_trees(early_lanes, early_next):
    """Round 5: once no more than MJPL_RRT_EARLY_LANES lanes of the first extension are still under way, the lanes that are
    through get their nearest node of the other tree on a second stream while the tail runs; the second extension looks up
    the rest and merges (mjpl_rrt.h: early_nn).  Behind that look-up, on the same stream, the NEXT round's targets are drawn
    and looked up in the nodes its growing tree holds now; the next round scans only what this round adds behind them
    (MJPL_RRT_EARLY_NEXT).  The same queries against the same snapshots: the trees of four rounds must be, bit for bit,
    those of a planner that looks every lane up when its extension begins (MJPL_RRT_EARLY_NN=0)."""
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    trees = []
    keys = ("MJPL_RRT_EARLY_NN", "MJPL_RRT_EARLY_LANES", "MJPL_RRT_EARLY_MIN_NODES", "MJPL_RRT_EARLY_NEXT")
    for env in (dict(MJPL_RRT_EARLY_NN="0"),
                dict(MJPL_RRT_EARLY_NN="1", MJPL_RRT_EARLY_LANES=early_lanes, MJPL_RRT_EARLY_MIN_NODES="1", MJPL_RRT_EARLY_NEXT=early_next)):
        old = {k: os.environ.pop(k, None) for k in keys}
        os.environ.update(env)
        try:
            cc, pc, q_goal = _constrained(m, q_init, 7)
            dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=17, goal_biasing_probability=0.05,
                                   batch=4096, capacity=1 << 21, pose=pc)
        finally:
            for k in keys:
                os.environ.pop(k, None)
                if old[k] is not None:
                    os.environ[k] = old[k]
        dev.rrt.reset(q_init[qidx], q_goal[qidx][None], 17)
        infos = [dev.rrt.round() for _ in range(4)]
        assert infos[-1].nodes[0] > 5000 and infos[-1].nodes[1] > 5000
        trees.append([dev.rrt.tree(t) for t in (0, 1)])
        dev.rrt.close()
    for t in (0, 1):
        np.testing.assert_array_equal(trees[0][t][0], trees[1][t][0])
        np.testing.assert_array_equal(trees[0][t][1], trees[1][t][1])


def test_look_ups_ahead_at_planner_size_make_the_same_trees():
    """The same comparison at a planner's size -- 65 536 lanes, three rounds, trees of several hundred thousand nodes --
    with the look-ups ahead switched on (round 6: off by default) and the other defaults: the connect phase's and the next round's look-ups on the second stream, the next round's scan of
    the appended nodes behind the early answer through the matrix-core screen (nearest_range), every chunk sized by its own
    lane count, the tail's rows a step ahead -- against a planner with all of that off."""
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    trees = []
    keys = ("MJPL_RRT_EARLY_NN", "MJPL_RRT_EXACT_COUNTS", "MJPL_RRT_AHEAD")
    for env in (dict(MJPL_RRT_EARLY_NN="0", MJPL_RRT_EXACT_COUNTS="0", MJPL_RRT_AHEAD="0"), dict(MJPL_RRT_EARLY_NN="1")):
        old = {k: os.environ.pop(k, None) for k in keys}
        os.environ.update(env)
        try:
            cc, pc, q_goal = _constrained(m, q_init, 7)
            dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=23, goal_biasing_probability=0.05,
                                   batch=65536, capacity=1 << 23, pose=pc)
        finally:
            for k in keys:
                os.environ.pop(k, None)
                if old[k] is not None:
                    os.environ[k] = old[k]
        dev.rrt.reset(q_init[qidx], q_goal[qidx][None], 23)
        infos = [dev.rrt.round() for _ in range(3)]
        assert infos[-1].nodes[0] > 200000 and infos[-1].nodes[1] > 200000
        trees.append([dev.rrt.tree(t) for t in (0, 1)])
        dev.rrt.close()
        cc.engine.close()
    for t in (0, 1):
        np.testing.assert_array_equal(trees[0][t][0], trees[1][t][0])
        np.testing.assert_array_equal(trees[0][t][1], trees[1][t][1])


def test_cell_ordered_look_ups_make_the_same_trees_at_planner_size():
    """Round 6: from 131 072 nodes on the planner's look-ups take the cell-ordered scan (mjpl_nearest_cells.h) -- whole-tree
    look-ups on both streams AND the ranged look-ups behind an earlier answer.  65 536 lanes, four rounds, trees of several
    hundred thousand nodes: bit for bit the trees of a planner whose look-ups scan every node (option nn_cells = 0)."""
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    trees = []
    for cells in (0, 1):
        with eng_mod.options(nn_cells=cells):
            cc, pc, q_goal = _constrained(m, q_init, 7)
        assert cc.engine.get_option("nn_cells") == cells
        dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=29, goal_biasing_probability=0.05,
                               batch=65536, capacity=1 << 23, pose=pc)
        dev.rrt.reset(q_init[qidx], q_goal[qidx][None], 29)
        infos = [dev.rrt.round() for _ in range(4)]
        assert infos[-1].nodes[0] > 300000 and infos[-1].nodes[1] > 300000
        assert cc.engine.get_option("nn_last_cells") == cells
        trees.append([dev.rrt.tree(t) for t in (0, 1)])
        dev.rrt.close()
        cc.engine.close()
    for t in (0, 1):
        np.testing.assert_array_equal(trees[0][t][0], trees[1][t][0])
        np.testing.assert_array_equal(trees[0][t][1], trees[1][t][1])
