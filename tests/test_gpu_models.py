"""GPU parity over randomly generated models: every joint/geom/pair type the engine accepts,
branching trees, off-centre hinges, margins, contype/conaffinity masks and allowed body pairs,
each checked bit-exact (verdicts) / 1e-6 (FK) against the CPU oracle through the C ABI."""
import os
import sys

import numpy as np
import pytest

import mjpl_amd as mjpl
from mjpl_amd import engine as eng_mod
from mjpl_amd import scenes
from mjpl_amd.model import ModelBuilder
from helpers import OracleCollisionConstraint, random_edges, uniform_configs

pytestmark = pytest.mark.gpu


def random_model(seed, moving_boxes=True):
    """moving_boxes=False keeps boxes on the world body only: such models run the queued float32
    interpreter (two candidate queues, pair-level exact re-check) instead of the immediate one."""
    rng = np.random.default_rng(seed)
    mb = ModelBuilder()
    if rng.random() < 0.8:
        mb.add_geom("world", "plane", (1, 1, 0.1), pos=(0, 0, -0.4))

    def rq():
        q = rng.normal(size=4)
        return q / np.linalg.norm(q)

    def add_geoms(body, n, allow_box):
        for _ in range(n):
            kind = rng.choice(["sphere", "capsule", "box"] if allow_box else ["sphere", "capsule"])
            size = {"sphere": (rng.uniform(0.03, 0.12),), "capsule": (rng.uniform(0.03, 0.08), rng.uniform(0.02, 0.2)),
                    "box": tuple(rng.uniform(0.03, 0.15, 3))}[kind]
            pos = rng.uniform(-0.15, 0.15, 3) if rng.random() < 0.7 else (0, 0, 0)
            quat = rq() if rng.random() < 0.7 else (1, 0, 0, 0)
            mb.add_geom(body, kind, size, pos=pos, quat=quat, contype=int(rng.choice([1, 1, 1, 2, 3])),
                        conaffinity=int(rng.choice([1, 1, 1, 2, 3])), margin=float(rng.choice([0, 0, 0, 0.01, 0.03])))

    add_geoms("world", int(rng.integers(2, 7)), allow_box=True)
    for g in mb.world.geoms[1:]:
        g["pos"] = rng.uniform(-0.7, 0.7, 3)
    nstatic = int(rng.integers(0, 2))
    nbody = int(rng.integers(3, 9))
    names = ["world"]
    for b in range(nbody):
        parent = names[int(rng.integers(max(0, len(names) - 3), len(names)))]
        name = f"b{b}"
        mb.add_body(name, parent, pos=rng.uniform(-0.25, 0.25, 3), quat=rq() if rng.random() < 0.8 else (1, 0, 0, 0))
        static = b < nstatic and parent == "world"
        if not static:
            for j in range(int(rng.choice([1, 1, 1, 2]))):
                jt = "hinge" if rng.random() < 0.75 else "slide"
                axis = rng.normal(size=3) if rng.random() < 0.6 else np.eye(3)[rng.integers(0, 3)]
                pos = rng.uniform(-0.1, 0.1, 3) if rng.random() < 0.4 else (0, 0, 0)
                rngj = (-2.5, 2.5) if jt == "hinge" else (-0.3, 0.3)
                mb.add_joint(name, f"j{b}_{j}", jt, axis=axis, pos=pos, range=rngj, ref=float(rng.choice([0, 0, 0.2])))
        add_geoms(name, int(rng.integers(1, 3)), allow_box=moving_boxes)
        names.append(name)
    model = mb.compile()
    nallowed = int(rng.integers(0, 3))
    allowed = [(model.body_names[int(rng.integers(0, model.nbody))], model.body_names[int(rng.integers(0, model.nbody))])
               for _ in range(nallowed)]
    return model, allowed


def _fuzz_seeds():
    import os
    n = int(os.environ.get("MJPL_FUZZ_SEEDS", "48"))
    return list(range(1000, 1000 + n))


@pytest.mark.parametrize("seed", list(range(24)) + _fuzz_seeds())
def test_random_models_match_oracle(oracle_mod, seed):
    model, allowed = random_model(seed % 1000 if seed >= 1000 else seed, moving_boxes=seed < 1000)
    # (every third model with moving boxes through the immediate interpreter, which only models
    # with more than 24 stored geoms would get by themselves)
    os.environ["MJPL_FORCE_IMMEDIATE"] = "1" if seed % 3 == 2 else "0"
    try:
        e = eng_mod.Engine(model, allowed)
    finally:
        os.environ.pop("MJPL_FORCE_IMMEDIATE", None)
    orc = oracle_mod.Oracle(model, allowed)
    Q = uniform_configs(model, 4096, seed=100 + seed)
    Q[::97] = model.qpos0  # exact reference pose: the angle == 0 shortcut of mju_axisAngle2Quat
    want = orc.valid_configs(Q, nthreads=8)
    np.testing.assert_array_equal(e.check_configs(Q), want)
    np.testing.assert_array_equal(e.check_configs(np.ascontiguousarray(Q.T), layout=eng_mod.SOA), want)
    fk_g, fk_o = e.fk(Q[:512]), orc.fk(Q[:512])
    for k in fk_g:
        assert np.abs(fk_g[k] - fk_o[k]).max() < 1e-6, k
        assert np.abs(fk_g[k] - fk_o[k]).max() < 1e-12, k
    qa, qb = random_edges(model, np.arange(model.nq), 1024, seed=200 + seed, eps=0.2)
    w, wfb, _ = orc.valid_edges(qa, qb, 0.03, nthreads=8, info=True)
    g, gfb = e.check_edges(qa, qb, 0.03, first_bad=True)
    np.testing.assert_array_equal(g, w)
    np.testing.assert_array_equal(gfb, wfb)
    gi = e.check_edges(qa, qb, 0.03, interior_only=True)
    wi = np.array([orc.valid_collision_interval(a, b, 0.03) for a, b in zip(qa[:128], qb[:128])])
    np.testing.assert_array_equal(gi[:128].astype(bool), wi)


def test_unsupported_models_fail_loudly():
    mb = ModelBuilder()
    mb.add_body("a")
    mb.add_joint("a", "ja")
    mb.add_geom("a", "cylinder", (0.1, 0.1))
    mb.add_geom("world", "sphere", (0.1,), pos=(1, 0, 0))
    with pytest.raises(eng_mod.MjplError, match="unsupported type"):
        eng_mod.Engine(mb.compile())


def test_bits_output_matches_bytes():
    m = scenes.franka_p(obstacles=True)
    e = eng_mod.Engine(m)
    Q = uniform_configs(m, 10007, seed=9)
    want = e.check_configs(Q)
    dq = e.alloc(Q.nbytes).upload(Q)
    nwords = (len(Q) + 63) // 64
    dbits = e.alloc(8 * nwords)
    e.check_configs_bits_dev(dq.ptr, len(Q), eng_mod.AOS, dbits.ptr)
    words = dbits.download(np.uint64, nwords)
    bits = ((words[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).reshape(-1)[: len(Q)]
    np.testing.assert_array_equal(bits.astype(np.uint8), want)


@pytest.mark.parametrize("cells", [1, 0])
def test_nearest_neighbour_kernel(cells):
    """(cells: the cell-ordered scan of big trees, mjpl_nearest_cells.h -- on by default -- or the full scan; the answers
    are the same, bit for bit)"""
    m = scenes.franka_p()
    e = eng_mod.Engine(m)
    e.set_option("nn_cells", cells)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    e.set_planning(qidx, m.keyframe("home").qpos)
    rng = np.random.default_rng(4)
    n, cap, M = 5000, 8192, 777
    nodes = np.zeros((7, cap))
    nodes[:, :n] = rng.uniform(-2, 2, size=(7, n))
    nodes[:, 17] = np.inf  # a sink node is never nearest
    queries = rng.uniform(-2, 2, size=(7, M))
    queries[:, 5] = nodes[:, 123]
    dn, dq = e.alloc(nodes.nbytes).upload(nodes), e.alloc(queries.nbytes).upload(queries)
    di, dd = e.alloc(4 * M), e.alloc(8 * M)
    e.nearest_dev(dn.ptr, n, cap, dq.ptr, M, di.ptr, dd.ptr)
    idx, d2 = di.download(np.int32, M), dd.download(np.float64, M)
    ref = ((nodes[:, None, :n] - queries[:, :, None]) ** 2)
    ref = np.where(np.isfinite(ref), ref, np.inf).sum(0)
    np.testing.assert_array_equal(idx, ref.argmin(1))
    assert idx[5] == 123 and d2[5] == 0.0
    # ties go to the lowest index also across node chunks; tiny and ragged sizes
    for n2, M2 in ((1, 3), (127, 1), (129, 300), (40000, 513)):
        nodes2 = rng.uniform(-2, 2, size=(7, n2))
        if n2 > 100:
            nodes2[:, n2 - 1] = nodes2[:, 3]      # an exact duplicate far away in index
            nodes2[:, n2 // 2] = nodes2[:, 3]
        q2 = rng.uniform(-2, 2, size=(7, M2))
        q2[:, 0] = nodes2[:, min(3, n2 - 1)]
        dn2, dq2 = e.alloc(nodes2.nbytes).upload(nodes2), e.alloc(q2.nbytes).upload(q2)
        di2 = e.alloc(4 * M2)
        e.nearest_dev(dn2.ptr, n2, n2, dq2.ptr, M2, di2.ptr)
        got = di2.download(np.int32, M2)
        want = ((nodes2[:, None, :] - q2[:, :, None]) ** 2).sum(0).argmin(1)
        np.testing.assert_array_equal(got, want)
        assert got[0] == min(3, n2 - 1)
    # large trees and query sets: the first 16 384 nodes exactly, then a binary32 screen in front of the
    # float64 distances of the rest -- same winners, also among near-ties far below binary32's
    # resolution, exact duplicates (lowest index, on either side of the seed range), infinite and huge nodes
    n3, M3 = 300000, 16500
    nodes3 = rng.uniform(-2.9, 2.9, size=(7, n3))
    q3 = rng.uniform(-2.9, 2.9, size=(7, M3))
    for j in range(0, 600):  # a cluster of nodes within 1e-9 .. 1e-13 of each other near the query's nearest one
        k = int(rng.integers(0, n3 - 40))
        q3[:, j] = nodes3[:, k] + rng.normal(scale=0.05, size=7)
        for r in range(1, 12):
            nodes3[:, k + 3 * r] = nodes3[:, k] + rng.normal(scale=10.0 ** -rng.integers(9, 14), size=7)
    nodes3[:, 260001] = nodes3[:, 11]    # exact duplicates: the lowest index wins (seed range vs screened range)
    q3[:, 700] = nodes3[:, 11]
    nodes3[:, 299000] = nodes3[:, 150000]  # ... and within the screened range
    q3[:, 702] = nodes3[:, 150000]
    nodes3[:, 5] = np.inf
    nodes3[:, 6] = 1e25                  # (finite, beyond what the screen may square)
    nodes3[:, 200006] = 1e25
    q3[3, 701] = 1e22
    dn3, dq3 = e.alloc(nodes3.nbytes).upload(nodes3), e.alloc(q3.nbytes).upload(q3)
    di3, dd3 = e.alloc(4 * M3), e.alloc(8 * M3)
    e.nearest_dev(dn3.ptr, n3, n3, dq3.ptr, M3, di3.ptr, dd3.ptr)
    got3, gd3 = di3.download(np.int32, M3), dd3.download(np.float64, M3)
    sel = np.concatenate([np.arange(0, 800), rng.integers(800, M3, 200)])
    for lo in range(0, len(sel), 100):
        js = sel[lo:lo + 100]
        with np.errstate(over="ignore", invalid="ignore"):
            s3 = np.zeros((len(js), n3))
            for c in range(7):  # the kernel's sum order
                d = nodes3[c][None, :] - q3[c][js][:, None]
                s3 = s3 + d * d
        np.testing.assert_array_equal(got3[js], s3.argmin(1))
        np.testing.assert_array_equal(gd3[js], s3.min(1))
    assert got3[700] == 11 and got3[702] == 150000
    assert e.nearest_last_screen() == 1  # (coordinates of 1e25: not for binary16 operands)
    # ... and with coordinates of ordinary size the screen runs on the matrix cores (binary16 operands, one
    # v_mfma per 32 nodes x 32 queries): same winners and distances, bit for bit, among the same near-ties,
    # duplicates (in one tile, in the two lanes that share a query, across chunks) and the sink node
    for bad in (6, 200006):
        nodes3[:, bad] = rng.uniform(-2.9, 2.9, size=7)
    q3[3, 701] = 0.5
    nodes3[:, 150004] = nodes3[:, 150000]  # rows 0 and 4 of one tile of eight: the two lanes of a query
    nodes3[:, 40] = 1e-6 * rng.normal(size=7)  # below binary16's normal range
    q3[:, 703] = 2e-6 * rng.normal(size=7)
    dn3.upload(nodes3)
    dq3.upload(q3)
    for env_on in (True, False):
        e.nearest_dev(dn3.ptr, n3, n3, dq3.ptr, M3, di3.ptr, dd3.ptr)
        got4, gd4 = di3.download(np.int32, M3), dd3.download(np.float64, M3)
        assert e.nearest_last_screen() == 2 and e.get_option("nn_last_cells") == cells
        for lo in range(0, len(sel), 100):
            js = sel[lo:lo + 100]
            s3 = np.zeros((len(js), n3))
            for c in range(7):
                d = nodes3[c][None, :] - q3[c][js][:, None]
                s3 = s3 + d * d
            np.testing.assert_array_equal(got4[js], s3.argmin(1))
            np.testing.assert_array_equal(gd4[js], s3.min(1))
        assert got4[700] == 11 and got4[702] == 150000
        # queries that are not a multiple of the workgroup's 512, nodes not a multiple of 32
        n3, M3 = n3 - 13, M3 - 77
        nodes3 = np.ascontiguousarray(nodes3[:, :n3])
        q3 = np.ascontiguousarray(q3[:, :M3])
        sel = sel[sel < M3]
        dn3, dq3 = e.alloc(nodes3.nbytes).upload(nodes3), e.alloc(q3.nbytes).upload(q3)


@pytest.mark.parametrize("cells", [1, 0])
@pytest.mark.parametrize("scale,screen", [(30.0, 2), (96.0, 2), (110.0, 1), (250.0, 1)])
def test_nearest_neighbour_large_coordinates(scale, screen, cells):
    """Coordinates below the 256 limit of the matrix-core screen but whose squared norms leave binary16's range
    (65504: seven coordinates of 97, one of 256): the node norm travels as two binary16 numbers, and a node beyond
    that range would turn into inf - inf = NaN inside the instruction and never pass the screen.  Such a call
    must be served by the binary32 screen; either way the winners are the float64 scan's."""
    m = scenes.franka_p()
    e = eng_mod.Engine(m)
    e.set_planning(scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS), m.keyframe("home").qpos)
    e.set_option("nn_cells", cells)
    rng = np.random.default_rng(int(scale))
    n, M = 270000, 16400
    nodes = rng.uniform(0.3 * scale, scale, size=(7, n)) * rng.choice([-1.0, 1.0], size=(7, n))
    # with the largest scales EVERY node's squared norm is beyond binary16: nothing would pass a broken screen
    qs = nodes[:, rng.integers(0, n, M)] + rng.normal(scale=0.02 * scale, size=(7, M))
    qs = np.clip(qs, -scale, scale)
    dn, dq = e.alloc(nodes.nbytes).upload(nodes), e.alloc(qs.nbytes).upload(qs)
    di, dd = e.alloc(4 * M), e.alloc(8 * M)
    e.nearest_dev(dn.ptr, n, n, dq.ptr, M, di.ptr, dd.ptr)
    got, gd = di.download(np.int32, M), dd.download(np.float64, M)
    assert e.nearest_last_screen() == screen
    assert (got >= 0).all()
    sel = rng.integers(0, M, 160)
    s = np.zeros((len(sel), n))
    for c in range(7):  # the kernel's sum order
        d = nodes[c][None, :] - qs[c][sel][:, None]
        s = s + d * d
    np.testing.assert_array_equal(got[sel], s.argmin(1))
    np.testing.assert_array_equal(gd[sel], s.min(1))


@pytest.mark.parametrize("nplan", [2, 3, 5, 6])
def test_nearest_neighbour_matrix_core_screen_for_other_planning_sets(nplan):
    """The screened scan for planning sets other than the arm's seven joints (the operand rows keep seven coordinate
    slots per half; unused ones are zero): winners and float64 distances of the plain scan, exact duplicates included."""
    m = scenes.franka_p()
    e = eng_mod.Engine(m)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)[:nplan]
    e.set_planning(qidx, m.keyframe("home").qpos)
    assert e.get_option("nn_cells") == 1  # (the default: the cell-ordered scan)
    rng = np.random.default_rng(40 + nplan)
    n, M = 280000 + 17 * nplan, 16384 + 5 * nplan
    nodes = rng.uniform(-2.9, 2.9, size=(nplan, n))
    qs = rng.uniform(-2.9, 2.9, size=(nplan, M))
    nodes[:, 270000] = nodes[:, 12]
    qs[:, 1] = nodes[:, 12]
    qs[:, 2] = nodes[:, 270000] + 1e-12
    dn, dq = e.alloc(nodes.nbytes).upload(nodes), e.alloc(qs.nbytes).upload(qs)
    di, dd = e.alloc(4 * M), e.alloc(8 * M)
    e.nearest_dev(dn.ptr, n, n, dq.ptr, M, di.ptr, dd.ptr)
    assert e.nearest_last_screen() == 2 and e.get_option("nn_last_cells") == 1
    got, gd = di.download(np.int32, M), dd.download(np.float64, M)
    sel = np.concatenate([np.arange(0, 64), rng.integers(64, M, 200)])
    s = np.zeros((len(sel), n))
    for c in range(nplan):  # the kernel's sum order
        d = nodes[c][None, :] - qs[c][sel][:, None]
        s = s + d * d
    np.testing.assert_array_equal(got[sel], s.argmin(1))
    np.testing.assert_array_equal(gd[sel], s.min(1))
    assert got[1] == 12 and gd[1] == 0.0
    e.close()


def test_dropin_constraint_and_planner_equivalence(oracle_mod):
    """The HIP CollisionConstraint dropped into the reference-shaped planner makes the same
    decisions as the CPU path: identical waypoint lists for a fixed seed (UR5e, config 1)."""
    m = scenes.ur5e()
    joints = mjpl.all_joints(m)
    gpu_cc = mjpl.CollisionConstraint(m)
    cpu_cc = OracleCollisionConstraint(m, pyoracle=oracle_mod)
    q_init = m.keyframe("home").qpos.copy()
    plans = []
    for cc in (gpu_cc, cpu_cc):
        constraints = [mjpl.JointLimitConstraint(m), cc]
        q_goal = mjpl.random_config(m, q_init, joints, 3, constraints)
        planner = mjpl.RRT(m, joints, constraints, collision_interval_check=(0.02, cc), seed=3,
                           goal_biasing_probability=0.1, max_planning_time=60.0)
        wps = planner.plan_to_config(q_init, q_goal)
        assert wps
        plans.append(mjpl.smooth_path(wps, constraints, (0.02, cc), eps=planner.epsilon, seed=3, sparse=True))
    assert len(plans[0]) == len(plans[1])
    for a, b in zip(*plans):
        np.testing.assert_array_equal(a, b)
    # scalar surface semantics (collision_constraint.py:26-33)
    q = q_init.copy()
    assert gpu_cc.valid_config(q) is True and gpu_cc.apply(None, q) is q
    q_bad = q.copy()
    q_bad[1] = 0.0  # upper arm horizontal, forearm folded into the floor/base region
    assert gpu_cc.valid_config(q_bad) == cpu_cc.valid_config(q_bad)
    with pytest.raises(ValueError, match="step_dist"):
        gpu_cc.valid_interval(q, q_bad, 0.0)


def test_parallel_birrt_hip_matches_oracle_backend(oracle_mod):
    """BASELINE config 4 shape on one GPU: the frontier planner validated by the HIP engine
    returns the same path as when validated by the CPU oracle (same seed, same samples)."""
    from mjpl_amd.planning.parallel_rrt import EdgeValidator

    class OracleValidator(EdgeValidator):
        def __init__(self, model, qidx, base):
            self.o = oracle_mod.Oracle(model, planning_qidx=qidx, qpos_base=base)

        def valid_edges(self, QA, QB, step):
            if step is None:
                return self.o.valid_configs(QB, nthreads=8).astype(bool)
            return self.o.valid_edges(QA, QB, step, nthreads=8).astype(bool)

    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    cpu = OracleValidator(m, qidx, q_init)
    gpu = mjpl.HipEdgeValidator(mjpl.CollisionConstraint(m), qidx, q_init)
    rng = np.random.default_rng(11)
    goals = []
    while len(goals) < 3:
        g = q_init.copy()
        g[qidx] = rng.uniform(m.jnt_range[qidx, 0], m.jnt_range[qidx, 1])
        if cpu.valid_edges(g[qidx][None], g[qidx][None], None)[0]:
            goals.append(g)
    for k, g in enumerate(goals):
        paths = []
        for v in (gpu, cpu):
            p = mjpl.ParallelBiRRT(m, joints, v, q_init, epsilon=0.05, interval_step=0.01, seed=k, batch=256,
                                   goal_biasing_probability=0.1, max_planning_time=120.0)
            paths.append(p.plan_to_config(q_init, g))
        assert len(paths[0]) == len(paths[1]) > 2
        for a, b in zip(*paths):
            np.testing.assert_array_equal(a, b)


def test_gpu_nearest_neighbour_in_the_planner_matches_host(oracle_mod):
    """f2: the frontier planner's NN through mjpl_nearest_dev gives the same plan as the host NN."""
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    v = mjpl.HipEdgeValidator(mjpl.CollisionConstraint(m), qidx, q_init)
    rng = np.random.default_rng(3)
    while True:
        g = q_init.copy()
        g[qidx] = rng.uniform(m.jnt_range[qidx, 0], m.jnt_range[qidx, 1])
        if v.valid_edges(g[qidx][None], g[qidx][None], None)[0]:
            break
    paths = []
    for min_nodes in (0, 1 << 30):  # always GPU NN / never
        p = mjpl.ParallelBiRRT(m, joints, v, q_init, epsilon=0.05, interval_step=0.01, seed=5, batch=128,
                               goal_biasing_probability=0.1, max_planning_time=120.0)
        p.gpu_nn_min_nodes = min_nodes
        paths.append(p.plan_to_config(q_init, g))
    assert len(paths[0]) == len(paths[1]) > 2
    for a, b in zip(*paths):
        np.testing.assert_array_equal(a, b)


def test_dense_clutter_stresses_the_candidate_queues(oracle_mod):
    """A 7-link arm with two capsules per link inside 48 tightly packed static spheres, capsules
    and boxes: most lanes pass many bounding culls, so the per-wave candidate queues fill and
    drain several times per geom, and many pairs fall into the float32 tolerance band."""
    rng = np.random.default_rng(77)
    mb = ModelBuilder()
    mb.add_geom("world", "plane", (1, 1, 0.1), pos=(0, 0, -0.35))
    for k in range(48):
        kind = ["sphere", "capsule", "box"][k % 3]
        size = {"sphere": (0.07,), "capsule": (0.04, 0.1), "box": (0.06, 0.05, 0.07)}[kind]
        q = rng.normal(size=4)
        mb.add_geom("world", kind, size, pos=rng.uniform(-0.45, 0.45, 3), quat=q / np.linalg.norm(q))
    parent = "world"
    for b in range(7):
        name = f"l{b}"
        mb.add_body(name, parent, pos=(0, 0, 0.12) if b else (0, 0, -0.3))
        mb.add_joint(name, f"j{b}", "hinge", axis=np.eye(3)[b % 3], range=(-2.6, 2.6))
        mb.add_geom(name, "capsule", (0.03, 0.05), pos=(0, 0, 0.06))
        mb.add_geom(name, "sphere" if b % 2 else "capsule", (0.035, 0.03)[: 1 if b % 2 else 2], pos=(0.03, 0, 0.02))
        parent = name
    model = mb.compile()
    e = eng_mod.Engine(model)
    orc = oracle_mod.Oracle(model)
    Q = uniform_configs(model, 20000, seed=5)
    want = orc.valid_configs(Q, nthreads=8)
    np.testing.assert_array_equal(e.check_configs(Q), want)
    assert 0 < want.sum() < len(Q)
    qa, qb = random_edges(model, np.arange(model.nq), 6000, seed=6, eps=0.15)
    w, wfb, _ = orc.valid_edges(qa, qb, 0.02, nthreads=8, info=True)
    g, gfb = e.check_edges(qa, qb, 0.02, first_bad=True)
    np.testing.assert_array_equal(g, w)
    np.testing.assert_array_equal(gfb, wfb)
    assert e.last_undecided() > 0


@pytest.mark.parametrize("moving_boxes", [False, True])
def test_far_from_the_origin_the_filter_steps_aside(oracle_mod, moving_boxes):
    """A scene 1 km away from the world origin: float32 positions are only good to ~1e-4 m there,
    so the filter must hand every configuration to the exact kernels -- verdicts stay bit-exact."""
    import dataclasses
    model, allowed = random_model(5, moving_boxes=moving_boxes)
    shift = np.array([1000.0, -800.0, 300.0])
    body_pos = model.body_pos.copy()
    body_pos[model.body_parentid == 0] += shift  # every body hanging off the world
    body_pos[0] = 0
    geom_pos = model.geom_pos.copy()
    geom_pos[model.geom_bodyid == 0] += shift    # and the world's own geoms
    far = dataclasses.replace(model, body_pos=body_pos, geom_pos=geom_pos)
    e = eng_mod.Engine(far, allowed)
    orc = oracle_mod.Oracle(far, allowed)
    Q = uniform_configs(far, 2048, seed=3)
    want = orc.valid_configs(Q, nthreads=8)
    np.testing.assert_array_equal(e.check_configs(Q), want)
    assert e.last_undecided() == len(Q)  # nothing was decided in float32
    qa, qb = random_edges(far, np.arange(far.nq), 1024, seed=4, eps=0.2)
    w, wfb, _ = orc.valid_edges(qa, qb, 0.03, nthreads=8, info=True)
    g, gfb = e.check_edges(qa, qb, 0.03, first_bad=True)
    np.testing.assert_array_equal(g, w)
    np.testing.assert_array_equal(gfb, wfb)
    # the same scene at the origin is decided by the filter
    e0 = eng_mod.Engine(model, allowed)
    e0.check_configs(Q)
    assert e0.last_undecided() < len(Q) // 4


@pytest.mark.parametrize("cells_min", [131072, 16384])
@pytest.mark.parametrize("n0,n,M", [(200000, 300000, 16500), (290000, 300100, 4200), (300000, 300000, 4100), (0, 270000, 5000), (1000, 9000, 700)])
def test_nearest_neighbour_over_a_node_range_behind_an_earlier_answer(n0, n, M, cells_min):
    """mjpl_nearest_range_dev: a tree that was scanned when it had n0 nodes has n now; scanning [n0, n) behind the earlier
    answer must give the whole scan's winners and distances -- also where a new node ties an old one exactly (the old, lower
    index wins) and where an old node ties a new one (ditto), with the earlier distances as every query's screen bound (no
    sample), through the matrix-core screen, the plain scan (few new nodes / few queries) and an empty range."""
    m = scenes.franka_p()
    e = eng_mod.Engine(m)
    e.set_planning(scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS), m.keyframe("home").qpos)
    e.set_option("nn_cells_min_nodes", cells_min)  # (16 384: the range of 100 000 new nodes takes the cell-ordered scan as well)
    rng = np.random.default_rng(n0 + n + M)
    cap = n + 77
    nodes = np.zeros((7, cap))
    nodes[:, :n] = rng.uniform(-2.9, 2.9, size=(7, n))
    qs = rng.uniform(-2.9, 2.9, size=(7, M))
    if n > n0 > 0:
        nodes[:, n - 5] = nodes[:, 17]          # a new node equal to an old one: the old one wins
        qs[:, 3] = nodes[:, 17] + 1e-9
        nodes[:, n0 + 1] = qs[:, 4]             # a new node ON a query
        qs[:, 5] = nodes[:, min(n0 - 1, 40)]    # an old node on a query
    dn, dq = e.alloc(nodes.nbytes).upload(nodes), e.alloc(qs.nbytes).upload(qs)
    di0, dd0, di, dd = e.alloc(4 * M), e.alloc(8 * M), e.alloc(4 * M), e.alloc(8 * M)
    if n0 > 0:
        e.nearest_dev(dn.ptr, n0, cap, dq.ptr, M, di0.ptr, dd0.ptr)
        e.nearest_range_dev(dn.ptr, n0, n, cap, dq.ptr, M, di.ptr, dd.ptr, di0.ptr, dd0.ptr)
    else:
        e.nearest_range_dev(dn.ptr, 0, n, cap, dq.ptr, M, di.ptr, dd.ptr)
    got, gd = di.download(np.int32, M), dd.download(np.float64, M)
    sel = np.concatenate([np.arange(0, min(M, 64)), rng.integers(0, M, 150)])
    s = np.zeros((len(sel), n))
    for c in range(7):  # the kernel's sum order
        d = nodes[c][None, :n] - qs[c][sel][:, None]
        s = s + d * d
    np.testing.assert_array_equal(got[sel], s.argmin(1))
    np.testing.assert_array_equal(gd[sel], s.min(1))
    if n > n0 > 0:
        assert got[3] == 17 and got[4] == n0 + 1 and gd[4] == 0.0 and got[5] == min(n0 - 1, 40)
    # ... and without the distances (the planner's call): the same winners
    di2 = e.alloc(4 * M)
    if n0 > 0:
        e.nearest_range_dev(dn.ptr, n0, n, cap, dq.ptr, M, di2.ptr, None, di0.ptr, dd0.ptr)
        np.testing.assert_array_equal(di2.download(np.int32, M), got)
    e.close()


@pytest.mark.parametrize("cells", [1, 0, 2])
def test_nearest_neighbour_in_a_dense_tree(cells):
    """Queries that lie ON a dense tree (the connect phase of a search that has filled its manifold): 300 000 nodes on a
    two-dimensional sheet in the seven joints, a few thousandths of a radian apart, queries = nodes, nodes moved by 1e-4,
    and points between them.  Thousands of nodes are within the screen's arithmetic allowance of every query's best -- the
    allowance is per query (its own norm, the largest node norm) -- and the winners must be the float64 scan's all the same."""
    m = scenes.franka_p()
    e = eng_mod.Engine(m)
    e.set_planning(scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS), m.keyframe("home").qpos)
    e.set_option("nn_cells", 1 if cells else 0)
    if cells == 2:  # (the cell-ordered scan with its two optional passes: bounds tightened on the home sub-chunks, the binary32 second screen)
        e.set_option("nn_home", 1)
        e.set_option("nn_second_screen", 1)
    rng = np.random.default_rng(77)
    n, M = 300000, 8192
    u, v = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    base = np.array([0.3, -0.8, 1.1, -2.0, 0.4, 2.2, -0.6])
    a1, a2 = rng.normal(size=7), rng.normal(size=7)
    a1, a2 = a1 / np.linalg.norm(a1), a2 / np.linalg.norm(a2)
    nodes = (base[:, None] + 0.9 * a1[:, None] * u[None, :] + 0.9 * a2[:, None] * v[None, :] + 0.05 * np.sin(3 * u * v)[None, :])
    pick = rng.integers(0, n, M)
    qs = nodes[:, pick].copy()
    qs[:, M // 3: 2 * M // 3] += rng.normal(scale=1e-4, size=(7, 2 * M // 3 - M // 3))
    qs[:, 2 * M // 3:] = 0.5 * (qs[:, 2 * M // 3:] + nodes[:, rng.integers(0, n, M - 2 * M // 3)])
    dn, dq = e.alloc(nodes.nbytes).upload(nodes), e.alloc(qs.nbytes).upload(qs)
    di, dd = e.alloc(4 * M), e.alloc(8 * M)
    e.nearest_dev(dn.ptr, n, n, dq.ptr, M, di.ptr, dd.ptr)
    assert e.nearest_last_screen() == 2 and e.get_option("nn_last_cells") == (1 if cells else 0)
    got, gd = di.download(np.int32, M), dd.download(np.float64, M)
    sel = np.concatenate([np.arange(0, 40), M // 3 + np.arange(0, 40), 2 * M // 3 + np.arange(0, 40), rng.integers(0, M, 120)])
    s = np.zeros((len(sel), n))
    for c in range(7):  # the kernel's sum order
        d = nodes[c][None, :] - qs[c][sel][:, None]
        s = s + d * d
    np.testing.assert_array_equal(got[sel], s.argmin(1))
    np.testing.assert_array_equal(gd[sel], s.min(1))
    e.close()


def test_cell_ordered_scan_differential_stress(monkeypatch, tmp_path):
    """tools/nn_stress.py, eight of its random cases: the cell-ordered scan against the full scan on uniform, clustered,
    duplicated and chain-like nodes, planning sets of 2 ... 7 columns, whole and ranged look-ups, a sink node -- indices and
    distances equal on every query, and a sample of them against NumPy (profiles/r06_nn_stress.txt: 32 cases)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("nn_stress", os.path.join(root, "tools", "nn_stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(sys, "argv", ["nn_stress.py", "8"])
    mod.main()
