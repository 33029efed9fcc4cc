"""Row f3 on the GPU: batched IK seeds (k_ik_solve through the C ABI) in the role of
MinkIKSolver -- tolerance-level checks, as the reference's test asserts
(test/test_mink_ik_solver.py:16-70): pose within tolerance, constraints obeyed."""
import numpy as np
import pytest

import mjpl_amd as mjpl
from mjpl_amd import scenes
from mjpl_amd.lie import SE3, SO3

pytestmark = pytest.mark.gpu


def _pose_error(oracle_mod, m, site, q, pose: SE3):
    po = oracle_mod.PoseOracle(m, site, (np.array([1.0, 0, 0, 0]), np.zeros(3)), [(-np.inf, np.inf)] * 6)
    pos, mat = po.site_pose(q)
    e_pos = np.linalg.norm(pose.translation() - pos)
    e_ori = np.linalg.norm((pose.rotation() @ SO3.from_matrix(mat).inverse()).log())
    return e_pos, e_ori


def test_ik_like_the_reference(oracle_mod):
    """test_mink_ik_solver.py:16-70 on the capsule UR5e: target = FK of a seeded random
    configuration; solve with and without an initial guess."""
    m = scenes.ur5e()
    site = "attachment_site"
    cc = mjpl.CollisionConstraint(m)
    constraints = [mjpl.JointLimitConstraint(m), cc]
    q_init = m.keyframe("home").qpos.copy()
    rng = np.random.default_rng(seed=12345)
    q_target = rng.uniform(*m.jnt_range.T)
    target = mjpl.site_pose(m, q_target, site, engine=cc.engine)
    solver = mjpl.HipIKSolver(model=m, joints=mjpl.all_joints(m), constraints=constraints, pos_tolerance=1e-3,
                              ori_tolerance=1e-3, seed=12345, max_attempts=5, engine=cc.engine)
    sols = []
    sols.extend(solver.solve_ik(pose=target, site=site, q_init_guess=q_init)[:1])
    sols.extend(solver.solve_ik(pose=target, site=site, q_init_guess=None)[:1])
    assert len(sols) == 2
    for q in sols:
        assert mjpl.obeys_constraints(q, constraints)
        e_pos, e_ori = _pose_error(oracle_mod, m, site, q, target)
        assert e_pos <= 1e-3 and e_ori <= 1e-3
    with pytest.raises(ValueError, match="joints"):
        mjpl.HipIKSolver(m, [], engine=cc.engine)
    with pytest.raises(ValueError, match="max_attempts"):
        mjpl.HipIKSolver(m, mjpl.all_joints(m), max_attempts=0, engine=cc.engine)
    with pytest.raises(ValueError, match="iterations"):
        mjpl.HipIKSolver(m, mjpl.all_joints(m), iterations=0, engine=cc.engine)


def test_held_joints_stay_put_and_solutions_are_sorted(oracle_mod):
    m = scenes.franka_p(obstacles=True)
    site = "ee_site"
    joints = scenes.FRANKA_ARM_JOINTS
    cc = mjpl.CollisionConstraint(m)
    constraints = [mjpl.JointLimitConstraint(m), cc]
    q_init = m.keyframe("home").qpos.copy()
    q_t = mjpl.random_config(m, q_init, joints, 21, constraints)
    target = mjpl.site_pose(m, q_t, site, engine=cc.engine)
    solver = mjpl.HipIKSolver(m, joints, constraints, seed=1, max_attempts=3, num_seeds=512, engine=cc.engine)
    sols = solver.solve_ik(target, site, q_init)
    assert len(sols) >= 1, solver.stats
    d = [np.linalg.norm(q - q_init) for q in sols]
    assert d == sorted(d)
    fixed = np.setdiff1d(np.arange(m.nq), scenes.planning_index(m, joints))
    for q in sols:
        np.testing.assert_array_equal(q[fixed], q_init[fixed])
        assert mjpl.obeys_constraints(q, constraints)
        e_pos, e_ori = _pose_error(oracle_mod, m, site, q, target)
        assert e_pos <= 1e-3 and e_ori <= 1e-3


def test_config5_shape_seeds_fk_collision_filter(oracle_mod):
    """BASELINE config 5 on one GPU: 16 384 IK seeds (128k / 8 GPUs) -> FK -> collision filter.
    Size-independent properties: every reported solution is within tolerance (device-reported
    error, spot-checked against the oracle's FK), inside the joint ranges, and the collision
    filter's verdicts equal the oracle's on a sample."""
    m = scenes.franka_p(obstacles=True)
    site, joints = "ee_site", scenes.FRANKA_ARM_JOINTS
    cc = mjpl.CollisionConstraint(m)
    q_init = m.keyframe("home").qpos.copy()
    q_t = mjpl.random_config(m, q_init, joints, 5, [mjpl.JointLimitConstraint(m), cc])
    target = mjpl.site_pose(m, q_t, site, engine=cc.engine)
    solver = mjpl.HipIKSolver(m, joints, [], seed=3, num_seeds=16384, iterations=200, engine=cc.engine)
    Q0 = solver._seeds(q_init, np.random.default_rng(3))
    Q, ok, iters, err = cc.engine.ik_solve(site, target.translation(), target.rotation().wxyz, Q0,
                                            solver.movable, iterations=200, restarts=8, restart_seed=11)
    # measured on this target and these seeds: 0.977 (profiles/r02f_ik.json, target_seed 5, uniform);
    # the other targets of that file 0.81 - 0.98
    assert ok.mean() >= 0.75, ok.mean()
    assert (err[ok, 0] <= 1e-3).all() and (err[ok, 1] <= 1e-3).all()
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    assert np.all((Q >= lo - 1e-15) & (Q <= hi + 1e-15))
    valid = cc.valid_configs(Q[ok])
    assert 0 < valid.sum()
    orc = oracle_mod.Oracle(m)
    sample = np.flatnonzero(ok)[:512]
    np.testing.assert_array_equal(valid[:512], orc.valid_configs(Q[sample], nthreads=4).astype(bool))
    for i in sample[:32]:
        e_pos, e_ori = _pose_error(oracle_mod, m, site, Q[i], target)
        assert e_pos <= 1e-3 + 1e-9 and e_ori <= 1e-3 + 1e-9
        assert abs(e_pos - err[i, 0]) < 1e-9 and abs(e_ori - err[i, 1]) < 1e-9


def test_cartesian_plan_follows_a_line(oracle_mod):
    """cartesian_planner.py:44-104: a 6 cm straight-line move of the Franka-P end effector."""
    m = scenes.franka_p(obstacles=False)
    site, joints = "ee_site", scenes.FRANKA_ARM_JOINTS
    cc = mjpl.CollisionConstraint(m)
    constraints = [mjpl.JointLimitConstraint(m), cc]
    q_init = m.keyframe("home").qpos.copy()
    start = mjpl.site_pose(m, q_init, site, engine=cc.engine)
    goal = SE3.from_rotation_and_translation(start.rotation(), start.translation() + np.array([0.0, 0.06, 0.0]))
    solver = mjpl.HipIKSolver(m, joints, constraints, seed=0, max_attempts=2, num_seeds=64, engine=cc.engine)
    wps = mjpl.cartesian_plan(q_init, [start, goal], site, solver, constraints,
                              collision_interval_check=(0.01, cc))
    steps = int(np.ceil(np.linalg.norm(goal.minus(start)[:3]) / 0.01))  # 6, or 7 after rounding
    assert steps in (6, 7) and len(wps) == 1 + steps + 1  # q_init + the interpolated poses
    for k, q in enumerate(wps[1:]):
        want = start.translation() + np.array([0.0, 0.06 * k / steps, 0.0])
        e_pos, e_ori = _pose_error(oracle_mod, m, site, q, SE3.from_rotation_and_translation(start.rotation(), want))
        assert e_pos <= 1e-3 and e_ori <= 1e-3
    assert max(np.linalg.norm(b - a) for a, b in zip(wps[:-1], wps[1:])) < 0.2
    with pytest.raises(ValueError, match="site"):
        mjpl.cartesian_plan(q_init, [start, goal], "", solver, constraints)


def test_rrt_plan_to_pose_uses_the_batched_solver():
    m = scenes.ur5e()
    joints = mjpl.all_joints(m)
    cc = mjpl.CollisionConstraint(m)
    constraints = [mjpl.JointLimitConstraint(m), cc]
    q_init = m.keyframe("home").qpos.copy()
    q_goal = mjpl.random_config(m, q_init, joints, 3, constraints)
    pose = mjpl.site_pose(m, q_goal, "attachment_site", engine=cc.engine)
    planner = mjpl.RRT(m, joints, constraints, seed=3, goal_biasing_probability=0.1, max_planning_time=60.0)
    path = planner.plan_to_pose(q_init, pose, "attachment_site")
    assert len(path) >= 2
    np.testing.assert_array_equal(path[0], q_init)
    got = mjpl.site_pose(m, path[-1], "attachment_site", engine=cc.engine)
    assert np.linalg.norm(got.translation() - pose.translation()) <= 1e-3


@pytest.mark.parametrize("seed", [3, 8, 1004, 1009])
def test_ik_on_random_chains(oracle_mod, seed):
    """Slide joints, off-centre hinges, short chains (rank-deficient Jacobians): the seeds the
    kernel reports as solved reach the target (oracle FK), stay inside the joint ranges and leave
    the joints outside the chain where they were."""
    import dataclasses
    from test_gpu_models import random_model
    model, _ = random_model(seed % 1000, moving_boxes=seed < 1000)
    q = np.array([0.5, 0.5, -0.5, 0.5])
    model = dataclasses.replace(model, nsite=1, site_bodyid=np.array([model.nbody - 1], np.int32),
                                site_pos=np.array([[0.05, -0.02, 0.08]]), site_quat=q[None] / np.linalg.norm(q),
                                site_names=["tip"])
    eng = mjpl.engine.Engine(model)
    rng = np.random.default_rng(seed)
    lo, hi = model.jnt_range[:, 0], model.jnt_range[:, 1]
    q_t = rng.uniform(lo, hi) * 0.6
    target = mjpl.site_pose(model, q_t, "tip", engine=eng)
    Q0 = np.clip(q_t + rng.normal(scale=0.3, size=(512, model.nq)), lo, hi)
    movable = np.ones(model.njnt, np.uint8)
    Q, ok, iters, err = eng.ik_solve("tip", target.translation(), target.rotation().wxyz, Q0, movable, iterations=300)
    assert ok.any(), (ok.mean(), err.min(axis=0))
    assert np.all((Q >= lo - 1e-15) & (Q <= hi + 1e-15))
    for i in np.flatnonzero(ok)[:16]:
        e_pos, e_ori = _pose_error(oracle_mod, model, "tip", Q[i], target)
        assert e_pos <= 1e-3 + 1e-9 and e_ori <= 1e-3 + 1e-9
    # joints that do not move the site are never touched
    chain, b = set(), int(model.site_bodyid[0])
    while b > 0:
        chain.update(range(model.body_jntadr[b], model.body_jntadr[b] + model.body_jntnum[b]))
        b = int(model.body_parentid[b])
    off = [j for j in range(model.njnt) if j not in chain]
    np.testing.assert_array_equal(Q[:, off], Q0[:, off])


def test_restarts_lift_convergence_and_match_the_cpu_statement(oracle_mod):
    """One GPU's share of configs[4] in miniature: uniform seeds stall on joint limits (local minima
    of the clamped problem) in about half the rows; re-drawing a stalled row inside its iteration
    budget -- the reference restarts failed attempts from random_config, mink_ik_solver.py:108-115 --
    must lift convergence well above that, and the CPU statement of the same iteration
    (oracle/mjpl_oracle_pose.c) must converge on the same fraction of the same seeds."""
    import mjpl_amd as mjpl
    from mjpl_amd import scenes
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    q_home = m.keyframe("home").qpos.copy()
    cc = mjpl.CollisionConstraint(m)
    solver = mjpl.HipIKSolver(m, joints, [], seed=3, num_seeds=4096, iterations=200, engine=cc.engine)
    q_t = mjpl.random_config(m, q_home, joints, 5, [mjpl.JointLimitConstraint(m), cc])
    target = mjpl.site_pose(m, q_t, "ee_site", engine=cc.engine)
    Q0 = solver._seeds(q_home, np.random.default_rng(3))
    args = ("ee_site", target.translation(), target.rotation().wxyz, Q0, solver.movable)
    _, ok0, _, _ = cc.engine.ik_solve(*args, iterations=200)
    Qs, ok8, its, err = cc.engine.ik_solve(*args, iterations=200, restarts=8, restart_seed=11)
    assert ok8.mean() > 0.9 > ok0.mean()
    assert (err[ok8] <= 1e-3).all() and its.max() <= 200
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    assert np.all((Qs >= lo) & (Qs <= hi)) and np.all(Qs[:, 7:] == q_home[7:])
    _, okc, _, errc = oracle_mod.ik_solve_batch(m, *args, iterations=200, restarts=8, restart_seed=11, nthreads=8)
    assert abs(okc.mean() - ok8.mean()) < 0.03
    assert (okc == ok8).mean() > 0.9  # row by row they mostly agree (not bit for bit: another sin/cos)


def test_generated_chain_gives_the_interpreting_kernels_iterates_bit_for_bit(monkeypatch):
    """Franka-P's library carries the chain to the site's body as straight-line code (mjpl_project.h: IkStatic):
    the same iteration statement for statement -- every seed's final configuration, verdict, iteration count and
    error norms equal those of the interpreting kernel (MJPL_POSE_SPEC=0) bit for bit, restarts included."""
    import mjpl_amd as mjpl
    from mjpl_amd import scenes
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    q_home = m.keyframe("home").qpos.copy()
    results = {}
    for tag, env in (("generated", None), ("interpreting", "0")):
        monkeypatch.delenv("MJPL_POSE_SPEC", raising=False)
        if env is not None:
            monkeypatch.setenv("MJPL_POSE_SPEC", env)
        cc = mjpl.CollisionConstraint(m)
        assert cc.engine.spec_kind() == 1  # (all nine joints planned from qpos0: tests/spec_models.py has that library, chain included)
        solver = mjpl.HipIKSolver(m, joints, [], seed=3, num_seeds=4096, iterations=200, engine=cc.engine)
        q_t = mjpl.random_config(m, q_home, joints, 5, [mjpl.JointLimitConstraint(m), cc])
        target = mjpl.site_pose(m, q_t, "ee_site", engine=cc.engine)
        Q0 = solver._seeds(q_home, np.random.default_rng(3))
        args = ("ee_site", target.translation(), target.rotation().wxyz, Q0, solver.movable)
        results[tag] = cc.engine.ik_solve(*args, iterations=200, restarts=8, restart_seed=11)
        cc.engine.close()
    monkeypatch.delenv("MJPL_POSE_SPEC", raising=False)
    (q0, ok0, it0, e0), (q1, ok1, it1, e1) = results["generated"], results["interpreting"]
    assert ok0.mean() > 0.9 and it0.max() > 50
    np.testing.assert_array_equal(ok0, ok1)
    np.testing.assert_array_equal(it0, it1)
    np.testing.assert_array_equal(np.ascontiguousarray(q0).view(np.uint64), np.ascontiguousarray(q1).view(np.uint64))
    np.testing.assert_array_equal(np.ascontiguousarray(e0).view(np.uint64), np.ascontiguousarray(e1).view(np.uint64))


def test_quick_first_pass_returns_valid_solutions_and_falls_back(oracle_mod):
    """An attempt first runs its seeds for `quick_iterations` and spends the whole budget only if none came back valid
    (a launch lasts as long as its slowest seed).  Both ways the solutions meet the pose tolerances and the
    constraints; with a first pass too short for any seed to converge the whole budget still finds them."""
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    q_home = m.keyframe("home").qpos.copy()
    cc = mjpl.CollisionConstraint(m)
    cons = [mjpl.JointLimitConstraint(m), cc]
    q_t = mjpl.random_config(m, q_home, joints, 5, cons)
    target = mjpl.site_pose(m, q_t, "ee_site", engine=cc.engine)
    got = {}
    for quick in (64, 0, 2):
        solver = mjpl.HipIKSolver(m, joints, cons, seed=3, max_attempts=2, iterations=300, engine=cc.engine, quick_iterations=quick)
        sols = solver.solve_ik(target, "ee_site", q_init_guess=q_home)
        assert sols, quick
        for q in sols[:8]:
            assert mjpl.obeys_constraints(q, cons)
            e_pos, e_ori = _pose_error(oracle_mod, m, "ee_site", q, target)
            assert e_pos <= 1e-3 and e_ori <= 1e-3
        got[quick] = (len(sols), solver.stats["mean_iters"])
    # (two iterations converge nothing: that attempt went on to the whole budget and found what quick = 0 finds)
    assert got[2][0] == got[0][0]
    assert got[64][0] >= 1 and got[64][1] <= 64
