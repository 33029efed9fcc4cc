"""Row f1 on the GPU: mjpl_amd.PoseConstraint (k_pose_apply / k_pose_valid through the C ABI)
against the CPU oracle and the reference's analytic test (test/test_pose_constraint.py:16-50)."""
import numpy as np
import pytest

import mjpl_amd as mjpl
from mjpl_amd import scenes
from mjpl_amd.lie import SE3, SO3

pytestmark = pytest.mark.gpu
INF = (-np.inf, np.inf)


def test_translation_limit_kat_on_gpu(monkeypatch):
    m = scenes.two_dof_ball()
    home = mjpl.site_pose(m, np.zeros(2), "ball_site")
    pc = mjpl.PoseConstraint(m, "ball_site", home, x_translation=(-0.1, 0.1), q_step=np.inf)
    # (the model's library carries its chain -- two slide joints -- as straight-line code; the interpreting kernel
    #  must give the same projection bit for bit)
    assert pc._proj.spec_loaded()
    pc.engine.set_option("pose_spec", 0)  # (through the ABI: mjpl_set_option -- the library reads no environment)
    pi = mjpl.PoseConstraint(m, "ball_site", home, x_translation=(-0.1, 0.1), q_step=np.inf, engine=pc.engine)
    pc.engine.set_option("pose_spec", 1)
    assert not pi._proj.spec_loaded()
    rng = np.random.default_rng(2)
    Q = rng.uniform(-0.5, 0.5, size=(4096, 2))
    a, b = pc.apply_batch(np.zeros_like(Q), Q), pi.apply_batch(np.zeros_like(Q), Q)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8))
    assert a[1].all() and (np.abs(a[0][:, 0]) <= 0.1 + 1e-3).all()
    # ... and the IK iteration around the same chain (IkStatic, slide columns)
    movable = np.ones(m.njnt, np.uint8)
    args = ("ball_site", home.translation() + np.array([0.15, -0.2, 0.0]), home.rotation().wxyz, Q[:1024], movable)
    ik_gen = pc.engine.ik_solve(*args, iterations=100)
    pc.engine.set_option("pose_spec", 0)
    ik_int = pc.engine.ik_solve(*args, iterations=100)
    pc.engine.set_option("pose_spec", 1)
    for x, y in zip(ik_gen, ik_int):
        np.testing.assert_array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8))
    assert ik_gen[1].mean() > 0.9
    q = np.array([0.2, 0.0])
    assert not pc.valid_config(q)
    qc = pc.apply(np.array([0.0, 0.0]), q)
    assert qc is not None
    np.testing.assert_allclose(qc, [0.1, 0.0], rtol=0, atol=1e-12)
    assert pc.valid_config(qc)
    pc.q_step = 1e-5
    assert pc.apply(np.array([0.0, 0.0]), q) is None
    with pytest.raises(ValueError, match="tolerance"):
        mjpl.PoseConstraint(m, "ball_site", home, tolerance=-1.0, engine=pc.engine)
    with pytest.raises(ValueError, match="q_step"):
        mjpl.PoseConstraint(m, "ball_site", home, q_step=0.0, engine=pc.engine)
    with pytest.raises(KeyError):
        mjpl.PoseConstraint(m, "no_such_site", home, engine=pc.engine)


def _setup(oracle_mod, scene):
    if scene == "franka":
        m, site = scenes.franka_p(obstacles=False), "ee_site"
    else:
        m, site = scenes.ur5e(), "attachment_site"
    q_home = m.keyframe("home").qpos.copy()
    kw = dict(z_translation=(-0.05, 0.05), roll=(-0.1, 0.1), pitch=(-0.1, 0.1), q_step=0.5)
    eng = mjpl.engine.Engine(m)
    frame = mjpl.site_pose(m, q_home, site, engine=eng)
    pc = mjpl.PoseConstraint(m, site, frame, engine=eng, **kw)
    inv = frame.inverse()
    bounds = [INF, INF, kw["z_translation"], kw["roll"], kw["pitch"], INF]
    po = oracle_mod.PoseOracle(m, site, (inv.wxyz_xyz[:4], inv.wxyz_xyz[4:]), bounds, q_step=0.5)
    return m, q_home, pc, po


@pytest.mark.parametrize("scene", ["franka", "ur5e"])
def test_site_pose_and_validity_match_the_oracle(oracle_mod, scene):
    m, q_home, pc, po = _setup(oracle_mod, scene)
    rng = np.random.default_rng(11)
    Q = rng.uniform(m.jnt_range[:, 0], m.jnt_range[:, 1], size=(512, m.nq))
    Q[:128] = q_home + rng.normal(scale=0.01, size=(128, m.nq))
    xpos, xmat = pc.site_poses(Q)
    valid = pc.valid_configs(Q)
    for i in range(len(Q)):
        p, R = po.site_pose(Q[i])
        np.testing.assert_allclose(xpos[i], p, atol=1e-13)
        np.testing.assert_allclose(xmat[i], R, atol=1e-13)
        assert bool(valid[i]) == po.valid_config(Q[i])
    assert 0 < valid.sum() < len(Q)
    home = pc.site_pose(q_home)
    assert isinstance(home, SE3)
    np.testing.assert_allclose(home.translation(), po.site_pose(q_home)[0], atol=1e-13)


@pytest.mark.parametrize("scene", ["franka", "ur5e"])
def test_batched_projection_matches_the_oracle(oracle_mod, scene):
    m, q_home, pc, po = _setup(oracle_mod, scene)
    rng = np.random.default_rng(12)
    n = 4096
    Q = np.clip(q_home + rng.normal(scale=0.06, size=(n, m.nq)), m.jnt_range[:, 0], m.jnt_range[:, 1])
    Q_old = np.clip(q_home + rng.normal(scale=0.02, size=(n, m.nq)), m.jnt_range[:, 0], m.jnt_range[:, 1])
    got, ok, iters = pc.apply_batch(Q_old, Q)
    ref, rok, riters = po.apply_batch(Q_old, Q, nthreads=8)
    # float64 on both sides with the same operation order; sin/cos/atan2/asin differ by an ulp
    # between libm and the device library, which may flip a `<= tolerance` test in rare rows
    same = (ok == rok) & (iters == riters)
    assert same.mean() > 0.995, same.mean()
    both = same & ok
    assert both.sum() > n // 4
    np.testing.assert_allclose(got[both], ref[both], rtol=0, atol=1e-9)
    # rows that were rejected keep whatever iterate they stopped at on both sides
    rej = same & ~ok
    np.testing.assert_allclose(got[rej], ref[rej], rtol=0, atol=1e-9)
    assert (iters[both] > 0).any()


def test_generated_projection_is_the_interpreting_one_bit_for_bit(oracle_mod, monkeypatch):
    """Franka-P's libraries carry the chain to the site's body as straight-line code (mjpl_project.h: PoseStatic,
    specialise.generate_pose): the same statements in the same order -- every projected configuration, verdict and
    iteration count equals the interpreting kernel's bit for bit.  A handle made under MJPL_POSE_SPEC=0, and one on
    an engine without a library (MJPL_SPEC=0), run the interpreting kernel."""
    m = scenes.franka_p(obstacles=False)
    q_home = m.keyframe("home").qpos.copy()
    rng = np.random.default_rng(21)
    n = 30000
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    Q_old = np.clip(q_home + rng.normal(scale=0.08, size=(n, m.nq)), lo, hi)
    Q_old[:, 7:] = q_home[7:]
    d = rng.normal(size=(n, m.nq))
    d[:, 7:] = 0
    Q = np.clip(Q_old + rng.choice([0.02, 0.05, 0.2], size=(n, 1)) * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)
    kw = dict(z_translation=(-0.05, 0.05), roll=(-0.1, 0.1), pitch=(-0.1, 0.1), q_step=0.5)
    results = {}
    for tag, env in (("generated", {}), ("interpreting", {"MJPL_POSE_SPEC": "0"}), ("no library", {"MJPL_SPEC": "0"})):
        for k in ("MJPL_POSE_SPEC", "MJPL_SPEC"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = mjpl.engine.Engine(m)
        frame = mjpl.site_pose(m, q_home, "ee_site", engine=eng)
        pc = mjpl.PoseConstraint(m, "ee_site", frame, engine=eng, **kw)
        assert pc._proj.spec_loaded() == (tag == "generated"), tag
        results[tag] = pc.apply_batch(Q_old, Q)
        eng.close()
    for k in ("MJPL_POSE_SPEC", "MJPL_SPEC"):
        monkeypatch.delenv(k, raising=False)
    q0, ok0, it0 = results["generated"]
    assert 0.05 < ok0.mean() < 1.0 and it0.max() >= 3, (ok0.mean(), it0.max())
    for tag in ("interpreting", "no library"):
        q1, ok1, it1 = results[tag]
        np.testing.assert_array_equal(ok0, ok1, err_msg=tag)
        np.testing.assert_array_equal(it0, it1, err_msg=tag)
        np.testing.assert_array_equal(np.ascontiguousarray(q0).view(np.uint64), np.ascontiguousarray(q1).view(np.uint64), err_msg=tag)


def test_projection_properties_at_config4_size():
    """131 072 rows (config 4's 1 048 576 samples / 8 GPUs): every accepted row satisfies the
    constraint, the joint limits and the 2*q_step bound; projecting a projected row is a no-op."""
    m = scenes.franka_p(obstacles=False)
    q_home = m.keyframe("home").qpos.copy()
    eng = mjpl.engine.Engine(m)
    frame = mjpl.site_pose(m, q_home, "ee_site", engine=eng)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), engine=eng)
    rng = np.random.default_rng(4)
    n = 131072
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    Q_old = np.clip(q_home + rng.normal(scale=0.01, size=(n, m.nq)), lo, hi)
    d = rng.normal(size=(n, m.nq))
    d[:, 7:] = 0
    Q = np.clip(Q_old + 0.05 * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)
    out, ok, iters = pc.apply_batch(Q_old, Q)
    assert ok.mean() > 0.5
    assert (iters >= 0).all()
    assert pc.valid_configs(out[ok]).all()
    assert np.all((out[ok] >= lo) & (out[ok] <= hi))
    assert (np.linalg.norm(out[ok] - Q_old[ok], axis=1) <= 2 * pc.q_step + 1e-12).all()
    again, ok2, it2 = pc.apply_batch(Q_old[ok], out[ok])
    assert ok2.all() and (it2 == 0).all()
    np.testing.assert_array_equal(again, out[ok])


def test_constraint_composition_with_projection(oracle_mod):
    """apply_constraints([pose, limits, collision]) (constraint/utils.py:22-43) with the
    projecting constraint first, as in the reference's constrained example (:60-64)."""
    m = scenes.franka_p(obstacles=True)
    q_home = m.keyframe("home").qpos.copy()
    eng = mjpl.engine.Engine(m)
    frame = mjpl.site_pose(m, q_home, "ee_site", engine=eng)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), engine=eng)
    cons = [pc, mjpl.JointLimitConstraint(m), mjpl.CollisionConstraint(m)]
    rng = np.random.default_rng(9)
    accepted = 0
    for _ in range(30):
        q = q_home.copy()
        q[:7] += rng.normal(scale=0.03, size=7)
        out = mjpl.apply_constraints(q_home, q, cons)
        if out is not None:
            accepted += 1
            assert mjpl.obeys_constraints(out, cons)
    assert accepted >= 5


def test_frontier_planner_with_pose_projection():
    """Config-4 shape on one GPU: bi-RRT over [PoseConstraint(roll, pitch +-0.1), joint limits,
    collision]; every node of the returned path satisfies all three."""
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    cc = mjpl.CollisionConstraint(m)
    frame = mjpl.site_pose(m, q_init, "ee_site", engine=cc.engine)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), engine=cc.engine)
    cons = [pc, mjpl.JointLimitConstraint(m), cc]
    pc.q_step = np.inf
    q_goal = mjpl.random_config(m, q_init, joints, 7, cons)
    pc.q_step = 0.05
    v = mjpl.HipEdgeValidator(cc, qidx, q_init, pose_constraint=pc)
    planner = mjpl.ParallelBiRRT(m, joints, v, q_init, epsilon=0.05, interval_step=0.01, seed=7, batch=256,
                                 goal_biasing_probability=0.1, max_planning_time=120.0)
    path = planner.plan_to_config(q_init, q_goal)
    assert len(path) > 2, planner.stats
    np.testing.assert_array_equal(path[0], q_init)
    np.testing.assert_array_equal(path[-1], q_goal)
    P = np.stack(path)
    assert pc.valid_configs(P).all()
    cc.set_planning(np.arange(m.nq), m.qpos0)
    assert cc.valid_configs(P).all()
    assert (np.linalg.norm(np.diff(P, axis=0), axis=1) <= 0.05 + 2 * 0.05 + 1e-9).all()


@pytest.mark.parametrize("seed", [3, 8, 1004, 1009])
def test_projection_on_random_models(oracle_mod, seed):
    """Chains with slide joints, off-centre hinges, two joints per body and rotated site frames:
    site pose, validity and projection of the GPU kernels against the oracle."""
    from test_gpu_models import random_model
    model, _ = random_model(seed % 1000, moving_boxes=seed < 1000)
    mb_site_body = model.body_names[-1]
    # add a site to the last body by rebuilding the arrays the projection reads
    import dataclasses
    q = np.array([0.5, 0.5, -0.5, 0.5])
    model = dataclasses.replace(model, nsite=1, site_bodyid=np.array([model.nbody - 1], np.int32),
                                site_pos=np.array([[0.05, -0.02, 0.08]]), site_quat=q[None] / np.linalg.norm(q),
                                site_names=["tip"])
    assert mb_site_body == model.body_names[model.site_bodyid[0]]
    q0 = np.asarray(model.qpos0, dtype=np.float64).copy()
    eng = mjpl.engine.Engine(model)
    frame = mjpl.site_pose(model, q0, "tip", engine=eng)
    kw = dict(x_translation=(-0.02, 0.02), z_translation=(-0.03, 0.03), yaw=(-0.1, 0.1), q_step=0.4)
    pc = mjpl.PoseConstraint(model, "tip", frame, engine=eng, **kw)
    inv = frame.inverse()
    bounds = [kw["x_translation"], INF, kw["z_translation"], INF, INF, kw["yaw"]]
    po = oracle_mod.PoseOracle(model, "tip", (inv.wxyz_xyz[:4], inv.wxyz_xyz[4:]), bounds, q_step=0.4)
    rng = np.random.default_rng(seed)
    lo, hi = model.jnt_range[:, 0], model.jnt_range[:, 1]
    n = 1024
    Q = np.clip(q0 + rng.normal(scale=0.08, size=(n, model.nq)), lo, hi)
    Q_old = np.clip(q0 + rng.normal(scale=0.02, size=(n, model.nq)), lo, hi)
    xpos, xmat = pc.site_poses(Q[:64])
    for i in range(64):
        p, R = po.site_pose(Q[i])
        np.testing.assert_allclose(xpos[i], p, atol=1e-13)
        np.testing.assert_allclose(xmat[i], R, atol=1e-13)
    got, ok, iters = pc.apply_batch(Q_old, Q)
    ref, rok, riters = po.apply_batch(Q_old, Q, nthreads=8)
    same = (ok == rok) & (iters == riters)
    assert same.mean() > 0.99, same.mean()
    np.testing.assert_allclose(got[same], ref[same], rtol=0, atol=1e-8)
    assert np.array_equal(pc.valid_configs(got[same & ok]), np.ones((same & ok).sum(), bool))


def test_constrained_move_to_pose_example(monkeypatch):
    """examples/franka_constrained_move_to_pose.py end to end: PoseConstraint + joint limits +
    collision, goal from the batched IK solver, serial RRT, shortcutting."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                        "franka_constrained_move_to_pose.py")
    spec = importlib.util.spec_from_file_location("franka_constrained_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", [path, "-s", "5"])
    assert mod.main()


def test_row_kernels_give_the_same_bytes_with_one_four_and_eight_lanes_per_row(monkeypatch):
    """mjpl_rows.h: the projection and the IK seeds around a generated chain give a row 1, 4 or 8 lanes (by the batch
    size; MJPL_ROWS_G forces one).  The lanes of a row share out the statements -- which lane computes a sine changes,
    not the sine -- so projected configurations, verdicts, iteration counts, IK solutions and errors are the same bytes,
    and equal to the interpreting kernel's (MJPL_POSE_SPEC=0)."""
    m = scenes.franka_p(obstacles=False)
    q_home = m.keyframe("home").qpos.copy()
    rng = np.random.default_rng(33)
    n = 5000
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    Q_old = np.clip(q_home + rng.normal(scale=0.08, size=(n, m.nq)), lo, hi)
    Q_old[:, 7:] = q_home[7:]
    d = rng.normal(size=(n, m.nq))
    d[:, 7:] = 0
    Q = np.clip(Q_old + rng.choice([0.02, 0.05, 0.2], size=(n, 1)) * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)
    kw = dict(z_translation=(-0.05, 0.05), roll=(-0.1, 0.1), pitch=(-0.1, 0.1), q_step=0.5)
    eng = mjpl.engine.Engine(m)
    frame = mjpl.site_pose(m, q_home, "ee_site", engine=eng)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, engine=eng, **kw)
    assert pc._proj.spec_loaded()
    movable = np.zeros(m.njnt, np.uint8)
    movable[:7] = 1
    target = mjpl.site_pose(m, np.clip(q_home + 0.3, lo, hi), "ee_site", engine=eng)
    seeds = np.repeat(q_home[None], 1024, axis=0)
    seeds[:, :7] = rng.uniform(lo[:7], hi[:7], size=(1024, 7))
    out = {}
    for g in ("1", "4", "8"):
        eng.set_option("rows_g", int(g))
        out[g] = list(pc.apply_batch(Q_old, Q)) + list(eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, seeds, movable,
                                                                     iterations=120, restarts=4, restart_seed=5))
    eng.set_option("rows_g", 0)
    eng.set_option("pose_spec", 0)
    pi = mjpl.PoseConstraint(m, "ee_site", frame, engine=eng, **kw)
    assert not pi._proj.spec_loaded()
    out["interpreting"] = list(pi.apply_batch(Q_old, Q)) + list(eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, seeds, movable,
                                                                               iterations=120, restarts=4, restart_seed=5))
    eng.set_option("pose_spec", 1)
    assert 0.05 < out["1"][1].mean() < 1.0 and out["1"][4].mean() > 0.5
    for tag in ("4", "8", "interpreting"):
        for x, y in zip(out["1"], out[tag]):
            np.testing.assert_array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8), err_msg=tag)
    eng.close()
