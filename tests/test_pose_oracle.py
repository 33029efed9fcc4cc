"""Row f1 (PoseConstraint projection): the oracle's restatement pinned against the reference's
own analytic test (test/test_pose_constraint.py:16-50), against np.linalg.pinv, against a
finite-difference Jacobian and against a NumPy restatement of the apply loop
(pose_constraint.py:78-91).  CPU only."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from mjpl_amd import scenes
from mjpl_amd.lie import SE3, SO3

INF = (-np.inf, np.inf)


def make_oracle(oracle_mod, model, site, frame: SE3, bounds, **kw):
    inv = frame.inverse()
    return oracle_mod.PoseOracle(model, site, (inv.wxyz_xyz[:4], inv.wxyz_xyz[4:]), bounds, **kw)


def site_frame(oracle_mod, model, site, q) -> SE3:
    po = oracle_mod.PoseOracle(model, site, (np.array([1.0, 0, 0, 0]), np.zeros(3)), [INF] * 6)
    pos, mat = po.site_pose(q)
    return SE3.from_rotation_and_translation(SO3.from_matrix(mat), pos)


def test_translation_limit_kat(oracle_mod):
    """test_pose_constraint.py:16-50: [0.2, 0] -> [0.1, 0] (atol 1e-12); q_step 1e-5 -> None."""
    m = scenes.two_dof_ball()
    home = site_frame(oracle_mod, m, "ball_site", np.zeros(2))
    po = make_oracle(oracle_mod, m, "ball_site", home, [(-0.1, 0.1)] + [INF] * 5, q_step=np.inf)
    q = np.array([0.2, 0.0])
    assert not po.valid_config(q)
    qc = po.apply(np.zeros(2), q)
    assert qc is not None
    np.testing.assert_allclose(qc, [0.1, 0.0], rtol=0, atol=1e-12)
    assert po.valid_config(qc)
    po.set_q_step(1e-5)
    assert po.apply(np.zeros(2), q) is None


def test_lie_types_against_scipy():
    rng = np.random.default_rng(0)
    for _ in range(50):
        r = Rotation.random(random_state=rng.integers(1 << 30))
        so3 = SO3.from_matrix(r.as_matrix())
        np.testing.assert_allclose(so3.as_matrix(), r.as_matrix(), atol=1e-14)
        rpy = so3.as_rpy_radians()
        np.testing.assert_allclose([rpy.roll, rpy.pitch, rpy.yaw], r.as_euler("xyz"), atol=1e-12)
        back = SO3.from_rpy_radians(rpy.roll, rpy.pitch, rpy.yaw)
        np.testing.assert_allclose(back.as_matrix(), r.as_matrix(), atol=1e-12)
        t = rng.normal(size=3)
        T = SE3.from_rotation_and_translation(so3, t)
        I = T.inverse().multiply(T)
        np.testing.assert_allclose(I.rotation().as_matrix(), np.eye(3), atol=1e-14)
        np.testing.assert_allclose(I.translation(), 0, atol=1e-14)
        v = rng.normal(size=3)
        np.testing.assert_allclose(so3.apply(v), r.apply(v), atol=1e-14)


def test_pinv_matches_numpy(oracle_mod):
    rng = np.random.default_rng(1)
    for nv in (1, 2, 3, 5, 6, 7, 9):
        for _ in range(20):
            J = rng.normal(size=(6, nv))
            A = J @ J.T
            ref = np.linalg.pinv(A)
            np.testing.assert_allclose(oracle_mod.pinv_sym6(A), ref, atol=1e-9 * max(1.0, np.abs(ref).max()))
    np.testing.assert_array_equal(oracle_mod.pinv_sym6(np.zeros((6, 6))), np.zeros((6, 6)))


def _numpy_displacement(po, frame_inv: SE3, bounds, q):
    pos, mat = po.site_pose(q)
    world_T_site = SE3.from_rotation_and_translation(SO3.from_matrix(mat), pos)
    c_T_site = frame_inv.multiply(world_T_site)
    rpy = c_T_site.rotation().as_rpy_radians()
    d = np.concatenate([c_T_site.translation(), [rpy.roll, rpy.pitch, rpy.yaw]])
    b = np.asarray(bounds, dtype=np.float64)
    dx = np.zeros(6)
    over, under = d > b[:, 1], d < b[:, 0]
    dx[over] = d[over] - b[over, 1]
    dx[under] = d[under] - b[under, 0]
    return dx


def _fd_rpy_jacobian(po, q, h=1e-6):
    """E_rpy @ geometric Jacobian by central differences of the site pose."""
    n = len(q)
    J = np.zeros((6, n))
    pos0, mat0 = po.site_pose(q)
    for j in range(n):
        qp, qm = q.copy(), q.copy()
        qp[j] += h
        qm[j] -= h
        pp, mp = po.site_pose(qp)
        pm, mm = po.site_pose(qm)
        J[:3, j] = (pp - pm) / (2 * h)
        W = (mp - mm) / (2 * h) @ mat0.T  # skew(omega)
        J[3:, j] = [W[2, 1], W[0, 2], W[1, 0]]
    rpy = SO3.from_matrix(mat0).as_rpy_radians()
    cp, cy, sp, sy = np.cos(rpy.pitch), np.cos(rpy.yaw), np.sin(rpy.pitch), np.sin(rpy.yaw)
    E = np.eye(6)
    E[3:6, 3:5] = [[cy / cp, sy / cp], [-sy, cp], [cy * (sp / cp), sy * (sp / cp)]]
    return E @ J


@pytest.mark.parametrize("scene,site", [("franka", "ee_site"), ("ur5e", "attachment_site")])
def test_displacement_jacobian_and_apply_against_numpy(oracle_mod, scene, site):
    m = scenes.franka_p(obstacles=False) if scene == "franka" else scenes.ur5e()
    q_home = m.keyframe("home").qpos.copy()
    frame = site_frame(oracle_mod, m, site, q_home)
    bounds = [INF, INF, (-0.05, 0.05), (-0.1, 0.1), (-0.1, 0.1), INF]
    po = make_oracle(oracle_mod, m, site, frame, bounds, q_step=0.5)
    inv = frame.inverse()
    rng = np.random.default_rng(5)
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    nproj = 0
    for _ in range(60):
        q = q_home + rng.normal(scale=0.05, size=m.nq)
        q = np.clip(q, lo, hi)
        np.testing.assert_allclose(po.displacement(q), _numpy_displacement(po, inv, bounds, q), atol=1e-13)
        np.testing.assert_allclose(po.jacobian(q), _fd_rpy_jacobian(po, q), atol=2e-6)
        # the apply loop with np.linalg.pinv (pose_constraint.py:78-91)
        qp, ref = q.copy(), None
        for _it in range(200):
            dx = _numpy_displacement(po, inv, bounds, qp)
            if np.linalg.norm(dx) <= 0.001:
                ref = qp
                break
            J = po.jacobian(qp)
            qp = qp - J.T @ np.linalg.pinv(J @ J.T) @ dx
            if not np.all((qp >= lo) & (qp <= hi)) or np.linalg.norm(qp - q_home) > 2 * 0.5:
                break
        got = po.apply(q_home, q)
        assert (got is None) == (ref is None)
        if got is not None:
            nproj += 1
            np.testing.assert_allclose(got, ref, atol=1e-9)
            assert po.valid_config(got)
    assert nproj >= 10


def test_rotation_limit_like_the_reference(oracle_mod):
    """test_pose_constraint.py:52-118 on the capsule UR5e: roll/pitch kept within +-0.1 of home."""
    m = scenes.ur5e()
    q_init = m.keyframe("home").qpos.copy()
    frame = site_frame(oracle_mod, m, "attachment_site", q_init)
    init_rpy = frame.rotation().as_rpy_radians()
    lim = (-0.1, 0.1)
    po = make_oracle(oracle_mod, m, "attachment_site", frame, [INF, INF, INF, lim, lim, INF], q_step=np.inf)
    rng = np.random.default_rng(123)
    done = 0
    for _ in range(200):
        q_rand = rng.uniform(m.jnt_range[:, 0], m.jnt_range[:, 1])
        if po.valid_config(q_rand):
            continue
        qc = po.apply(q_init, q_rand)
        if qc is None:
            continue
        done += 1
        assert po.valid_config(qc)
        pos, mat = po.site_pose(qc)
        c = frame.inverse().multiply(SE3.from_rotation_and_translation(SO3.from_matrix(mat), pos))
        rpy = c.rotation().as_rpy_radians()
        assert lim[0] - 1e-3 <= rpy.roll <= lim[1] + 1e-3
        assert lim[0] - 1e-3 <= rpy.pitch <= lim[1] + 1e-3
        po.set_q_step(1e-5)
        assert np.linalg.norm(qc - q_init) > 1e-5
        assert po.apply(q_init, q_rand) is None
        po.set_q_step(np.inf)
        if done >= 5:
            break
    assert done >= 3
    assert np.isfinite([init_rpy.roll, init_rpy.pitch, init_rpy.yaw]).all()


def test_se3_exp_log_minus_interpolate():
    rng = np.random.default_rng(2)
    for _ in range(30):
        a = SE3.from_rotation_and_translation(
            SO3.from_matrix(Rotation.random(random_state=rng.integers(1 << 30)).as_matrix()), rng.normal(size=3))
        b = SE3.from_rotation_and_translation(
            SO3.from_matrix(Rotation.random(random_state=rng.integers(1 << 30)).as_matrix()), rng.normal(size=3))
        d = b.minus(a)
        c = a.multiply(SE3.exp(d))
        np.testing.assert_allclose(c.translation(), b.translation(), atol=1e-12)
        np.testing.assert_allclose(c.rotation().as_matrix(), b.rotation().as_matrix(), atol=1e-12)
        np.testing.assert_allclose(SE3.exp(d).log(), d, atol=1e-12)
        mid = a.interpolate(b, 0.5)
        np.testing.assert_allclose(mid.minus(a), 0.5 * d, atol=1e-12)
        w = rng.normal(size=3)
        np.testing.assert_allclose(SO3.exp(w).as_matrix(), Rotation.from_rotvec(w).as_matrix(), atol=1e-13)
    with pytest.raises(ValueError):
        a.interpolate(b, 1.5)


def test_cartesian_plan_host_logic():
    """cartesian_planner.py:11-104 with a scripted solver: interpolation counts, the
    closest-candidate pick, the empty result, the argument checks."""
    from mjpl_amd.constraint import Constraint
    from mjpl_amd.inverse_kinematics import IKSolver
    from mjpl_amd.planning.cartesian_planner import _interpolate_poses, cartesian_plan

    a = SE3.from_translation([0.0, 0.0, 0.0])
    b = SE3.from_rotation_and_translation(SO3.from_rpy_radians(0.0, 0.0, 0.25), [0.035, 0.0, 0.0])
    poses = _interpolate_poses(a, b, 0.01, 0.1)
    assert len(poses) == 5  # max(ceil(.035/.01)=4, ceil(.25/.1)=3) steps
    np.testing.assert_allclose(poses[0].wxyz_xyz, a.wxyz_xyz, atol=1e-15)
    np.testing.assert_allclose(poses[-1].wxyz_xyz, b.wxyz_xyz, atol=1e-12)
    assert len(_interpolate_poses(a, a, 0.01, 0.1)) == 2
    with pytest.raises(ValueError, match="lin_threshold"):
        _interpolate_poses(a, b, 0.0, 0.1)
    with pytest.raises(ValueError, match="ori_threshold"):
        _interpolate_poses(a, b, 0.01, -1.0)

    class Line(IKSolver):  # "IK" of a point robot: q = x, two candidates per pose
        def solve_ik(self, pose, site, q_init_guess):
            x = pose.translation()[0]
            return [np.array([x + 0.5]), np.array([x])]

    class Below(Constraint):
        def __init__(self, limit):
            self.limit = limit

        def valid_config(self, q):
            return bool(q[0] <= self.limit)

        def apply(self, q_old, q):
            return q if self.valid_config(q) else None

    far = SE3.from_translation([0.03, 0.0, 0.0])
    wps = cartesian_plan(np.array([0.0]), [a, far], "s", Line(), [Below(1.0)])
    np.testing.assert_allclose(np.array(wps).ravel(), [0.0, 0.0, 0.01, 0.02, 0.03], atol=1e-12)
    assert cartesian_plan(np.array([0.0]), [a, far], "s", Line(), [Below(0.015)]) == []
    with pytest.raises(ValueError, match="site"):
        cartesian_plan(np.array([0.0]), [a, far], "", Line(), [])
