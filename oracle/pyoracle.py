"""ctypes binding of the CPU ORACLE (test infrastructure, NOT the product).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
module.  It wraps ``oracle/libmjpl_oracle.so`` (built by ``oracle/Makefile``), the plain-C
restatement of the reference path ``CollisionConstraint.valid_config``
(src/mjpl/constraint/collision_constraint.py:26-30) and ``_valid_collision_interval``
(src/mjpl/planning/utils.py:188-216).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmjpl_oracle.so")

_I32P = C.POINTER(C.c_int32)
_F64P = C.POINTER(C.c_double)


class _OrcModel(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("njnt", C.c_int32), ("nbody", C.c_int32), ("ngeom", C.c_int32),
        ("body_parentid", _I32P), ("body_weldid", _I32P), ("body_jntadr", _I32P),
        ("body_jntnum", _I32P), ("body_pos", _F64P), ("body_quat", _F64P),
        ("jnt_type", _I32P), ("jnt_qposadr", _I32P), ("jnt_axis", _F64P), ("jnt_pos", _F64P),
        ("qpos0", _F64P),
        ("geom_type", _I32P), ("geom_bodyid", _I32P), ("geom_contype", _I32P),
        ("geom_conaffinity", _I32P), ("geom_size", _F64P), ("geom_pos", _F64P),
        ("geom_quat", _F64P), ("geom_rbound", _F64P), ("geom_margin", _F64P),
    ]


class _OrcPose(C.Structure):
    _fields_ = [
        ("site_body", C.c_int32), ("site_pos", C.c_double * 3), ("site_quat", C.c_double * 4),
        ("c_quat", C.c_double * 4), ("c_pos", C.c_double * 3), ("lo", C.c_double * 6),
        ("hi", C.c_double * 6), ("tolerance", C.c_double), ("q_step", C.c_double),
        ("jnt_range", _F64P), ("max_iters", C.c_int32),
    ]


class _OrcBatch(C.Structure):
    _fields_ = [
        ("qpos_base", _F64P), ("qidx", _I32P), ("nplan", C.c_int32), ("layout", C.c_int32),
        ("allowed", _I32P), ("nallowed", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f))
                for f in ("mjpl_oracle.c", "mjpl_oracle_pose.c", "mjpl_oracle.h", "orc_math.h"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.run(["make", "-C", _HERE, "-B", "libmjpl_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


CPU_REF_PATH = os.path.join(_HERE, "libmjpl_cpu_ref.so")


def build_cpu_ref(force: bool = False) -> str:
    """The oracle behind include/mjpl_hip.h's host-pointer entry points (oracle/mjpl_cpu_ref.c)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f))
                for f in ("mjpl_cpu_ref.c", "mjpl_oracle.c", "mjpl_oracle_pose.c", "mjpl_oracle.h", "orc_math.h"))
    if force or not os.path.exists(CPU_REF_PATH) or os.path.getmtime(CPU_REF_PATH) < src_m:
        subprocess.run(["make", "-C", _HERE, "-B", "libmjpl_cpu_ref.so"], check=True, stdout=subprocess.DEVNULL)
    return CPU_REF_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        # MJPL_ORACLE_LIB: another build of the same sources (tools/count_flops.py loads the counting one)
        _lib = C.CDLL(os.environ.get("MJPL_ORACLE_LIB") or _LIB_PATH)
        _lib.orc_kinematics.restype = C.c_int
        _lib.orc_collision.restype = C.c_int
        _lib.orc_obeys_ruleset.restype = C.c_int
        _lib.orc_valid_config.restype = C.c_int
        _lib.orc_step.restype = None
        _lib.orc_valid_collision_interval.restype = C.c_int
        _lib.orc_valid_configs.restype = C.c_int
        _lib.orc_valid_edges.restype = C.c_int
        _lib.orc_fk_batch.restype = C.c_int
        _lib.orc_pair_test.restype = C.c_int
        for f in ("orc_site_pose", "orc_pose_displacement", "orc_pose_jacobian", "orc_pose_valid",
                  "orc_pose_apply", "orc_pose_apply_batch"):
            getattr(_lib, f).restype = C.c_int
        _lib.orc_pinv_sym6.restype = None
    return _lib


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(t)


class portable_trig:
    """Context manager: the oracle's sin/cos become the bit-reproducible fdlibm restatement
    (oracle/orc_math.h) instead of libm's.  Used by the at-threshold parity tests only: any two
    libm's (and the GPU's routine) differ in the last bit of some sin/cos, which can flip a verdict
    whose signed distance lies within ~1e-15 of zero."""

    def __enter__(self):
        self.old = lib().orc_get_trig()
        lib().orc_set_trig(1)
        return self

    def __exit__(self, *exc):
        lib().orc_set_trig(self.old)


def sorted_allowed(model, allowed_collision_bodies) -> np.ndarray:
    """np.sort(body_ids, axis=1) of collision_constraint.py:60-64."""
    if not allowed_collision_bodies:
        return np.zeros((0, 2), np.int32)
    ids = [(model.body(a).id, model.body(b).id) for a, b in allowed_collision_bodies]
    return np.sort(np.asarray(ids, dtype=np.int32), axis=1)


class Oracle:
    """Float64 CPU restatement of the collision-validation path for one model."""

    def __init__(self, model, allowed_collision_bodies=(), planning_qidx=None, qpos_base=None):
        self.model = model
        self._keep = {}
        m = _OrcModel()
        m.nq, m.njnt, m.nbody, m.ngeom = model.nq, model.njnt, model.nbody, model.ngeom
        for name, typ in _OrcModel._fields_[4:]:
            arr = getattr(model, name)
            arr = _i32(arr) if typ is _I32P else _f64(arr)
            self._keep[name] = arr
            setattr(m, name, _p(arr, typ))
        self._m = m
        self.allowed = _i32(sorted_allowed(model, list(allowed_collision_bodies)))
        self.set_planning(planning_qidx, qpos_base)

    def set_planning(self, planning_qidx=None, qpos_base=None):
        nq = self.model.nq
        self.qidx = _i32(np.arange(nq) if planning_qidx is None else planning_qidx)
        self.qbase = _f64(np.zeros(nq) if qpos_base is None else qpos_base)
        assert self.qbase.shape == (nq,)

    def _batch(self, layout):
        b = _OrcBatch()
        b.qpos_base = _p(self.qbase, _F64P)
        b.qidx = _p(self.qidx, _I32P)
        b.nplan = len(self.qidx)
        b.layout = layout
        b.allowed = _p(self.allowed, _I32P)
        b.nallowed = len(self.allowed)
        return b

    # ---- single-configuration entry points (full nq vectors, like the reference)
    def kinematics(self, qpos):
        m = self.model
        q = _f64(qpos)
        xpos, xquat = np.zeros((m.nbody, 3)), np.zeros((m.nbody, 4))
        xmat = np.zeros((m.nbody, 9))
        gx, gm = np.zeros((m.ngeom, 3)), np.zeros((m.ngeom, 9))
        st = lib().orc_kinematics(C.byref(self._m), _p(q, _F64P), _p(xpos, _F64P), _p(xquat, _F64P),
                                  _p(xmat, _F64P), _p(gx, _F64P), _p(gm, _F64P))
        if st != 0:
            raise RuntimeError(f"orc_kinematics status {st}")
        return dict(xpos=xpos, xquat=xquat, xmat=xmat, geom_xpos=gx, geom_xmat=gm)

    def contacts(self, qpos) -> np.ndarray:
        """``data.contact.geom`` after mj_kinematics + mj_collision: int32 [ncon, 2]."""
        k = self.kinematics(qpos)
        con = np.zeros((4096, 2), np.int32)
        n = C.c_int32(0)
        st = lib().orc_collision(C.byref(self._m), _p(k["geom_xpos"], _F64P),
                                 _p(k["geom_xmat"], _F64P), _p(con, _I32P), 4096, C.byref(n))
        if st != 0:
            raise RuntimeError(f"orc_collision status {st}")
        return con[: n.value].copy()

    def obeys_ruleset(self, collision_geometries) -> bool:
        cg = np.asarray(collision_geometries)
        if cg.ndim != 2 or cg.shape[1] != 2:
            raise ValueError("`collision_geometries` must be a nx2 matrix.")
        cg = _i32(cg)
        return bool(lib().orc_obeys_ruleset(C.byref(self._m), _p(cg, _I32P), len(cg),
                                            _p(self.allowed, _I32P), len(self.allowed)))

    def valid_config(self, qpos) -> bool:
        q = _f64(qpos)
        assert q.shape == (self.model.nq,)
        v = lib().orc_valid_config(C.byref(self._m), _p(self.allowed, _I32P), len(self.allowed),
                                   _p(q, _F64P))
        if v < 0:
            raise RuntimeError(f"orc_valid_config status {v}")
        return bool(v)

    def valid_collision_interval(self, start, end, step_dist, info=False):
        if step_dist <= 0.0:
            raise ValueError("`step_dist` must be > 0")
        s, e = _f64(start), _f64(end)
        nwp, bad = C.c_int32(0), C.c_int32(0)
        v = lib().orc_valid_collision_interval(
            C.byref(self._m), _p(self.allowed, _I32P), len(self.allowed), _p(s, _F64P),
            _p(e, _F64P), C.c_double(step_dist), C.byref(nwp), C.byref(bad))
        if v < 0:
            raise RuntimeError(f"orc_valid_collision_interval status {v}")
        return (bool(v), nwp.value, bad.value) if info else bool(v)

    # ---- batched entry points (planning columns)
    def valid_configs(self, Q, layout=1, nthreads=1) -> np.ndarray:
        Q = _f64(Q)
        n = Q.shape[1] if layout == 0 else Q.shape[0]
        assert Q.shape == ((len(self.qidx), n) if layout == 0 else (n, len(self.qidx)))
        out = np.zeros(n, np.uint8)
        b = self._batch(layout)
        st = lib().orc_valid_configs(C.byref(self._m), C.byref(b), _p(Q, _F64P), C.c_int64(n),
                                     nthreads, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        if st != 0:
            raise RuntimeError(f"orc_valid_configs status {st}")
        return out

    def valid_edges(self, QA, QB, step_dist, layout=1, nthreads=1, info=False):
        if step_dist <= 0.0:
            raise ValueError("`step_dist` must be > 0")
        QA, QB = _f64(QA), _f64(QB)
        assert QA.shape == QB.shape
        n = QA.shape[1] if layout == 0 else QA.shape[0]
        out = np.zeros(n, np.uint8)
        fb = np.zeros(n, np.int32)
        nc = np.zeros(n, np.int32)
        b = self._batch(layout)
        st = lib().orc_valid_edges(C.byref(self._m), C.byref(b), _p(QA, _F64P), _p(QB, _F64P),
                                   C.c_int64(n), C.c_double(step_dist), nthreads,
                                   out.ctypes.data_as(C.POINTER(C.c_uint8)), _p(fb, _I32P),
                                   _p(nc, _I32P))
        if st != 0:
            raise RuntimeError(f"orc_valid_edges status {st}")
        return (out, fb, nc) if info else out

    def fk(self, Q, layout=1):
        m = self.model
        Q = _f64(Q)
        n = Q.shape[1] if layout == 0 else Q.shape[0]
        xpos, xquat = np.zeros((n, m.nbody, 3)), np.zeros((n, m.nbody, 4))
        gx, gm = np.zeros((n, m.ngeom, 3)), np.zeros((n, m.ngeom, 9))
        b = self._batch(layout)
        st = lib().orc_fk_batch(C.byref(self._m), C.byref(b), _p(Q, _F64P), C.c_int64(n),
                                _p(xpos, _F64P), _p(xquat, _F64P), _p(gx, _F64P), _p(gm, _F64P))
        if st != 0:
            raise RuntimeError(f"orc_fk_batch status {st}")
        return dict(xpos=xpos, xquat=xquat, geom_xpos=gx, geom_xmat=gm)


def step(start, target, max_step_dist) -> np.ndarray:
    """_step (planning/utils.py:167-185) with the sequential-sum norm."""
    if max_step_dist <= 0.0:
        raise ValueError("`max_step_dist` must be > 0.0")
    s, t = _f64(start), _f64(target)
    out = np.zeros_like(s)
    lib().orc_step(_p(s, _F64P), _p(t, _F64P), len(s), C.c_double(max_step_dist), _p(out, _F64P))
    return out


def pair_test(type1, pos1, mat1, size1, type2, pos2, mat2, size2, margin=0.0) -> int:
    a = [_f64(x) for x in (pos1, np.asarray(mat1).reshape(9), size1, pos2,
                           np.asarray(mat2).reshape(9), size2)]
    return lib().orc_pair_test(int(type1), _p(a[0], _F64P), _p(a[1], _F64P), _p(a[2], _F64P),
                               int(type2), _p(a[3], _F64P), _p(a[4], _F64P), _p(a[5], _F64P),
                               C.c_double(margin))


def pinv_sym6(A) -> np.ndarray:
    """The oracle's np.linalg.pinv for a symmetric 6x6."""
    A = _f64(A).reshape(36)
    out = np.empty(36)
    lib().orc_pinv_sym6(_p(A, _F64P), _p(out, _F64P))
    return out.reshape(6, 6)


class PoseOracle:
    """Float64 CPU restatement of PoseConstraint (pose_constraint.py:11-171) for one model,
    site and constraint frame.  ``c_T_world`` is (wxyz[4], xyz[3]) of reference_frame.inverse()."""

    def __init__(self, model, site: str, c_T_world, bounds, tolerance=0.001, q_step=0.05, max_iters=1000):
        self.orc = Oracle(model)
        self.model = model
        sid = model.site(site).id
        p = _OrcPose()
        p.site_body = int(model.site_bodyid[sid])
        p.site_pos[:] = list(map(float, model.site_pos[sid]))
        p.site_quat[:] = list(map(float, model.site_quat[sid]))
        cq, cp = c_T_world
        p.c_quat[:] = list(map(float, cq))
        p.c_pos[:] = list(map(float, cp))
        b = np.asarray(bounds, dtype=np.float64).reshape(6, 2)
        p.lo[:] = list(map(float, b[:, 0]))
        p.hi[:] = list(map(float, b[:, 1]))
        p.tolerance, p.q_step, p.max_iters = float(tolerance), float(q_step), int(max_iters)
        self._rng = _f64(model.jnt_range).reshape(-1)
        p.jnt_range = _p(self._rng, _F64P)
        self._p = p

    def set_q_step(self, q_step: float):
        self._p.q_step = float(q_step)

    def site_pose(self, q):
        q = _f64(q)
        pos, mat = np.empty(3), np.empty(9)
        rc = lib().orc_site_pose(C.byref(self.orc._m), C.byref(self._p), _p(q, _F64P), _p(pos, _F64P), _p(mat, _F64P))
        assert rc == 0, rc
        return pos, mat.reshape(3, 3)

    def displacement(self, q) -> np.ndarray:
        q = _f64(q)
        dx = np.empty(6)
        rc = lib().orc_pose_displacement(C.byref(self.orc._m), C.byref(self._p), _p(q, _F64P), _p(dx, _F64P))
        assert rc == 0, rc
        return dx

    def jacobian(self, q) -> np.ndarray:
        q = _f64(q)
        J = np.empty(6 * self.model.njnt)
        rc = lib().orc_pose_jacobian(C.byref(self.orc._m), C.byref(self._p), _p(q, _F64P), _p(J, _F64P))
        assert rc == 0, rc
        return J.reshape(6, self.model.njnt)

    def valid_config(self, q) -> bool:
        q = _f64(q)
        rc = lib().orc_pose_valid(C.byref(self.orc._m), C.byref(self._p), _p(q, _F64P))
        assert rc >= 0, rc
        return bool(rc)

    def apply(self, q_old, q):
        q_old, q = _f64(q_old), _f64(q)
        out = np.empty_like(q)
        it = C.c_int32(0)
        rc = lib().orc_pose_apply(C.byref(self.orc._m), C.byref(self._p), _p(q_old, _F64P), _p(q, _F64P),
                                  _p(out, _F64P), C.byref(it))
        self.last_iters = it.value
        return out if rc == 1 else None

    def apply_batch(self, Q_old, Q, nthreads=1):
        Q_old, Q = _f64(Q_old), _f64(Q)
        n = len(Q)
        out = np.empty_like(Q)
        ok = np.zeros(n, np.uint8)
        iters = np.zeros(n, np.int32)
        rc = lib().orc_pose_apply_batch(C.byref(self.orc._m), C.byref(self._p), _p(Q_old, _F64P), _p(Q, _F64P),
                                        C.c_int64(n), C.c_int32(nthreads), _p(out, _F64P),
                                        ok.ctypes.data_as(C.POINTER(C.c_uint8)), _p(iters, _I32P))
        assert rc == 0, rc
        return out, ok.astype(bool), iters


class _OrcIK(C.Structure):
    _fields_ = [("site_body", C.c_int32), ("site_pos", C.c_double * 3), ("site_quat", C.c_double * 4),
                ("target_pos", C.c_double * 3), ("target_quat", C.c_double * 4), ("pos_tolerance", C.c_double),
                ("ori_tolerance", C.c_double), ("iterations", C.c_int32), ("damping", C.c_double),
                ("lm_damping", C.c_double), ("max_step", C.c_double), ("jnt_range", _F64P),
                ("movable", C.POINTER(C.c_uint8)), ("restarts", C.c_int32), ("restart_seed", C.c_uint64)]


def ik_solve_batch(model, site: str, target_pos, target_quat, Q, movable, pos_tolerance=1e-3, ori_tolerance=1e-3,
                   iterations=200, restarts=0, restart_seed=0, nthreads=1):
    """CPU statement of the product's IK-seed iteration (oracle/mjpl_oracle_pose.c: orc_ik_solve).
    -> (Q_out [N, nq], ok bool[N], iters int32[N], err [N, 2])"""
    orc = Oracle(model)
    sid = model.site(site).id
    d = _OrcIK()
    d.site_body = int(model.site_bodyid[sid])
    d.site_pos[:] = list(map(float, model.site_pos[sid]))
    d.site_quat[:] = list(map(float, model.site_quat[sid]))
    d.target_pos[:] = list(map(float, target_pos))
    d.target_quat[:] = list(map(float, target_quat))
    d.pos_tolerance, d.ori_tolerance, d.iterations = float(pos_tolerance), float(ori_tolerance), int(iterations)
    d.damping, d.lm_damping, d.max_step = 0.0, -1.0, 0.0
    rng = _f64(model.jnt_range).reshape(-1)
    mv = np.ascontiguousarray(movable, dtype=np.uint8)
    d.jnt_range, d.movable = _p(rng, _F64P), mv.ctypes.data_as(C.POINTER(C.c_uint8))
    d.restarts, d.restart_seed = int(restarts), int(restart_seed) & (2**64 - 1)
    Q = _f64(Q)
    n = len(Q)
    out, ok, its, err = np.empty_like(Q), np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros((n, 2))
    f = lib().orc_ik_solve_batch
    f.restype = C.c_int
    rc = f(C.byref(orc._m), C.byref(d), _p(Q, _F64P), C.c_int64(n), C.c_int32(nthreads), _p(out, _F64P),
           ok.ctypes.data_as(C.POINTER(C.c_uint8)), its.ctypes.data_as(_I32P), _p(err, _F64P))
    if rc < 0:
        raise RuntimeError(f"orc_ik_solve_batch status {rc}")
    return out, ok.astype(bool), its, err
