"""Model helpers used by the planners (reference src/mjpl/utils.py:10-107)."""
from __future__ import annotations

import numpy as np

from .constraint.constraint_interface import Constraint
from .constraint.utils import apply_constraints
from .model import JNT_BALL, JNT_FREE

_QPOS_WIDTH = {JNT_FREE: 7, JNT_BALL: 4}  # slide/hinge: 1
_DOF_WIDTH = {JNT_FREE: 6, JNT_BALL: 3}


def all_joints(model) -> list[str]:
    return [model.joint(j).name for j in range(model.njnt)]


def qpos_idx(model, joints: list[str]) -> list[int]:
    idx: list[int] = []
    for name in joints:
        j = model.joint(name).id
        adr = int(model.jnt_qposadr[j])
        idx.extend(range(adr, adr + _QPOS_WIDTH.get(int(model.jnt_type[j]), 1)))
    return idx


def qvel_idx(model, joints: list[str]) -> list[int]:
    idx: list[int] = []
    for name in joints:
        j = model.joint(name).id
        adr = int(model.jnt_dofadr[j])
        idx.extend(range(adr, adr + _DOF_WIDTH.get(int(model.jnt_type[j]), 1)))
    return idx


def random_config(model, q_init: np.ndarray, joints: list[str], seed: int | None = None,
                  constraints: list[Constraint] = []) -> np.ndarray:
    """Rejection-sample a configuration that obeys ``constraints`` (:78-107).  Draws njnt
    uniforms per attempt and keeps the ``joints`` entries, like the reference, so the same
    seed gives the same configuration."""
    q_idx = qpos_idx(model, joints)
    rng = np.random.default_rng(seed=seed)
    lo, hi = model.jnt_range.T
    q = q_init.copy()
    while True:
        q[q_idx] = rng.uniform(lo, hi)[q_idx]
        out = apply_constraints(q_init, q, constraints)
        if out is not None:
            return out


def site_pose(model, q: np.ndarray, site_name: str, engine=None):
    """World pose of a site at configuration ``q`` as an :class:`mjpl_amd.lie.SE3` (reference
    ``site_pose(data, site_name)``, :60-75; there is no MjData here, so the configuration is
    passed instead of a data object).  FK runs on the GPU (``mjpl_pose_valid``)."""
    from . import engine as _engine
    from .lie import SE3, SO3
    eng = engine if engine is not None else _engine.Engine(model)
    inf = [(-np.inf, np.inf)] * 6
    proj = None
    try:
        proj = _engine.PoseProjector(eng, site_name, [1.0, 0, 0, 0], [0.0, 0, 0], inf, 0.0, np.inf)
        _, xpos, xmat = proj.valid(np.asarray(q, dtype=np.float64)[None, :], poses=True)
    finally:
        if proj is not None:
            proj.close()
        if engine is None:
            eng.close()  # a temporary engine of this call: do not leave its device memory to __del__
    return SE3.from_rotation_and_translation(SO3.from_matrix(xmat[0]), xpos[0])
