"""Write tests/golden/fk_*.json: forward-kinematics fixtures computed INDEPENDENTLY of both the
oracle and the HIP kernels -- homogeneous 4x4 transforms chained with scipy.spatial.transform
Rotation (matrix algebra, not MuJoCo's quaternion recurrences).  No reference test pins a
6/7-DoF FK number, so these closed-form fixtures are the anchor for a3.
"""
import json
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mjpl_amd import scenes  # noqa: E402
from mjpl_amd.model import JNT_HINGE  # noqa: E402


def rot(quat_wxyz):
    w, x, y, z = quat_wxyz
    return Rotation.from_quat([x, y, z, w]).as_matrix()


def fk(model, qpos):
    T = [np.eye(4) for _ in range(model.nbody)]
    for b in range(1, model.nbody):
        A = np.eye(4)
        A[:3, :3] = rot(model.body_quat[b])
        A[:3, 3] = model.body_pos[b]
        M = T[model.body_parentid[b]] @ A
        for j in range(model.body_jntadr[b], model.body_jntadr[b] + model.body_jntnum[b]):
            val = qpos[model.jnt_qposadr[j]] - model.qpos0[model.jnt_qposadr[j]]
            J = np.eye(4)
            if model.jnt_type[j] == JNT_HINGE:
                R = Rotation.from_rotvec(model.jnt_axis[j] * val).as_matrix()
                p = model.jnt_pos[j]
                J[:3, :3] = R
                J[:3, 3] = p - R @ p  # rotate about the anchor
            else:
                J[:3, 3] = model.jnt_axis[j] * val
            M = M @ J
        T[b] = M
    xpos = np.array([t[:3, 3] for t in T])
    xmat = np.array([t[:3, :3].reshape(9) for t in T])
    gpos, gmat = [], []
    for g in range(model.ngeom):
        A = np.eye(4)
        A[:3, :3] = rot(model.geom_quat[g])
        A[:3, 3] = model.geom_pos[g]
        M = T[model.geom_bodyid[g]] @ A
        gpos.append(M[:3, 3])
        gmat.append(M[:3, :3].reshape(9))
    return xpos, xmat, np.array(gpos), np.array(gmat)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, model in (("franka_p", scenes.franka_p(obstacles=True)), ("ur5e_c", scenes.ur5e()),
                        ("two_dof_ball", scenes.two_dof_ball())):
        rng = np.random.default_rng(7)
        qs = []
        if model.key_qpos.shape[0]:
            qs.append(model.key_qpos[0])
        qs.append(np.zeros(model.nq))
        for _ in range(6):
            qs.append(rng.uniform(model.jnt_range[:, 0], model.jnt_range[:, 1]))
        cases = []
        for q in qs:
            xpos, xmat, gpos, gmat = fk(model, np.asarray(q, float))
            cases.append(dict(qpos=list(map(float, q)), xpos=xpos.tolist(), xmat=xmat.tolist(),
                              geom_xpos=gpos.tolist(), geom_xmat=gmat.tolist()))
        with open(os.path.join(out_dir, f"fk_{name}.json"), "w") as f:
            json.dump(dict(generator="tools/make_fk_golden.py (scipy Rotation, 4x4 chain)",
                           model=name, cases=cases), f)
        print("wrote", name, len(cases), "cases")


if __name__ == "__main__":
    main()
