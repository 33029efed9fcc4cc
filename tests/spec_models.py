"""The (model, planning set) combinations whose specialised filter kernels are prebuilt by
``__graft_entry__.build()`` (hipcc, no GPU) and exercised by tests/test_gpu_spec.py on the GPU box,
where nothing may be compiled from a process that has touched the GPU."""
import numpy as np

from mjpl_amd import scenes


def spec_models():
    out = []
    m = scenes.franka_p(obstacles=True)
    arm = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    out.append(("franka_p+16obs, arm planned (bench.py, BASELINE configs[2])", m, (), arm, m.keyframe("home").qpos.copy()))
    out.append(("franka_p+16obs, all nine joints planned", m, (), np.arange(m.nq, dtype=np.int32), np.asarray(m.qpos0, float).copy()))
    out.append(("franka_p+16obs, fingers allowed to touch", m, (("left_finger", "right_finger"),), arm,
                m.keyframe("home").qpos.copy()))
    m0 = scenes.franka_p(obstacles=False)
    out.append(("franka_p self-collision (BASELINE configs[1])", m0, (), np.arange(m0.nq, dtype=np.int32),
                np.asarray(m0.qpos0, float).copy()))
    u = scenes.ur5e()
    out.append(("ur5e_c", u, (), np.arange(u.nq, dtype=np.int32), np.asarray(u.qpos0, float).copy()))
    from test_gpu_models import random_model
    for seed in (1002, 1005):  # seeded random trees with slides, off-centre hinges, branching, static boxes
        rm, allowed = random_model(seed, moving_boxes=False)
        out.append((f"random_model({seed})", rm, tuple(allowed), np.arange(rm.nq, dtype=np.int32), np.asarray(rm.qpos0, float).copy()))
    return out
