"""Config-2 timing: 65 536 Franka-P configurations (self-collision + floor) per launch."""
import sys; sys.path.insert(0,'.')
import numpy as np
from mjpl_amd import engine, scenes
for obstacles in (False, True):
    m = scenes.franka_p(obstacles=obstacles); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    e = engine.Engine(m); e.set_planning(qidx, m.keyframe("home").qpos.copy())
    N = 65536
    Q = np.random.default_rng(1).uniform(m.jnt_range[qidx, 0], m.jnt_range[qidx, 1], size=(N, len(qidx)))
    h = np.ascontiguousarray(Q.T)
    dq = e.alloc(h.nbytes).upload(h); dv = e.alloc(N)
    ms = e.time_configs_dev(dq.ptr, N, engine.SOA, dv.ptr, 12)
    print("obstacles", obstacles, "ms", ms[2:].mean(), "configs/s %.3g" % (N / (ms[2:].mean() * 1e-3)), "undecided", e.last_undecided())
