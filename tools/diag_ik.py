"""Diagnostic (GPU box): which IK seeds do not converge, and why (joint limits, iteration budget)."""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import mjpl_amd as mjpl
from mjpl_amd import scenes
m = scenes.franka_p(obstacles=True); joints = scenes.FRANKA_ARM_JOINTS
q_home = m.keyframe("home").qpos.copy()
cc = mjpl.CollisionConstraint(m); eng = cc.engine
solver = mjpl.HipIKSolver(m, joints, [], seed=3, num_seeds=16384, iterations=200, engine=eng)
for tseed in (5, 6):
    q_t = mjpl.random_config(m, q_home, joints, tseed, [mjpl.JointLimitConstraint(m), cc])
    target = mjpl.site_pose(m, q_t, "ee_site", engine=eng)
    Q0 = solver._seeds(q_home, np.random.default_rng(3))
    for iters in (200, 1000):
        Qs, oks, its, err = eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, Q0, solver.movable, iterations=iters)
        bad = ~oks
        e = np.hypot(err[bad, 0], err[bad, 1])
        lo, hi = m.jnt_range[:7, 0], m.jnt_range[:7, 1]
        atlim = ((np.abs(Qs[bad][:, :7] - lo) < 1e-6) | (np.abs(Qs[bad][:, :7] - hi) < 1e-6)).sum(axis=1)
        print(tseed, iters, "conv", oks.mean(), "nonconv err quantiles", np.quantile(e, [0.1, 0.5, 0.9]), "joints at limit (mean)", atlim.mean(),
              "frac with >=1 at limit", (atlim > 0).mean())
