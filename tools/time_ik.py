"""IK seeds at one GPU's share of BASELINE configs[4] (16 384 seeds, <= 200 iterations): converged
fraction, iterations, time; several targets, uniform seeds and seeds drawn near a previous waypoint."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mjpl_amd as mjpl
from mjpl_amd import scenes
m = scenes.franka_p(obstacles=True)
joints = scenes.FRANKA_ARM_JOINTS
q_home = m.keyframe("home").qpos.copy()
cc = mjpl.CollisionConstraint(m)
eng = cc.engine
solver = mjpl.HipIKSolver(m, joints, [], seed=3, num_seeds=16384, iterations=200, engine=eng)
rows = []
for tseed in (5, 6, 7, 8):
    q_t = mjpl.random_config(m, q_home, joints, tseed, [mjpl.JointLimitConstraint(m), cc])
    target = mjpl.site_pose(m, q_t, "ee_site", engine=eng)
    Q0 = solver._seeds(q_home, np.random.default_rng(3))
    near = np.repeat(q_t[None], 16384, axis=0)
    near[:, :7] += np.random.default_rng(4).normal(scale=0.3, size=(16384, 7))
    near = np.clip(near, m.jnt_range[:, 0], m.jnt_range[:, 1])
    for tag, Q in (("uniform", Q0), ("near(0.3 rad)", near)):
        eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, Q, solver.movable, iterations=200, restarts=8, restart_seed=11)
        t0 = time.perf_counter()
        Qs, oks, its, err = eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, Q, solver.movable, iterations=200, restarts=8, restart_seed=11)
        dt = time.perf_counter() - t0
        rows.append(dict(target_seed=tseed, seeds=tag, converged=float(oks.mean()), mean_iters=float(its.mean()), ms=dt * 1e3,
                         collision_free_of_converged=float(cc.valid_configs(Qs[oks]).mean()) if oks.any() else None))
        print(rows[-1], flush=True)
json.dump(rows, open("gpurun_out/ik.json", "w"), indent=1)
