"""Diagnostic (GPU box): where a plan_to_pose call spends its time (IK, goal validation, search); cProfile."""
import time, sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
import mjpl_amd as mjpl
from mjpl_amd import scenes
model = scenes.franka_p(obstacles=True)
joints = scenes.FRANKA_ARM_JOINTS
q_init = model.keyframe("home").qpos.copy()
cc = mjpl.CollisionConstraint(model)
constraints = [mjpl.JointLimitConstraint(model), cc]
seed = 42
q_goal = mjpl.random_config(model, q_init, joints, seed, constraints)
goal_pose = mjpl.site_pose(model, q_goal, "ee_site", engine=cc.engine)
p = mjpl.DeviceBiRRT(model, joints, cc, q_init, batch=512, capacity=1 << 21, epsilon=0.05, seed=seed, goal_biasing_probability=0.1, max_planning_time=10.0)
solver = mjpl.HipIKSolver(model, joints, constraints, seed=seed, max_attempts=5, engine=cc.engine)
for k in range(3):
    t0 = time.time(); sols = solver.solve_ik(goal_pose, "ee_site", q_init); t1 = time.time()
    path = p.plan_to_configs(q_init, sols) if hasattr(p, "plan_to_configs") else None
    t2 = time.time()
    print("ik", t1 - t0, "n", len(sols), "plan", t2 - t1, len(path) if path else None)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
path = p.plan_to_pose(q_init, goal_pose, "ee_site", solver=solver)
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(18)
