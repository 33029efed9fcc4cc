"""cProfile of the frontier planner on the benchmark scene (host-side cost breakdown)."""
import cProfile, pstats, sys
sys.path.insert(0, '.')
import numpy as np
import mjpl_amd as mjpl
from mjpl_amd import scenes
model = scenes.franka_p(obstacles=True)
joints = scenes.FRANKA_ARM_JOINTS
qidx = scenes.planning_index(model, joints)
q_init = model.keyframe("home").qpos.copy()
cc = mjpl.CollisionConstraint(model)
cons = [mjpl.JointLimitConstraint(model), cc]
v = mjpl.HipEdgeValidator(cc, qidx, q_init)
goals = [mjpl.random_config(model, q_init, joints, 42 + k, cons) for k in range(6)]
def run():
    for k, g in enumerate(goals):
        p = mjpl.ParallelBiRRT(model, joints, v, q_init, epsilon=0.05, interval_step=0.01, seed=42 + k,
                               goal_biasing_probability=0.1, batch=512, max_planning_time=10.0)
        path = p.plan_to_config(q_init, g)
        print(k, len(path), p.stats)
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
