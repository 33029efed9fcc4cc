#!/usr/bin/env python3
"""Planning-time harness in the shape of the reference's examples/benchmark.py (:26-91):
N seeded plan-to-config attempts on the Franka scene, success rate and median planning time.
Constraints = joint limits + collision, validated by the MI355X engine.

    python examples/benchmark.py [--attempts 15] [--obstacles] [--planner parallel|rrt]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mjpl_amd as mjpl  # noqa: E402
from mjpl_amd import scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--attempts", type=int, default=15)
    ap.add_argument("--obstacles", action="store_true")
    ap.add_argument("--planner", choices=["parallel", "rrt"], default="parallel")
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()

    model = scenes.franka_p(obstacles=args.obstacles)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(model, joints)
    q_init = model.keyframe("home").qpos.copy()
    cc = mjpl.CollisionConstraint(model)
    constraints = [mjpl.JointLimitConstraint(model), cc]
    validator = mjpl.HipEdgeValidator(cc, qidx, q_init)

    times, ok = [], 0
    for k in range(args.attempts):
        seed = args.seed + k
        q_goal = mjpl.random_config(model, q_init, joints, seed, constraints)
        t0 = time.time()
        if args.planner == "parallel":
            planner = mjpl.ParallelBiRRT(model, joints, validator, q_init, epsilon=0.05, interval_step=0.01,
                                         seed=seed, goal_biasing_probability=0.1, batch=512,
                                         max_planning_time=10.0)
            path = planner.plan_to_config(q_init, q_goal)
        else:
            planner = mjpl.RRT(model, joints, constraints, collision_interval_check=(0.01, cc), seed=seed,
                               goal_biasing_probability=0.1, max_planning_time=10.0, epsilon=0.05)
            path = planner.plan_to_config(q_init, q_goal)
        dt = time.time() - t0
        if path:
            ok += 1
            times.append(dt)
        print(f"attempt {k}: {'ok' if path else 'FAILED'} in {dt:.3f}s, {len(path)} waypoints")
    print(f"success rate {ok}/{args.attempts}; median planning time "
          f"{np.median(times) if times else float('nan'):.4f}s")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
