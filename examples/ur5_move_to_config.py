#!/usr/bin/env python3
"""UR5e plan-to-config + shortcutting, the script shape of the reference's
examples/ur5_move_to_config.py (:24-66) with the MI355X CollisionConstraint dropped in.

    python examples/ur5_move_to_config.py [-s SEED]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mjpl_amd as mjpl  # noqa: E402
from mjpl_amd import scenes  # noqa: E402


def main() -> bool:
    ap = argparse.ArgumentParser(description="Plan to a target configuration.")
    ap.add_argument("-s", "--seed", type=int, default=3)
    seed = ap.parse_args().seed

    model = scenes.ur5e()
    arm_joints = mjpl.all_joints(model)
    cc = mjpl.CollisionConstraint(model)
    constraints = [mjpl.JointLimitConstraint(model), cc]
    q_init = model.keyframe("home").qpos.copy()
    q_goal = mjpl.random_config(model, q_init, arm_joints, seed, constraints)

    planner = mjpl.RRT(model, arm_joints, constraints, seed=seed, goal_biasing_probability=0.1)
    print("Planning...")
    start = time.time()
    waypoints = planner.plan_to_config(q_init, q_goal)
    if not waypoints:
        print("Planning failed")
        return False
    print(f"Planning took {(time.time() - start):.4f}s ({len(waypoints)} waypoints)")

    print("Shortcutting...")
    start = time.time()
    short = mjpl.smooth_path(waypoints, constraints, eps=planner.epsilon, seed=seed, sparse=True)
    print(f"Shortcutting took {(time.time() - start):.4f}s ({len(short)} waypoints, "
          f"length {mjpl.path_length(waypoints):.3f} -> {mjpl.path_length(short):.3f})")
    return True


if __name__ == "__main__":
    sys.exit(0 if main() else 1)
