#!/usr/bin/env python3
"""Constrained plan-to-pose on the primitive Franka scene, the script shape of the reference's
examples/franka_constrained_move_to_pose.py (:19-104) without trajectory generation and
visualisation (out of scope): the end effector may roll / pitch at most 0.1 rad away from its
initial pose (PoseConstraint, projected on the GPU), goals come from the batched IK solver, the
path is shortcut afterwards.

    python examples/franka_constrained_move_to_pose.py [-s SEED] [--obstacles]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mjpl_amd as mjpl  # noqa: E402
from mjpl_amd import scenes  # noqa: E402

EE_SITE = "ee_site"


def main() -> bool:
    ap = argparse.ArgumentParser(description="Plan to a goal pose under an end-effector pose constraint.")
    ap.add_argument("-s", "--seed", type=int, default=5)
    ap.add_argument("--obstacles", action="store_true")
    args = ap.parse_args()
    seed = args.seed

    model = scenes.franka_p(obstacles=args.obstacles)
    arm_joints = scenes.FRANKA_ARM_JOINTS
    q_init = model.keyframe("home").qpos.copy()

    collision = mjpl.CollisionConstraint(model)
    ee_init_pose = mjpl.site_pose(model, q_init, EE_SITE, engine=collision.engine)
    ee_pose_constraint = mjpl.PoseConstraint(model, EE_SITE, ee_init_pose, roll=(-0.1, 0.1), pitch=(-0.1, 0.1),
                                             engine=collision.engine)
    constraints = [ee_pose_constraint, mjpl.JointLimitConstraint(model), collision]

    # a goal pose derived from a valid configuration: lift q_step while sampling it (:66-75)
    q_step = ee_pose_constraint.q_step
    ee_pose_constraint.q_step = np.inf
    q_goal = mjpl.random_config(model, q_init, arm_joints, seed, constraints)
    ee_pose_constraint.q_step = q_step
    goal_pose = mjpl.site_pose(model, q_goal, EE_SITE, engine=collision.engine)

    planner = mjpl.RRT(model, arm_joints, constraints, seed=seed, goal_biasing_probability=0.1,
                       max_planning_time=60.0)
    solver = mjpl.HipIKSolver(model, arm_joints, constraints, seed=seed, max_attempts=5, engine=collision.engine)
    print("Planning...")
    t0 = time.time()
    waypoints = planner.plan_to_pose(q_init, goal_pose, EE_SITE, solver=solver)
    if not waypoints:
        print("Planning failed")
        return False
    print(f"Planning took {time.time() - t0:.4f}s ({len(waypoints)} waypoints)")

    print("Shortcutting...")
    t0 = time.time()
    short = mjpl.smooth_path(waypoints, constraints, eps=planner.epsilon, seed=seed)
    print(f"Shortcutting took {time.time() - t0:.4f}s ({len(short)} waypoints, "
          f"length {mjpl.path_length(waypoints):.3f} -> {mjpl.path_length(short):.3f})")
    ok = all(mjpl.obeys_constraints(q, constraints) for q in short)
    reached = mjpl.site_pose(model, short[-1], EE_SITE, engine=collision.engine)
    err = np.linalg.norm(reached.translation() - goal_pose.translation())
    print(f"all waypoints obey the constraints: {ok}; goal position error {err:.2e} m")
    return ok and err <= 2e-3


if __name__ == "__main__":
    sys.exit(0 if main() else 1)
