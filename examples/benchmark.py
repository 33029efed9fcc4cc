#!/usr/bin/env python3
"""Planning-time harness in the shape of the reference's examples/benchmark.py (:26-91):
`number_of_attempts` = 15 plans from the "home" keyframe to the end-effector pose of
random_config(seed), ONE seed for every attempt, `plan_to_pose` (IK inside the timed region),
constraints = joint limits + collision, no interval check, epsilon 0.05, goal bias 0.1, 10 s limit;
prints the success rate and the median planning time of the successful attempts.

    python examples/benchmark.py [--planner device|host|rrt] [--obstacles] [--interval 0.01]
                                 [--goal config] [--vary-seed] [--attempts 15] [--batch 512]

--planner device : frontier bi-RRT resident on the GPU (mjpl_amd.DeviceBiRRT)
          host   : the same algorithm in NumPy over the GPU kernels (mjpl_amd.ParallelBiRRT)
          rrt    : the reference-shaped serial CBiRRT with batched extensions (mjpl_amd.RRT)
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mjpl_amd as mjpl  # noqa: E402
from mjpl_amd import scenes  # noqa: E402


def run(planner="device", attempts=15, obstacles=True, goal="pose", interval=0.0, seed=42, vary_seed=False,
        batch=512, device=0, quiet=False, collision=None, check_path=None):
    """The reference's loop (examples/benchmark.py:28-91) -> dict(successes, attempts, planning_times, paths,
    paths_valid).  `collision`: a Constraint to validate with instead of mjpl.CollisionConstraint (the
    serial planner only: bench.py times it on the CPU oracle that way); `check_path(path) -> bool`: an
    independent check of every returned path."""
    model = scenes.franka_p(obstacles=obstacles)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(model, joints)
    q_init = model.keyframe("home").qpos.copy()
    cc = mjpl.CollisionConstraint(model, device=device)
    constraints = [mjpl.JointLimitConstraint(model), collision if collision is not None else cc]
    step = interval if interval > 0 else None
    if collision is not None and planner != "rrt":
        raise ValueError("an external collision constraint drives the serial planner only (--planner rrt)")

    planners = {}

    def planner_for(sd):
        # planners are built once per seed (device buffers, compiled model); planning is what is timed
        if sd in planners:
            return planners[sd]
        kw = dict(epsilon=0.05, seed=sd, goal_biasing_probability=0.1, max_planning_time=10.0)
        if planner == "device":
            p = mjpl.DeviceBiRRT(model, joints, cc, q_init, interval_step=step, batch=batch, capacity=1 << 21, **kw)
        elif planner == "host":
            p = mjpl.ParallelBiRRT(model, joints, mjpl.HipEdgeValidator(cc, qidx, q_init), q_init,
                                   interval_step=step, batch=batch, **kw)
        else:
            p = mjpl.RRT(model, joints, constraints,
                         collision_interval_check=(step, constraints[1]) if step else None, **kw)
        planners[sd] = p
        return p

    times, paths, ok, all_valid = [], [], 0, True
    for k in range(attempts):
        sd = seed + k if vary_seed else seed
        q_goal = mjpl.random_config(model, q_init, joints, sd, constraints)
        goal_pose = mjpl.site_pose(model, q_goal, "ee_site", engine=cc.engine)
        pl = planner_for(sd)
        solver = mjpl.HipIKSolver(model, joints, constraints, seed=sd, max_attempts=5, engine=cc.engine)
        t0 = time.time()
        if goal == "pose":
            path = pl.plan_to_pose(q_init, goal_pose, "ee_site", solver=solver)
        else:
            path = pl.plan_to_config(q_init, q_goal)
        dt = time.time() - t0
        if path:
            ok += 1
            times.append(dt)
            paths.append(path)
            if check_path is not None and not check_path(path):
                all_valid = False
        if not quiet:
            print(f"attempt {k}: {'ok' if path else 'FAILED'} in {dt:.4f}s, {len(path)} waypoints")
    return dict(successes=ok, attempts=attempts, planning_times=times, paths=paths, paths_valid=all_valid,
                model=model, qidx=qidx, q_init=q_init, interval=step)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--attempts", type=int, default=15)
    ap.add_argument("--obstacles", action="store_true")
    ap.add_argument("--planner", choices=["device", "host", "rrt"], default="device")
    ap.add_argument("--goal", choices=["pose", "config"], default="pose")
    ap.add_argument("--interval", type=float, default=0.0, help="collision_interval_check step; 0 = none (reference)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--vary-seed", action="store_true", help="seed + k for attempt k instead of one seed")
    ap.add_argument("--batch", type=int, default=512)
    args = ap.parse_args()
    res = run(planner=args.planner, attempts=args.attempts, obstacles=args.obstacles, goal=args.goal, interval=args.interval,
              seed=args.seed, vary_seed=args.vary_seed, batch=args.batch)
    times, ok = res["planning_times"], res["successes"]
    print(f"planner {args.planner}, goal {args.goal}, interval {res['interval']}: success rate {ok}/{args.attempts}; "
          f"median planning time {np.median(times) if times else float('nan'):.4f}s")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
