"""Dump the two trees of a configs[3]-shaped search (131 072 lanes, [PoseConstraint, JointLimit, Collision]) after a few
rounds, as float16 rows, with the next round's targets and what the lanes of the last round reached: the data
tools/nn_prune_study.py sizes the cell-ordered nearest-neighbour scan with (CPU, NumPy).  Run on the GPU box:
    python3 tools/dump_trees.py gpurun_out/trees_r6.npz 6"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mjpl_amd as mjpl  # noqa: E402
from mjpl_amd import scenes  # noqa: E402
from mjpl_amd.planning import parallel_rrt as pr  # noqa: E402


def main():
    out, rounds = sys.argv[1], int(sys.argv[2])
    L = 131072
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    cc = mjpl.CollisionConstraint(m)
    frame = mjpl.site_pose(m, q_init, "ee_site", engine=cc.engine)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), engine=cc.engine)
    cons = [pc, mjpl.JointLimitConstraint(m), cc]
    pc.q_step = np.inf
    q_goal = mjpl.random_config(m, q_init, joints, 7, cons)
    pc.q_step = 0.05
    dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=3, goal_biasing_probability=0.05,
                           batch=L, capacity=1 << 23, pose=pc)
    dev.rrt.reset(q_init[qidx], q_goal[qidx][None], 3)
    for _ in range(rounds):
        info = dev.rrt.round()
        print("round", info.round, "nodes", info.nodes[0], info.nodes[1], flush=True)
    Q0, _ = dev.rrt.tree(0)
    Q1, _ = dev.rrt.tree(1)
    lo, hi = m.jnt_range[qidx, 0], m.jnt_range[qidx, 1]
    T, _ = pr.sample_targets(pr.rrt_key(3, 0, rounds + 1), L, lo, hi, 0.05, rounds % 2, q_init[qidx], q_goal[qidx][None])
    np.savez_compressed(out, Q0=Q0.astype(np.float16), Q1=Q1.astype(np.float16), T=T.astype(np.float32), lo=lo, hi=hi)
    print("saved", out, Q0.shape, Q1.shape)


if __name__ == "__main__":
    main()
