"""One GPU's share of BASELINE configs[3]: rounds of the device-resident frontier bi-RRT with
131 072 samples each under [PoseConstraint(roll, pitch +-0.1), JointLimit, Collision], through a
one-rank RCCL communicator; and the unconstrained variant ([JointLimit, Collision])."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mjpl_amd as mjpl
from mjpl_amd import engine as eng_mod, scenes

L = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
m = scenes.franka_p(obstacles=True)
joints = scenes.FRANKA_ARM_JOINTS
qidx = scenes.planning_index(m, joints)
q_init = m.keyframe("home").qpos.copy()
out = {}
for tag in ("pose+limits+collision", "limits+collision"):
    cc = mjpl.CollisionConstraint(m)
    pc = None
    cons = [mjpl.JointLimitConstraint(m), cc]
    if tag.startswith("pose"):
        frame = mjpl.site_pose(m, q_init, "ee_site", engine=cc.engine)
        pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), engine=cc.engine)
        cons = [pc] + cons
        pc.q_step = np.inf
    q_goal = mjpl.random_config(m, q_init, joints, 7, cons)
    if pc is not None:
        pc.q_step = 0.05
    dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=3, goal_biasing_probability=0.05,
                           batch=L, capacity=1 << 24, pose=pc, comm=(eng_mod.comm_unique_id(), 0, 1))
    dev.rrt.reset(q_init[qidx], q_goal[qidx][None], 3)
    rows = []
    for k in range(ROUNDS):
        t0 = time.perf_counter()
        info = dev.rrt.round()
        dt = time.perf_counter() - t0
        rows.append(dict(round=k + 1, ms=dt * 1e3, new=(info.new_nodes[0], info.new_nodes[1]), nodes=(info.nodes[0], info.nodes[1]),
                         connected=int(info.connected)))
        print(tag, rows[-1], flush=True)
    tot = sum(r["ms"] for r in rows[1:]) / 1e3
    out[tag] = dict(lanes=L, rounds=rows, samples_per_s=L * (len(rows) - 1) / tot if tot > 0 else None,
                    new_nodes_per_s=sum(sum(r["new"]) for r in rows[1:]) / tot if tot > 0 else None)
    dev.rrt.close()
    cc.engine.close()
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/rrt_rounds.json", "w"), indent=1)
