#!/usr/bin/env python3
"""Predict the planner's 1 -> 8 GPU curve on ONE GPU (BASELINE configs[3]).

A world of W device planners (mjpl_rrt_set_world(k, W), 131 072 lanes each, the constraints of configs[3]) runs in
lockstep on one device: every rank's round_begin (its own draws, look-ups, extensions) and round_finish (appending ALL
ranks' slabs) is the arithmetic a rank of a W-GPU job does, on the trees a W-GPU job has -- W times the nodes per
round.  Rank 0's halves are timed on the wall clock (the others run before / after it, one at a time); the exchange
between them is MODELLED: every rank receives (W - 1) slabs over its W - 1 direct xGMI links, one slab per link, at a
stated unidirectional link rate, plus a stated latency per collective (three per round: headers, rows, parents).
    python3 tools/world_model.py [--rounds 6] [--worlds 1,2,4,8] -> gpurun_out/world_model.json"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mjpl_amd as mjpl  # noqa: E402
from mjpl_amd import engine as eng_mod  # noqa: E402
from mjpl_amd import scenes  # noqa: E402
from mjpl_amd.planning import parallel_rrt as pr  # noqa: E402

LINK_GBS = 50.0        # unidirectional payload rate assumed per xGMI link (of ~153 GB/s per link both ways, MI355X_MICROARCH.md)
COLLECTIVE_US = 40.0   # latency assumed per RCCL collective on an 8-GPU node


def run_world(W, L, rounds, capacity):
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    ccs, pcs, rrts = [], [], []
    q_goal = None
    for k in range(W):
        cc = mjpl.CollisionConstraint(m)
        frame = mjpl.site_pose(m, q_init, "ee_site", engine=cc.engine)
        pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), engine=cc.engine)
        if q_goal is None:
            pc.q_step = np.inf
            q_goal = mjpl.random_config(m, q_init, joints, 7, [pc, mjpl.JointLimitConstraint(m), cc])
        pc.q_step = 0.05
        cc.set_planning(qidx, q_init)
        cc._ensure_planning()
        r = eng_mod.DeviceRRT(cc.engine, L, capacity, m.jnt_range[qidx, 0], m.jnt_range[qidx, 1], epsilon=0.05, interval_step=0.01,
                              goal_bias=0.05, seed=3, pose=pc._proj, max_steps_per_round=64)
        r.set_world(k, W)
        r.reset(q_init[qidx], q_goal[qidx][None], 3)
        ccs.append(cc); pcs.append(pc); rrts.append(r)
    nplan = len(qidx)
    out = []
    for rnd in range(1, rounds + 1):
        # ---- lockstep_round's steps, with rank 0's halves on the clock
        heads = []
        t_begin = None
        for k, r in enumerate(rrts):
            r.eng.sync()
            t0 = time.perf_counter()
            heads.append(r.round_begin(False))
            if k == 0:
                t_begin = (time.perf_counter() - t0) * 1e3
        heads = np.stack(heads)
        stride = [int(heads[:, 0].max()), int(heads[:, 1].max())]
        gathered = []
        for which in (0, 1):
            rows = np.zeros((W, stride[which], nplan))
            par = np.zeros((W, stride[which]), np.int32)
            for k, r in enumerate(rrts):
                cnt = int(heads[k, which])
                if cnt:
                    drows, dpar = r.round_slabs(which)
                    r.eng._ok(r.eng.lib.mjpl_d2h(r.eng.h, rows[k].ctypes.data, drows, cnt * nplan * 8))
                    r.eng._ok(r.eng.lib.mjpl_d2h(r.eng.h, par[k].ctypes.data, dpar, cnt * 4))
                    r.eng.sync()
            gathered.append((rows, par))
        t_finish = None
        info0 = None
        for k, r in enumerate(rrts):
            bufs = [(r.eng.alloc(g[0].nbytes).upload(g[0]), r.eng.alloc(g[1].nbytes).upload(g[1])) if stride[i] else (None, None)
                    for i, g in enumerate(gathered)]
            r.eng.sync()
            t0 = time.perf_counter()
            info = r.round_finish(heads, [b[0].ptr if b[0] else None for b in bufs], [b[1].ptr if b[1] else None for b in bufs], stride)
            r.eng.sync()
            if k == 0:
                t_finish = (time.perf_counter() - t0) * 1e3
                info0 = info
            for b in bufs:
                for x in b:
                    if x is not None:
                        x.free()
        slab_bytes = sum(stride[i] * (nplan * 8 + 4) for i in (0, 1))
        exch_ms = (slab_bytes / (LINK_GBS * 1e9)) * 1e3 + 3 * COLLECTIVE_US * 1e-3 if W > 1 else 0.0
        out.append({"round": rnd, "begin_ms": t_begin, "finish_ms": t_finish, "exchange_ms_modelled": exch_ms,
                    "slab_bytes_per_rank": slab_bytes, "nodes": [int(info0.nodes[0]), int(info0.nodes[1])],
                    "new_nodes_all_ranks": [int(info0.new_nodes[0]), int(info0.new_nodes[1])]})
        print(W, out[-1], flush=True)
    for r in rrts:
        r.close()
    for cc in ccs:
        cc.engine.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--lanes", type=int, default=131072)
    args = ap.parse_args()
    res = {"lanes_per_rank": args.lanes, "link_GBs_assumed": LINK_GBS, "collective_latency_us_assumed": COLLECTIVE_US, "worlds": {}}
    for W in [int(x) for x in args.worlds.split(",")]:
        rows = run_world(W, args.lanes, args.rounds, 1 << 24)
        timed = rows[1:]  # (round 1 grows from two single-node trees: the warm-up, as in bench.py)
        per_round = [r["begin_ms"] + r["exchange_ms_modelled"] + r["finish_ms"] for r in timed]
        res["worlds"][str(W)] = {"rounds": rows, "round_ms": per_round, "mean_round_ms": float(np.mean(per_round)),
                                 "samples_per_s_predicted": W * args.lanes / (float(np.mean(per_round)) * 1e-3)}
    base = res["worlds"].get("1", {}).get("samples_per_s_predicted")
    if base:
        for W, d in res["worlds"].items():
            d["speedup_over_one_gpu"] = d["samples_per_s_predicted"] / base
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/world_model.json", "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({W: (d["mean_round_ms"], d["samples_per_s_predicted"], d.get("speedup_over_one_gpu")) for W, d in res["worlds"].items()}))


if __name__ == "__main__":
    main()
