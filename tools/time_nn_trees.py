#!/usr/bin/env python3
"""The nearest-neighbour look-ups of a planner round on REAL trees (tools/dump_trees.py -> tools/data/trees_r5.npz: both
trees of a configs[3] search after five rounds): the next round's 131 072 targets in the tree that grows, and 131 072
nodes of one tree in the other (the connect phase) -- with the cell-ordered scan (mjpl_nearest_cells.h) and with the full
scan, answers compared query by query, and a sample of them against NumPy.
    python tools/time_nn_trees.py [tools/data/trees_r5.npz] -> gpurun_out/nn_trees.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjpl_amd import engine as eng_mod  # noqa: E402
from mjpl_amd import scenes  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "trees_r5.npz")
    d = np.load(path)
    Q0, Q1, T = d["Q0"].astype(np.float64), d["Q1"].astype(np.float64), d["T"].astype(np.float64)
    rng = np.random.default_rng(1)
    # (float16 rows of the dump: a tiny jitter makes the nodes distinct again, as the planner's are)
    Q0 += rng.normal(scale=1e-4, size=Q0.shape)
    Q1 += rng.normal(scale=1e-4, size=Q1.shape)
    m = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    e = eng_mod.Engine(m)
    e.set_planning(qidx, m.keyframe("home").qpos.copy())
    out = {}
    cases = (("targets of the next round in the tree that grows", Q1, T),
             ("nodes of one tree in the other (connect phase)", Q0, Q1[rng.choice(len(Q1), 131072, replace=False)]))
    for label, nodes_rows, q_rows in cases:
        nodes, qs = np.ascontiguousarray(nodes_rows.T), np.ascontiguousarray(q_rows.T)
        n, M = nodes.shape[1], qs.shape[1]
        dn, dq = e.alloc(nodes.nbytes).upload(nodes), e.alloc(qs.nbytes).upload(qs)
        res = {}
        for cells in (1, 0):
            e.set_option("nn_cells", cells)
            di, dd = e.alloc(4 * M), e.alloc(8 * M)
            e.nearest_dev(dn.ptr, n, n, dq.ptr, M, di.ptr, dd.ptr)
            e.sync()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                e.nearest_dev(dn.ptr, n, n, dq.ptr, M, di.ptr, dd.ptr)
            e.sync()
            res[cells] = ((time.perf_counter() - t0) / reps * 1e3, di.download(np.int32, M), dd.download(np.float64, M))
            assert e.get_option("nn_last_cells") == cells
            if cells:
                frac = e.get_option("nn_last_candidate_fraction")
        same = bool(np.array_equal(res[1][1], res[0][1]) and np.array_equal(res[1][2], res[0][2]))
        for j in range(0, M, M // 32):  # ... and a sample against NumPy: sequential-sum squared norms, lowest index wins
            dif = nodes - qs[:, j:j + 1]
            s = np.zeros(n)
            for c in range(7):
                s = s + dif[c] * dif[c]
            assert res[1][1][j] == int(np.argmin(s)), (label, j)
        out[label] = {"nodes": n, "queries": M, "cells_ms": res[1][0], "full_ms": res[0][0], "answers_equal": same, "candidate_fraction": frac,
                      "median_distance": float(np.sqrt(np.median(res[0][2])))}
        print(label, out[label], flush=True)
        assert same
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/nn_trees.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
