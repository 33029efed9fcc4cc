#!/usr/bin/env python3
"""The sample pass's size (option nn_cells_sample) on the planner's real trees: ms per look-up and the share of sub-chunks on a
wave's list, same box.  python tools/nn_sample_ab.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjpl_amd import engine as eng_mod, scenes  # noqa: E402

d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "trees_r5.npz"))
Q0, Q1, T = d["Q0"].astype(np.float64), d["Q1"].astype(np.float64), d["T"].astype(np.float64)
rng = np.random.default_rng(1)
Q0 += rng.normal(scale=1e-4, size=Q0.shape); Q1 += rng.normal(scale=1e-4, size=Q1.shape)
m = scenes.franka_p(obstacles=True)
e = eng_mod.Engine(m)
e.set_planning(scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS), m.keyframe("home").qpos.copy())
for label, nodes_rows, q_rows in (("targets", Q1, T), ("connect", Q0, Q1[rng.choice(len(Q1), 131072, replace=False)])):
    nodes, qs = np.ascontiguousarray(nodes_rows.T), np.ascontiguousarray(q_rows.T)
    n, M = nodes.shape[1], qs.shape[1]
    dn, dq, di = e.alloc(nodes.nbytes).upload(nodes), e.alloc(qs.nbytes).upload(qs), e.alloc(4 * M)
    ref = None
    for samp in (65536, 32768, 16384, 8192, 131072):
        e.set_option("nn_cells_sample", samp)
        e.nearest_dev(dn.ptr, n, n, dq.ptr, M, di.ptr); e.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            e.nearest_dev(dn.ptr, n, n, dq.ptr, M, di.ptr)
        e.sync()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        got = di.download(np.int32, M)
        ref = got if ref is None else ref
        print(label, "sample", samp, "%.2f ms" % ms, "candidates %.4f" % e.get_option("nn_last_candidate_fraction"), "equal", bool(np.array_equal(got, ref)), flush=True)
