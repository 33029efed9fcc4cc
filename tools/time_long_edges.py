"""Latency of validating a few long edges (path shortcutting: smooth_path, planning/utils.py:9-87)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS); base = m.keyframe("home").qpos.copy()
rng = np.random.default_rng(0)
lo, hi = m.jnt_range[qidx, 0], m.jnt_range[qidx, 1]
for mode in ("1", "0"):
    os.environ["MJPL_EXPAND"] = mode
    e = engine.Engine(m); e.set_planning(qidx, base)
    for n in (1, 16):
        qa = np.repeat(base[qidx][None], n, 0); qb = qa + rng.normal(scale=0.25, size=qa.shape) * 0 + 0.3
        qb = np.clip(qb, lo, hi)
        e.check_edges(qa, qb, 0.01)
        t0 = time.perf_counter()
        for _ in range(20):
            v = e.check_edges(qa, qb, 0.01)
        dt = (time.perf_counter() - t0) / 20
        print("expand", mode, "edges", n, "waypoints/edge ~%d" % (np.linalg.norm(qb[0] - qa[0]) / 0.01), "ms %.3f" % (dt * 1e3), "valid", v.tolist()[:4], "items", e.last_items())
    e.close()
