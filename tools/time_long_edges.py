"""Latency of validating a few long edges (path shortcutting: smooth_path, planning/utils.py:9-87): the fused kernel
(undecided waypoints rebuilt from the start of the edge), the two persistent kernels (from checkpoints) and the
walking kernel (MJPL_EXPAND=0), with the default tolerance band and with a wide one (many undecided pairs)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS); base = m.keyframe("home").qpos.copy()
rng = np.random.default_rng(0)
lo, hi = m.jnt_range[qidx, 0], m.jnt_range[qidx, 1]
for name, env in (("fused", {}), ("two kernels", {"MJPL_FUSED": "0"}), ("walking", {"MJPL_EXPAND": "0"})):
    for k in ("MJPL_FUSED", "MJPL_EXPAND"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = engine.Engine(m); e.set_planning(qidx, base)
    for tol in (None, 2e-2):
        if tol:
            e.set_filter(True, tol)
        for n, step, reach in ((1, 0.01, 0.3), (16, 0.01, 0.3), (16, 0.002, 0.9), (64, 0.0005, 1.2)):
            qa = np.repeat(base[qidx][None], n, 0) + rng.normal(scale=0.02, size=(n, len(qidx)))
            qb = np.clip(qa + reach / np.sqrt(len(qidx)), lo, hi)
            e.check_edges(qa, qb, step)
            t0 = time.perf_counter()
            for _ in range(10):
                v = e.check_edges(qa, qb, step)
            dt = (time.perf_counter() - t0) / 10
            print(f"{name:12s} tol {tol or 'default':8} edges {n:3d} waypoints/edge ~{int(np.linalg.norm(qb[0] - qa[0]) / step):5d} "
                  f"ms {dt * 1e3:8.3f} valid {int(v.sum()):3d} items {e.last_items():7d} undecided {e.last_undecided():6d}", flush=True)
    e.close()
