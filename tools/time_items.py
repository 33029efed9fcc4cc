"""What would a lane-per-waypoint interior pass cost?  Times the configuration filter on the
interior waypoints (approximated with numpy) of the bench edges whose endpoint is free."""
import sys; sys.path.insert(0,'.')
import numpy as np
import bench
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS); base = m.keyframe("home").qpos.copy()
e = engine.Engine(m); e.set_planning(qidx, base)
E = 262144
qa, qb = bench.make_edges(m, qidx, E, 2)
ok = e.check_configs(qb).astype(bool)
qa, qb = qa[ok], qb[ok]
d = qb - qa
n = np.linalg.norm(d, axis=1, keepdims=True)
W = np.concatenate([qa + d / n * (0.01 * k) for k in range(1, 5)], axis=0)
print("survivors", ok.sum(), "items", len(W))
h = np.ascontiguousarray(W.T)
dq = e.alloc(h.nbytes).upload(h); dv = e.alloc(len(W))
ms = e.time_configs_dev(dq.ptr, len(W), engine.SOA, dv.ptr, 12)
print("items ms", ms[2:].mean(), "valid frac", dv.download(np.uint8, len(W)).mean(), "undecided", e.last_undecided())
