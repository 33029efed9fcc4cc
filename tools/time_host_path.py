"""PCIe-inclusive rate of the host-pointer entry point."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
e = engine.Engine(m); e.set_planning(qidx, m.keyframe("home").qpos.copy())
E = 262144
qa, qb = bench.make_edges(m, qidx, E, 2)
ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
e.check_edges(ha, hb, 0.01, layout=engine.SOA)
t = []
for _ in range(10):
    t0 = time.perf_counter(); e.check_edges(ha, hb, 0.01, layout=engine.SOA); t.append(time.perf_counter() - t0)
print("host-pointer mjpl_check_edges: median %.3f ms -> %.3g edges/s" % (np.median(t) * 1e3, E / np.median(t)))
