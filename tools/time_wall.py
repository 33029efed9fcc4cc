"""Wall-clock time per step without per-step events (how much does the instrumentation cost?)."""
import sys, time; sys.path.insert(0,'.')
import numpy as np
import bench
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS); base = m.keyframe("home").qpos.copy()
e = engine.Engine(m); e.set_planning(qidx, base)
E=262144
qa,qb = bench.make_edges(m,qidx,E,2)
ha,hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
dqa,dqb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb); dv = e.alloc(E)
for _ in range(5): e.check_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr)
e.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200): e.check_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr)
    e.sync()
    print("wall ms/step, no events:", (time.perf_counter() - t0) / 200 * 1e3)
ms, msk = e.time_edges_dev(dqa.ptr,dqb.ptr,E,0.01,engine.SOA,dv.ptr,50, first_kernel=True)
t0 = time.perf_counter(); ms, msk = e.time_edges_dev(dqa.ptr,dqb.ptr,E,0.01,engine.SOA,dv.ptr,50, first_kernel=True); dt = time.perf_counter() - t0
print("with events: wall ms/step", dt / 50 * 1e3, "event step ms", ms.mean())
