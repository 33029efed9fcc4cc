import sys; sys.path.insert(0,'.')
import numpy as np, csv
import bench
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS); base = m.keyframe("home").qpos.copy()
e = engine.Engine(m); e.set_planning(qidx, base)
E=262144
qa,qb = bench.make_edges(m,qidx,E,2)
ha,hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
dqa,dqb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb); dv = e.alloc(E)
ms = e.time_edges_dev(dqa.ptr,dqb.ptr,E,0.01,engine.SOA,dv.ptr,10)
print("ms", ms[2:].mean(), "undecided", e.last_undecided())
