"""Diagnostic (GPU box): undecided-item counts and verdict agreement of the float32 filter on the bench edges."""
import sys; sys.path.insert(0,'.')
import numpy as np
import bench
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS); base = m.keyframe("home").qpos.copy()
e = engine.Engine(m); e.set_planning(qidx, base)
E=262144
qa,qb = bench.make_edges(m,qidx,E,2)
ha,hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
dqa,dqb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb); dv = e.alloc(E)
for tol in (1e-4, 1e-5, 1e-3):
    e.set_filter(True, tol)
    ms = e.time_edges_dev(dqa.ptr,dqb.ptr,E,0.01,engine.SOA,dv.ptr,10)
    print("tol",tol,"ms",ms[2:].mean(),"undecided",e.last_undecided(), "of", E)
e.set_filter(False)
ms = e.time_edges_dev(dqa.ptr,dqb.ptr,E,0.01,engine.SOA,dv.ptr,10); print("exact ms", ms[2:].mean())
