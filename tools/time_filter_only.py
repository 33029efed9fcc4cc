"""Timing (GPU box): the float32 filter kernels of the bench step alone (no exact re-check)."""
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
e.time_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 3)  # warm-up (buffers are allocated on first use)
ms, msk = e.time_edges_dev(dqa.ptr,dqb.ptr,E,0.01,engine.SOA,dv.ptr,12, first_kernel=True)
print("step ms", ms[2:].mean(), "main kernel ms", msk[2:].mean(), "undecided", e.last_undecided(), "interior edges", e.last_interior_edges(), "items", e.last_items())
