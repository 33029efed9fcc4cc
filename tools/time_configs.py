"""Configuration-check kernel alone (no edge walk) at several batch sizes, specialised or not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
e = engine.Engine(m); e.set_planning(qidx, m.keyframe("home").qpos.copy())
print("spec", e.spec_loaded())
for N in (65536, 262144, 1048576):
    qa, qb = bench.make_edges(m, qidx, N, 2)
    h = np.ascontiguousarray(qb.T)
    dq, dv = e.alloc(h.nbytes).upload(h), e.alloc(N)
    e.time_configs_dev(dq.ptr, N, engine.SOA, dv.ptr, 5)
    ms = e.time_configs_dev(dq.ptr, N, engine.SOA, dv.ptr, 50)
    print(N, "configs: %.4f ms (all kernels of the call), %.3g configs/s" % (ms.mean(), N / ms.mean() * 1e3), "undecided", e.last_undecided())
