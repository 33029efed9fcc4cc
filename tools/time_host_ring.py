"""A host-array batch split across S engines (one host thread each): does the PCIe-inclusive rate of the host-pointer
entry points improve when copies and kernels of the slices overlap?  (Measured: no -- 0.88 ms on one engine, 0.91-0.99 on
2-4: the copies from pageable memory are the bound.)"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, '.')
import bench
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
base = m.keyframe("home").qpos.copy()
E = 262144
qa, qb = bench.make_edges(m, qidx, E, 2)
for S in (1, 2, 3, 4):
    engs = [engine.Engine(m) for _ in range(S)]
    for e in engs: e.set_planning(qidx, base)
    bounds = np.linspace(0, E, S + 1).astype(int)
    outs = [None] * S
    def one(k):
        outs[k] = engs[k].check_edges(qa[bounds[k]:bounds[k+1]], qb[bounds[k]:bounds[k+1]], 0.01)
    def run():
        th = [threading.Thread(target=one, args=(k,)) for k in range(1, S)]
        for t in th: t.start()
        one(0)
        for t in th: t.join()
        return np.concatenate(outs)
    for _ in range(5): v = run()
    t0 = time.perf_counter()
    for _ in range(30): v = run()
    dt = (time.perf_counter() - t0) / 30
    print(S, round(dt * 1e3, 3), "ms", round(E / dt / 1e6), "M edges/s", int(v.sum()))
    for e in engs: e.close()
