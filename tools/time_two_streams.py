"""Throughput of back-to-back edge batches with ONE engine (one stream) against TWO engines taking turns (two
streams: the tail of one batch's kernels overlaps the head of the next one's).  Same batch, same verdicts."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mjpl_amd import engine, scenes  # noqa: E402

m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
base = m.keyframe("home").qpos.copy()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
K = 1000
qa, qb = bench.make_edges(m, qidx, E, 2)
ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
for nen in (1, 2, 3, 1, 2):
    engs = []
    for _ in range(nen):
        e = engine.Engine(m)
        e.set_planning(qidx, base)
        engs.append((e, e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb), e.alloc(E)))
    for e, a, b, v in engs:
        for _ in range(20):
            e.check_edges_dev(a.ptr, b.ptr, E, 0.01, engine.SOA, v.ptr)
        e.sync()
    t0 = time.perf_counter()
    for i in range(K):
        e, a, b, v = engs[i % nen]
        e.check_edges_dev(a.ptr, b.ptr, E, 0.01, engine.SOA, v.ptr)
    for e, *_ in engs:
        e.sync()
    dt = time.perf_counter() - t0
    ref = engs[0][3].download(np.uint8, E)
    assert all(np.array_equal(x[3].download(np.uint8, E), ref) for x in engs)
    print(f"{nen} engine(s): {dt / K * 1e3:.4f} ms per batch, {E * K / dt / 1e6:.0f} M edges/s", flush=True)
    for e, *_ in engs:
        e.close()
