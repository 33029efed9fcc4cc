"""Does lane coherence pay?  Times the bench step with the edges pre-sorted on the host by a Morton
key of the endpoint's first joints (data reordering only)."""
import sys; sys.path.insert(0,'.')
import numpy as np
import bench
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS); base = m.keyframe("home").qpos.copy()
e = engine.Engine(m); e.set_planning(qidx, base)
E=262144
qa,qb = bench.make_edges(m,qidx,E,2)
lo, hi = m.jnt_range[qidx,0], m.jnt_range[qidx,1]
def morton(q, joints, bits):
    b = np.clip(((q[:, joints] - lo[joints]) / (hi[joints] - lo[joints]) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = np.zeros(len(q), np.int64)
    for bit in range(bits - 1, -1, -1):
        for j in range(len(joints)):
            key = (key << 1) | ((b[:, j] >> bit) & 1)
    return key
for name, order in (("unsorted", np.arange(E)),
                    ("morton q1-q4 x4bits", np.argsort(morton(qb, [0,1,2,3], 4), kind="stable")),
                    ("morton q1-q6 x3bits", np.argsort(morton(qb, [0,1,2,3,4,5], 3), kind="stable")),
                    ("morton q1-q3 x5bits", np.argsort(morton(qb, [0,1,2], 5), kind="stable")),
                    ("lexsort q1..q7", np.lexsort(qb.T[::-1]))):
    a, b = qa[order], qb[order]
    ha,hb = np.ascontiguousarray(a.T), np.ascontiguousarray(b.T)
    dqa,dqb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb); dv = e.alloc(E)
    e.time_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 3)  # warm-up (buffers are allocated on first use)
    ms, msk = e.time_edges_dev(dqa.ptr,dqb.ptr,E,0.01,engine.SOA,dv.ptr,12, first_kernel=True)
    print(f"{name:22s} step {ms[2:].mean():.4f} ms  items kernel {msk[2:].mean():.4f} ms  valid {dv.download(np.uint8,E).mean():.4f}")
    for x in (dqa,dqb,dv): x.free()
