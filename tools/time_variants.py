"""Time the bench step with each timing-only library in variants/ (results are NOT verdicts)."""
import glob, os, sys
sys.path.insert(0, '.')
import numpy as np
import bench
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS); base = m.keyframe("home").qpos.copy()
E = 262144
qa, qb = bench.make_edges(m, qidx, E, 2)
ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
libs = [None] + sorted(p for p in glob.glob("variants/lib_*.so") if "stamps" not in p)
if len(sys.argv) > 1:
    libs = [None] + [f"variants/lib_{n}.so" for n in sys.argv[1:]]
for lib in libs:
    e = engine.Engine(m, lib_path=os.path.abspath(lib) if lib else None); e.set_planning(qidx, base)
    dqa, dqb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb); dv = e.alloc(E)
    e.time_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 3)  # warm-up (buffers are allocated on first use)
    ms, msk = e.time_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 12, first_kernel=True)
    print(f"{lib or 'product':32s} step {ms[2:].mean():.4f} ms  first kernel {msk[2:].mean():.4f} ms  undecided {e.last_undecided()}")
