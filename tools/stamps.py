"""Phase accounting of the queued filter kernel (diagnostic build variants/lib_stamps.so):
s_memtime cycles per wave summed over the launch, as shares."""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
import bench
from mjpl_amd import engine, scenes
lib_path = os.path.abspath("variants/lib_stamps.so")
m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS); base = m.keyframe("home").qpos.copy()
e = engine.Engine(m, lib_path=lib_path); e.set_planning(qidx, base)
E = 262144
qa, qb = bench.make_edges(m, qidx, E, 2)
ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
dqa, dqb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb); dv = e.alloc(E)
out = (C.c_ulonglong * 16)()
e.time_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 2)
e.lib.mjpl_debug_stamps(out)
ms = e.time_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 5)
e.lib.mjpl_debug_stamps(out)
v = np.array(list(out), dtype=np.float64)
names = ["FK body", "geom record+pose", "culls+pushes", "drains", "slot store", "tail", "-", "config-waves"]
tot = v[:6].sum()
print("launch ms", ms.mean(), "config-waves", v[7])
for n, x in zip(names[:6], v[:6]):
    print(f"{n:18s} {x / tot * 100:5.1f} %   {x / v[7]:9.0f} cycles per config-wave")
for tag, o in (("general queue", 8), ("box queue", 12)):
    if v[o] > 0:
        print(f"{tag:14s} drains/config-wave {v[o] / v[7]:.2f}  lanes/drain {v[o + 1] / v[o]:.1f}  "
              f"pop+gather {v[o + 2] / v[o]:.0f} cycles  narrowphase {v[o + 3] / v[o]:.0f} cycles per drain")
