"""The fused kernel's edge certificate on the headline batch: edges/s, certified edges, waypoint items and verdict
equality with the certificate on and off (MJPL_FUSED_CERT) under both tile policies.  Prints one JSON object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from mjpl_amd import engine, scenes

m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
base = m.keyframe("home").qpos.copy()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
qa, qb = bench.make_edges(m, qidx, E, seed=2)
ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
out, ref = {}, None
for cert in ("1", "0"):
    for policy in ("0", "1"):
        os.environ["MJPL_FUSED_CERT"], os.environ["MJPL_FUSED_POLICY"] = cert, policy
        e = engine.Engine(m)
        e.set_planning(qidx, base)
        da, db, dv, dfb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb), e.alloc(E), e.alloc(4 * E)
        e.time_edges_stages_dev(da.ptr, db.ptr, E, 0.01, engine.SOA, dv.ptr, 300, 1 << 30)
        e.sync()
        t0 = time.perf_counter()
        e.time_edges_stages_dev(da.ptr, db.ptr, E, 0.01, engine.SOA, dv.ptr, 1000, 1 << 30)
        e.sync()
        dt = (time.perf_counter() - t0) / 1000
        v = dv.download(np.uint8, E)
        ref = v if ref is None else ref
        out[f"cert{cert}_policy{policy}"] = dict(ms=dt * 1e3, edges_per_s=E / dt, valid=float(v.mean()), certified=e.last_certified(),
                                                 interior=e.last_interior_edges(), items=e.last_items(), verdicts_equal=bool(np.array_equal(v, ref)))
        e.close()
print(json.dumps(out, indent=1))
