"""The float64-only edge path (filter off) on the headline batch: a library built with MJPL_SPEC_F64=1 (the generated
check, ExactFull: `tools/build_bench_spec.py F64:env=MJPL_SPEC_F64=1`, copied over the real library) against the
interpreting kernel (MJPL_F64_SPEC=0), verdicts and first-bad indices compared on all edges.  Prints one JSON object."""
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
for tag, env in (("generated", {}), ("interpreting", {"MJPL_F64_SPEC": "0"})):
    for k in ("MJPL_F64_SPEC",):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = engine.Engine(m)
    e.set_planning(qidx, base)
    e.set_filter(False)
    da, db, dv, dfb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb), e.alloc(E), e.alloc(4 * E)
    e.time_edges_stages_dev(da.ptr, db.ptr, E, 0.01, engine.SOA, dv.ptr, 30, 1 << 30)
    e.sync()
    t0 = time.perf_counter()
    e.time_edges_stages_dev(da.ptr, db.ptr, E, 0.01, engine.SOA, dv.ptr, 200, 1 << 30)
    e.sync()
    dt = (time.perf_counter() - t0) / 200
    got = e.check_edges(qa, qb, 0.01, first_bad=True)
    ref = got if ref is None else ref
    out[tag] = dict(ms=dt * 1e3, edges_per_s=E / dt, valid=float(got[0].mean()),
                    equal_to_first=bool(np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])))
    e.close()
print(json.dumps(out, indent=1))
