"""Per-kernel stage times of the edge pipeline (mjpl_time_edges_stages_dev) for one or more builds
of the library and batch sizes.  usage: python tools/time_stages.py [--libs a.so,b.so] [--edges n,n]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from mjpl_amd import engine, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--libs", default="")
ap.add_argument("--edges", default="262144")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--out", default="")
args = ap.parse_args()
m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
base = m.keyframe("home").qpos.copy()
res = []
for lib in [None] + [x for x in args.libs.split(",") if x]:
    for E in [int(x) for x in args.edges.split(",")]:
        qa, qb = bench.make_edges(m, qidx, E, 2)
        ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
        e = engine.Engine(m, lib_path=os.path.abspath(lib) if lib else None)
        e.set_planning(qidx, base)
        dqa, dqb, dv = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb), e.alloc(E)
        e.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 20, 1 << 30)
        mean, st, ns = e.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, args.iters, 4)
        row = dict(lib=os.path.basename(lib) if lib else "product", edges=E, step_ms=mean, edges_per_s=E / mean * 1e3,
                   stages_ms=st, lds_filter=e.info()["lds_bytes_filter"])
        res.append(row)
        print(json.dumps(row), flush=True)
        e.close()
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
