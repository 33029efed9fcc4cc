#!/usr/bin/env python3
"""A/B of the fused filter kernel (mjpl_fused.h) against the two persistent kernels on bench.py's batch:
verdicts and first-bad indices must be equal (and equal to the oracle's on a sample), then step and stage times.
    python tools/time_fused.py [--edges n,n] [--iters K] [--configs name=ENV:VAL,ENV:VAL ...] [--out file.json]
Every configuration is an engine created under the given environment (the MJPL_* switches are read at creation)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from mjpl_amd import engine, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--edges", default="262144")
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--configs", nargs="*", default=["two_kernels=MJPL_FUSED:0", "fused12=MJPL_FUSED:1",
                                                  "fused12_items_first=MJPL_FUSED:1,MJPL_FUSED_POLICY:1",
                                                  "fused6=MJPL_FUSED:1,MJPL_FUSED_WAVES:6"])
ap.add_argument("--oracle", type=int, default=4096, help="edges of the batch also checked against the CPU oracle")
ap.add_argument("--spec", type=int, default=1)
ap.add_argument("--out", default="")
args = ap.parse_args()

m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
base = m.keyframe("home").qpos.copy()
res = []
for E in [int(x) for x in args.edges.split(",")]:
    qa, qb = bench.make_edges(m, qidx, E, 2)
    ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
    ref = None
    for cfg in args.configs:
        name, _, envs = cfg.partition("=")
        saved = {}
        for kv in [x for x in envs.split(",") if x]:
            k, _, v = kv.partition(":")
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        e = engine.Engine(m)
        if not args.spec:
            e.set_spec(0)
        e.set_planning(qidx, base)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        info = e.info()
        dqa, dqb, dv, dfb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb), e.alloc(E), e.alloc(4 * E)
        e.check_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, dfb.ptr)
        e.sync()
        status = e.take_status()
        v, fb = dv.download(np.uint8, E), dfb.download(np.int32, E)
        items, interior, und = e.last_items(), e.last_interior_edges(), e.last_undecided()
        if ref is None:
            ref = (v, fb)
            if args.oracle:
                from oracle import pyoracle
                orc = pyoracle.Oracle(m, planning_qidx=qidx, qpos_base=base)
                n = min(E, args.oracle)
                want = orc.valid_edges(qa[:n], qb[:n], 0.01, nthreads=8)
                assert np.array_equal(v[:n].astype(bool), np.asarray(want).astype(bool)), "first configuration differs from the oracle"
        same = bool(np.array_equal(v, ref[0]) and np.array_equal(fb, ref[1]))
        e.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 20, 1 << 30)
        mean, st, ns = e.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, args.iters, 4)
        # the same without any event inside the launches
        mean0, _, _ = e.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, args.iters, 1 << 30)
        row = dict(config=name, edges=E, step_ms=mean, step_ms_no_marks=mean0, edges_per_s=E / mean0 * 1e3, stages_ms=st,
                   same_as_first=same, status=status, items=items, interior_edges=interior, undecided=und,
                   valid=int(v.sum()), fused=info.get("fused_edges"), waves=info.get("fused_waves"), spec=e.spec_kind())
        res.append(row)
        print(json.dumps(row), flush=True)
        if not same:
            bad = np.flatnonzero((v != ref[0]) | (fb != ref[1]))
            print(f"  !! {len(bad)} edges differ, first: {bad[:10]}, verdicts {v[bad[:10]]} vs {ref[0][bad[:10]]}, "
                  f"first_bad {fb[bad[:10]]} vs {ref[1][bad[:10]]}", flush=True)
        e.close()
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
