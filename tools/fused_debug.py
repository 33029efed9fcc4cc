#!/usr/bin/env python3
"""Per-wave accounting of one launch of the fused filter kernel (a -DMJPL_FUSED_DEBUG build of the bench model's
library must be in place: tools/build_bench_spec.py FUSED_DEBUG:-DMJPL_FUSED_DEBUG, copied over the real library).
    python tools/fused_debug.py [--edges N] [--env K:V,K:V]"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=262144)
ap.add_argument("--env", default="")
args = ap.parse_args()
path = os.path.join(tempfile.gettempdir(), f"fused_dbg_{os.getpid()}.bin")
os.environ["MJPL_FUSED_DEBUG"] = path
for kv in [x for x in args.env.split(",") if x]:
    k, _, v = kv.partition(":")
    os.environ[k] = v
import bench
from mjpl_amd import engine, scenes

m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
base = m.keyframe("home").qpos.copy()
E = args.edges
qa, qb = bench.make_edges(m, qidx, E, 2)
ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
e = engine.Engine(m)
e.set_planning(qidx, base)
info = e.info()
dqa, dqb, dv = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb), e.alloc(E)
for _ in range(3):
    e.check_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr)
e.sync()
mean, st, _ = e.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 20, 1)
e.close()
d = np.fromfile(path, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
os.remove(path)
full = d
d = d[d[:, 7] > 0]
nw = info["fused_waves"] or 12
if len(full) >= nw and len(d) % nw == 0:
    g = d.reshape(-1, nw, 8)
    end = g[:, :, 7].max(axis=1) / 100.0
    busy = (g[:, :, 3] + g[:, :, 4] + g[:, :, 6]).sum(axis=1) / 100.0 / nw
    items = g[:, :, 1].sum(axis=1)
    print(f"workgroups {len(g)}: last wave ends at us min {end.min():.1f} p5 {np.percentile(end,5):.1f} p50 {np.median(end):.1f} p95 {np.percentile(end,95):.1f} "
          f"max {end.max():.1f}; busy time per wave of a workgroup min {busy.min():.1f} mean {busy.mean():.1f} max {busy.max():.1f}; "
          f"item tiles per workgroup min {items.min():.0f} mean {items.mean():.1f} max {items.max():.0f}")
    print(f"  correlation(end, busy) {np.corrcoef(end, busy)[0,1]:.2f}; idle inside workgroups (end - mean wave lifetime) mean "
          f"{(end - g[:, :, 7].mean(axis=1) / 100.0).mean():.1f} us")
us = d[:, 3:] / 100.0  # wall_clock64: 100 MHz
print(f"env {args.env or '-'}: fused {info['fused_edges']} waves/wg {info['fused_waves']}  step {mean*1e3:.1f} us, stages {st}")
print(f"waves {len(d)}: endpoint tiles {d[:,0].sum():.0f} (per wave min/mean/max {d[:,0].min():.0f}/{d[:,0].mean():.2f}/{d[:,0].max():.0f}), "
      f"item tiles {d[:,1].sum():.0f} ({d[:,1].min():.0f}/{d[:,1].mean():.2f}/{d[:,1].max():.0f}), wait polls {d[:,2].sum():.0f}")
names = ["endpoint tiles", "item tiles", "waiting", "decide+lock", "wave lifetime"]
for k, n in enumerate(names):
    c = us[:, k]
    print(f"  {n:15s} per wave us: min {c.min():8.1f} mean {c.mean():8.1f} p50 {np.median(c):8.1f} p95 {np.percentile(c,95):8.1f} max {c.max():8.1f}")
ep, it = d[:, 0].sum(), d[:, 1].sum()
print(f"  per endpoint tile {us[:,0].sum()/max(ep,1):.1f} us, per item tile {us[:,1].sum()/max(it,1):.1f} us")
life = us[:, 4]
print(f"  lifetime histogram (us): " + " ".join(f"{int(x)}" for x in np.percentile(life, [0, 5, 25, 50, 75, 95, 100])))
