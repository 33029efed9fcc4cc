"""Latency of the host-pointer entry points at planner-sized batches (what the serial RRT pays per
extension): microseconds per mjpl_check_edges / mjpl_check_configs call."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
e = engine.Engine(m); e.set_planning(qidx, m.keyframe("home").qpos.copy())
out = {}
for E in (1, 16, 64, 256, 1024, 4096, 65536, 262144):
    qa, qb = bench.make_edges(m, qidx, E, 2)
    for _ in range(20):  # (both entry points warmed: the first call of either allocates)
        e.check_edges(qa, qb, 0.01)
        e.check_configs(qb)
    n = 300 if E <= 4096 else 30
    t0 = time.perf_counter()
    for _ in range(n):
        e.check_edges(qa, qb, 0.01)
    te = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        e.check_configs(qb)
    tc = (time.perf_counter() - t0) / n
    out[E] = dict(edges_us=te * 1e6, configs_us=tc * 1e6)
    print(E, out[E], flush=True)
json.dump(out, open("gpurun_out/host_latency.json", "w"), indent=1)
