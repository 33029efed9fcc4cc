"""How the CPU oracle scales with threads on this box (bench.py's cpu_baseline picks its thread
count from this kind of evidence): edges/s for 1, 2, 4 ... threads, plus the cgroup CPU quota."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mjpl_amd import scenes
from oracle import pyoracle

for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(p, open(p).read().strip())
    except OSError:
        pass
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), bench.host_cpu())
m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
base = m.keyframe("home").qpos.copy()
orc = pyoracle.Oracle(m, planning_qidx=qidx, qpos_base=base)
qa, qb = bench.make_edges(m, qidx, 65536, 2)
nt = 1
while nt <= (os.cpu_count() or 1):
    orc.valid_edges(qa, qb, 0.01, nthreads=nt)
    t0 = time.perf_counter()
    orc.valid_edges(qa, qb, 0.01, nthreads=nt)
    dt = time.perf_counter() - t0
    print(f"{nt:4d} threads: {65536 / dt:12.0f} edges/s")
    nt *= 2
