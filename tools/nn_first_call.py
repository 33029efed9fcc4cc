"""First and later calls of mjpl_nearest_dev on a growing tree in a fresh process (lazy kernel loading, buffers)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from mjpl_amd import engine as eng_mod, scenes
m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
e = eng_mod.Engine(m); e.set_planning(qidx, m.keyframe("home").qpos.copy())
rng = np.random.default_rng(0)
n, M, cap = 520000, 131072, 1 << 20
nodes = rng.uniform(-2.5, 2.5, size=(7, n)); qs = rng.uniform(-2.5, 2.5, size=(7, M))
nodes = rng.uniform(-2.5, 2.5, size=(7, cap))
big = e.alloc(nodes.nbytes).upload(nodes)
dq, di = e.alloc(qs.nbytes).upload(qs), e.alloc(4 * M)
e.sync()
for k in range(4):
    t0 = time.perf_counter(); e.nearest_dev(big.ptr, n + k * 100000, cap, dq.ptr, M, di.ptr); e.sync(); print(k, (time.perf_counter() - t0) * 1e3, flush=True)
