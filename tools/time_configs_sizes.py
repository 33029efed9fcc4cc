"""configs[1] (Franka-P self-collision + floor) over batch sizes: the step and, through option kernel_timer, the filter
kernel alone -- how the 65 536-configuration batch (1 024 waves = one per SIMD) sits on the kernel's latency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mjpl_amd import engine, scenes
m = scenes.franka_p(obstacles=False)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
e = engine.Engine(m); e.set_planning(qidx, m.keyframe("home").qpos.copy())
rng = np.random.default_rng(1)
for N in (16384, 32768, 65536, 131072, 196608, 262144, 524288, 1048576):
    Q = rng.uniform(m.jnt_range[qidx, 0], m.jnt_range[qidx, 1], size=(N, len(qidx)))
    h = np.ascontiguousarray(Q.T)
    dq, dv = e.alloc(h.nbytes).upload(h), e.alloc(N)
    e.time_configs_dev(dq.ptr, N, engine.SOA, dv.ptr, 20)
    e.set_option("kernel_timer", 1)
    ms = e.time_configs_dev(dq.ptr, N, engine.SOA, dv.ptr, 50)
    k = e.get_option("kernel_timer_ms") / max(1.0, e.get_option("kernel_timer_launches"))
    e.set_option("kernel_timer", 0)
    print(N, "configs: step %.4f ms, filter kernel %.4f ms, %.3g configs/s, %.1f waves per SIMD" % (ms.mean(), k, N / ms.mean() * 1e3, N / 64 / 1024.0), flush=True)
