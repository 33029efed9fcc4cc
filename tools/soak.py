"""Stability soak: many engine create/destroy cycles and long runs; prints device memory in use."""
import subprocess, sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
from mjpl_amd import engine, scenes
import mjpl_amd as mjpl

def used_mb():
    try:
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l and l[0].isalnum() and "card" in l.lower()][0]
        return int(line.split(",")[2]) / 1e6
    except Exception:
        return float("nan")

m = scenes.franka_p(obstacles=True); qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS); base = m.keyframe("home").qpos.copy()
qa, qb = bench.make_edges(m, qidx, 65536, 2)
print("start MB", used_mb())
ref = None
for cycle in range(40):
    e = engine.Engine(m); e.set_planning(qidx, base)
    v = e.check_edges(qa, qb, 0.01)
    c = e.check_configs(qb[:4096])
    if ref is None: ref = (v.copy(), c.copy())
    assert np.array_equal(v, ref[0]) and np.array_equal(c, ref[1])
    frame = mjpl.site_pose(m, base, "ee_site", engine=e)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), engine=e)
    pc.valid_config(base)
    del pc
    e.close()
    if cycle % 10 == 9: print("cycle", cycle, "MB", used_mb())
e = engine.Engine(m); e.set_planning(qidx, base)
ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
dqa, dqb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb); dv = e.alloc(len(qa))
t0 = time.time()
for rep in range(20):
    e.time_edges_dev(dqa.ptr, dqb.ptr, len(qa), 0.01, engine.SOA, dv.ptr, 500)
    assert np.array_equal(dv.download(np.uint8, len(qa)), ref[0])
print("10 000 steps ok in %.1f s, MB" % (time.time() - t0), used_mb())
