"""The projection of 131 072 rows in one launch (MJPL_POSE_PHASE_STEPS=0) against two (every row's first k Newton steps
with one lane per row, the rest with eight): time per batch, results compared byte for byte.  Prints one JSON object."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, time
sys.path.insert(0, ".")
import numpy as np
import mjpl_amd as mjpl
from mjpl_amd import scenes
m = scenes.franka_p(obstacles=True)
q_home = m.keyframe("home").qpos.copy()
eng = mjpl.engine.Engine(m)
frame = mjpl.site_pose(m, q_home, "ee_site", engine=eng)
pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), q_step=0.5, engine=eng)
rng = np.random.default_rng(4)
n = 131072
lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
Q_old = np.clip(q_home + rng.normal(scale=0.01, size=(n, m.nq)), lo, hi)
d = rng.normal(size=(n, m.nq)); d[:, 7:] = 0
Q = np.clip(Q_old + 0.3 * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)
dqo, dq = eng.alloc(Q_old.nbytes).upload(Q_old), eng.alloc(Q.nbytes).upload(Q)
dout, dok, dit = eng.alloc(Q.nbytes), eng.alloc(n), eng.alloc(4 * n)
for _ in range(20):
    pc._proj.apply_dev(dqo.ptr, dq.ptr, n, dout.ptr, dok.ptr, dit.ptr)
eng.sync(); t0 = time.perf_counter()
for _ in range(500):
    pc._proj.apply_dev(dqo.ptr, dq.ptr, n, dout.ptr, dok.ptr, dit.ptr)
eng.sync(); dt = (time.perf_counter() - t0) / 500
out = dout.download(np.float64, n * m.nq); ok = dok.download(np.uint8, n); it = dit.download(np.int32, n)
import hashlib
h = hashlib.sha256(out.tobytes() + ok.tobytes() + it.tobytes()).hexdigest()
print(json.dumps(dict(ms=dt * 1e3, rows_per_s=n / dt, accepted=float(ok.mean()), mean_steps=float(np.abs(it).mean()), max_steps=int(np.abs(it).max()), sha256=h)))
'''
res = {}
for steps in ("0", "2", "3", "4", "6"):
    env = dict(os.environ, MJPL_POSE_PHASE_STEPS=steps)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    res[f"phase_steps_{steps}"] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
ref = res["phase_steps_0"].get("sha256")
for v in res.values():
    v["same_bytes_as_one_launch"] = v.get("sha256") == ref
    v.pop("sha256", None)
print(json.dumps(res, indent=1))
