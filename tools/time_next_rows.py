"""Timing of the "next" rows on the GPU box: batched PoseConstraint projection
(config 4 shape: 131 072 rows per GPU), IK seeds (config 5 shape: 16 384 seeds per GPU) and
nearest neighbour; the pose projection is also timed on the CPU oracle for a bounded sample.
Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, '.')
import numpy as np
import mjpl_amd as mjpl
from mjpl_amd import engine, scenes

out = {}
m = scenes.franka_p(obstacles=True)
joints = scenes.FRANKA_ARM_JOINTS
q_home = m.keyframe("home").qpos.copy()
cc = mjpl.CollisionConstraint(m)
eng = cc.engine
frame = mjpl.site_pose(m, q_home, "ee_site", engine=eng)
pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), q_step=0.5, engine=eng)
rng = np.random.default_rng(4)
n = 131072
lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
Q_old = np.clip(q_home + rng.normal(scale=0.01, size=(n, m.nq)), lo, hi)
d = rng.normal(size=(n, m.nq)); d[:, 7:] = 0
Q = np.clip(Q_old + 0.3 * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)  # needs real projection
dqo, dq = eng.alloc(Q_old.nbytes).upload(Q_old), eng.alloc(Q.nbytes).upload(Q)
dout, dok, dit = eng.alloc(Q.nbytes), eng.alloc(n), eng.alloc(4 * n)
for _ in range(3):
    pc._proj.apply_dev(dqo.ptr, dq.ptr, n, dout.ptr, dok.ptr, dit.ptr)
eng.sync()
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    pc._proj.apply_dev(dqo.ptr, dq.ptr, n, dout.ptr, dok.ptr, dit.ptr)
eng.sync()
dt = (time.perf_counter() - t0) / reps
iters = dit.download(np.int32, n)
ok = dok.download(np.uint8, n)
out["pose_apply"] = {"rows": n, "ms": dt * 1e3, "rows_per_s": n / dt, "accepted": float(ok.mean()),
                     "mean_projection_steps": float(np.abs(iters).mean())}
try:
    from oracle import pyoracle
    inv = frame.inverse()
    po = pyoracle.PoseOracle(m, "ee_site", (inv.wxyz_xyz[:4], inv.wxyz_xyz[4:]),
                             [(-np.inf, np.inf)] * 3 + [(-0.1, 0.1)] * 2 + [(-np.inf, np.inf)], q_step=0.5)
    k = 32768
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    po.apply_batch(Q_old[:k], Q[:k], nthreads=cores)
    dtc = time.perf_counter() - t0
    out["pose_apply"]["cpu_oracle_rows_per_s"] = k / dtc
    out["pose_apply"]["cpu_threads"] = cores
except Exception as ex:  # the oracle is test infrastructure; absent -> no CPU figure
    out["pose_apply"]["cpu_oracle"] = repr(ex)

solver = mjpl.HipIKSolver(m, joints, [], seed=3, num_seeds=16384, iterations=200, engine=eng)
q_t = mjpl.random_config(m, q_home, joints, 5, [mjpl.JointLimitConstraint(m), cc])
target = mjpl.site_pose(m, q_t, "ee_site", engine=eng)
Q0 = solver._seeds(q_home, np.random.default_rng(3))
eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, Q0, solver.movable, iterations=200)
t0 = time.perf_counter()
for _ in range(5):
    Qs, oks, its, err = eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, Q0, solver.movable,
                                     iterations=200)
dt = (time.perf_counter() - t0) / 5
out["ik_solve"] = {"seeds": len(Q0), "ms_including_pcie": dt * 1e3, "seeds_per_s": len(Q0) / dt,
                   "converged": float(oks.mean()), "mean_iterations": float(its.mean())}

print(json.dumps(out), file=sys.stderr, flush=True)
cc.set_planning(scenes.planning_index(m, joints), q_home)  # the NN kernel works on planning columns
cc._ensure_planning()
nodes = rng.uniform(lo[:7], hi[:7], size=(65536, 7)); queries = rng.uniform(lo[:7], hi[:7], size=(4096, 7))
hn, hq = np.ascontiguousarray(nodes.T), np.ascontiguousarray(queries.T)
dn, dqq, di = eng.alloc(hn.nbytes).upload(hn), eng.alloc(hq.nbytes).upload(hq), eng.alloc(4 * 4096)
eng.nearest_dev(dn.ptr, 65536, 65536, dqq.ptr, 4096, di.ptr); eng.sync()
t0 = time.perf_counter()
for _ in range(20):
    eng.nearest_dev(dn.ptr, 65536, 65536, dqq.ptr, 4096, di.ptr)
eng.sync()
dt = (time.perf_counter() - t0) / 20
out["nearest"] = {"nodes": 65536, "queries": 4096, "ms": dt * 1e3, "pair_distances_per_s": 65536 * 4096 / dt}
print(json.dumps(out))
