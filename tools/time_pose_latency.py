"""What ONE evaluation of the projection's Newton step costs a wave that is alone on its SIMD: k_pose_apply over rows
that are all the SAME configuration (every lane takes the same number of iterations), 64 rows (one wave: pure latency)
up to 131 072 (the chip full).  Generated projection of the model's library against the interpreting kernel
(MJPL_POSE_SPEC=0 at creation).  -> gpurun_out/pose_latency.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mjpl_amd as mjpl
from mjpl_amd import scenes

m = scenes.franka_p(obstacles=False)
q_home = m.keyframe("home").qpos.copy()
out = {}
for tag, env in (("generated", "1"), ("interpreting", "0")):
    os.environ["MJPL_POSE_SPEC"] = env
    eng = mjpl.engine.Engine(m)
    frame = mjpl.site_pose(m, q_home, "ee_site", engine=eng)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), q_step=0.5, engine=eng)
    proj = pc._proj
    assert proj.spec_loaded() == (env == "1")
    rng = np.random.default_rng(3)
    q = q_home.copy()
    q[:7] += rng.normal(scale=0.08, size=7)
    for n in (64, 4096, 65536, 131072):
        Q = np.repeat(q[None], n, axis=0)
        Qo = np.repeat(q_home[None], n, axis=0)
        dq, dqo, dout = eng.alloc(Q.nbytes).upload(Q), eng.alloc(Qo.nbytes).upload(Qo), eng.alloc(Q.nbytes)
        dok, dit = eng.alloc(n), eng.alloc(4 * n)
        for _ in range(5):
            proj.apply_dev(dqo.ptr, dq.ptr, n, dout.ptr, dok.ptr, dit.ptr)
        eng.sync()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            proj.apply_dev(dqo.ptr, dq.ptr, n, dout.ptr, dok.ptr, dit.ptr)
        eng.sync()
        dt = (time.perf_counter() - t0) / reps
        it = dit.download(np.int32, n)
        assert (it == it[0]).all()
        evals = int(abs(it[0])) + 1  # Newton steps + the evaluation that finds the row within tolerance
        out[f"{tag} {n} rows"] = dict(us_per_launch=dt * 1e6, newton_steps=int(it[0]), chain_evaluations=evals,
                                      us_per_evaluation=dt * 1e6 / evals, ok=bool(dok.download(np.uint8, n)[0]))
        print(tag, n, out[f"{tag} {n} rows"], flush=True)
    eng.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/pose_latency.json", "w"), indent=1)
