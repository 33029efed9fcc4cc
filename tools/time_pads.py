"""Edges/s of Franka-P with the reference Panda's ten finger-pad boxes (moving boxes), against plain Franka-P: the
model's generated library, the interpreter (MJPL_SPEC=0), filter off.  Verdicts of every variant are compared with the
interpreter's on all edges.  `time_pads.py <label>`: the pads model only, results under "<label>" (a second build of
the library in MJPL_SPEC_DIR: a process loads one library per program hash)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mjpl_amd import engine, scenes

KEYS = ("MJPL_SPEC", "MJPL_FUSED", "MJPL_FILTER", "MJPL_FUSED_MBOX")


def make(m, qidx, base, **env):
    old = {k: os.environ.pop(k, None) for k in KEYS}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        e = engine.Engine(m)
        e.set_planning(qidx, base)
    finally:
        for k in KEYS:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]
    return e


out = {}
label = sys.argv[1] if len(sys.argv) > 1 else ""
models = (("franka_p+16obs", scenes.franka_p(True)), ("franka_p+16obs+10pads", scenes.franka_p(True, True)))
for tag, m in (models[1:] if label else models):
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    E = 262144
    qa, qb = bench.make_edges(m, qidx, E, 2)
    ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
    want = None
    variants = [("interpreter", {"MJPL_SPEC": 0}), ("interpreter fused", {"MJPL_SPEC": 0, "MJPL_FUSED_MBOX": 1}), ("library", {}),
                ("library two kernels", {"MJPL_FUSED": 0}), ("filter off", {"MJPL_FILTER": 0})]
    if label:
        variants = [variants[0], (f"library {label}", {}), (f"library {label} two kernels", {"MJPL_FUSED": 0})]
    for name, env in variants:
        e = make(m, qidx, base, **env)
        dqa, dqb, dv, dfb = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb), e.alloc(E), e.alloc(4 * E)
        e.check_edges_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, dfb.ptr)
        v, fb = dv.download(np.uint8, E), dfb.download(np.int32, E)
        if want is None:
            want = (v, fb)
        same = bool((v == want[0]).all() and (fb == want[1]).all())
        e.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 10, 1 << 30)
        mean, st, _ = e.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 60, 4)
        info = e.info()
        out[f"{tag} {name}"] = dict(step_ms=mean, edges_per_s=E / mean * 1e3, stages_ms={k: s for k, s in st.items() if s > 0}, npairs=info["npairs"],
                                    nslots=info["nslots"], spec_loaded=e.spec_kind(), fused_edges=info["fused_edges"],
                                    fused_waves=info["fused_waves"], undecided=e.last_undecided(), valid=float(v.mean()),
                                    same_as_interpreter=same)
        print(tag, name, out[f"{tag} {name}"], flush=True)
        e.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/r04_pads{'_' + label if label else ''}.json", "w"), indent=1)
