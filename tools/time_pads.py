"""Edges/s of Franka-P with the reference Panda's ten finger-pad boxes (moving boxes: immediate
interpreter), filter on / off, against plain Franka-P."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mjpl_amd import engine, scenes
out = {}
for tag, m in (("franka_p+16obs", scenes.franka_p(True)), ("franka_p+16obs+10pads", scenes.franka_p(True, True))):
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    E = 262144
    qa, qb = bench.make_edges(m, qidx, E, 2)
    ha, hb = np.ascontiguousarray(qa.T), np.ascontiguousarray(qb.T)
    for filt in (True, False):
        e = engine.Engine(m); e.set_planning(qidx, base)
        if not filt:
            e.set_filter(False)
        dqa, dqb, dv = e.alloc(ha.nbytes).upload(ha), e.alloc(hb.nbytes).upload(hb), e.alloc(E)
        e.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 10, 1 << 30)
        mean, st, _ = e.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, 0.01, engine.SOA, dv.ptr, 60, 4)
        info = e.info()
        out[f"{tag} filter={'on' if filt else 'off'}"] = dict(step_ms=mean, edges_per_s=E / mean * 1e3, stages_ms=st, npairs=info["npairs"],
                                                          nslots=info["nslots"], undecided=e.last_undecided(), valid=float(dv.download(np.uint8, E).mean()))
        print(tag, filt, out[f"{tag} filter={'on' if filt else 'off'}"], flush=True)
        e.close()
json.dump(out, open("gpurun_out/r04_pads.json", "w"), indent=1)
