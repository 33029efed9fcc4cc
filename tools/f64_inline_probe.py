#!/usr/bin/env python3
"""The wrong answers of the INLINED generated float64 check (profiles/README.md, round 5: "observed, not explained").
build: python3 tools/f64_inline_probe.py build   -> variants/f64x/<name>/libmjpl_spec_<hash>.so, one directory per variant
run  : python3 tools/f64_inline_probe.py         (GPU) every variant's float64-only verdicts on the headline batch against
       the interpreting kernel's, with where the differences are (endpoint / waypoint tiles)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from mjpl_amd import scenes, specialise  # noqa: E402

VARIANTS = {
    "outlined": ({"MJPL_SPEC_F64": "1"}, []),
    "inlined": ({"MJPL_SPEC_F64": "1", "MJPL_SPEC_F64_INLINE": "1"}, []),
    "inlined_O1": ({"MJPL_SPEC_F64": "1", "MJPL_SPEC_F64_INLINE": "1"}, ["-O1"]),
    "inlined_no_sgpr_to_vgpr": ({"MJPL_SPEC_F64": "1", "MJPL_SPEC_F64_INLINE": "1"}, ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"]),
    "inlined_no_machine_sched": ({"MJPL_SPEC_F64": "1", "MJPL_SPEC_F64_INLINE": "1"}, ["-mllvm", "-enable-misched=0"]),
    "inlined_no_licm": ({"MJPL_SPEC_F64": "1", "MJPL_SPEC_F64_INLINE": "1"}, ["-mllvm", "-disable-licm-promotion", "-mllvm", "-disable-machine-licm"]),
    "inlined_no_licm_promotion": ({"MJPL_SPEC_F64": "1", "MJPL_SPEC_F64_INLINE": "1"}, ["-mllvm", "-disable-licm-promotion"]),
    "inlined_no_machine_licm": ({"MJPL_SPEC_F64": "1", "MJPL_SPEC_F64_INLINE": "1"}, ["-mllvm", "-disable-machine-licm"]),
    "inlined_no_machine_sink": ({"MJPL_SPEC_F64": "1", "MJPL_SPEC_F64_INLINE": "1"}, ["-mllvm", "-disable-machine-sink"]),
}
BASE = os.path.join(ROOT, "variants", "f64x")


def model():
    m = scenes.franka_p(obstacles=True)
    return m, scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS), m.keyframe("home").qpos.copy()


def build():
    from concurrent.futures import ThreadPoolExecutor
    m, qidx, base = model()
    name = os.path.basename(specialise.spec_path(specialise.dump_program(m, (), qidx, base)[3].hash))
    jobs = []
    # (the generator's switches are environment variables: one variant's source at a time, the compiles side by side)
    with ThreadPoolExecutor(max_workers=6) as pool:
        for tag, (env, flags) in VARIANTS.items():
            if os.path.isdir(os.path.join(BASE, tag)) and os.listdir(os.path.join(BASE, tag)) and "--all" not in sys.argv:
                continue  # (built already)
            os.makedirs(os.path.join(BASE, tag), exist_ok=True)
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                src = specialise.build  # noqa: F841
                jobs.append(pool.submit(specialise.build, m, (), qidx, base, force=True, keep_source=False, extra_flags=flags,
                                        output=os.path.join(BASE, tag, name)))
                import time
                time.sleep(8)  # (the source is generated in the first seconds of build(); then the environment may change)
            finally:
                for k, v in old.items():
                    os.environ.pop(k, None)
                    if v is not None:
                        os.environ[k] = v
        for j in jobs:
            print(j.result())


def run():
    import bench
    from mjpl_amd import engine
    m, qidx, base = model()
    E = 262144
    qa, qb = bench.make_edges(m, qidx, E, seed=2)
    out = {}
    engine.set_spec_dir(None)
    ref_e = engine.Engine(m, options={"f64_spec": 0})
    ref_e.set_planning(qidx, base)
    ref_e.set_filter(False)
    ref = ref_e.check_edges(qa, qb, 0.01, first_bad=True)
    ref_e.close()
    for tag in VARIANTS:
        d = os.path.join(BASE, tag)
        if not os.path.isdir(d) or not os.listdir(d):
            out[tag] = "not built"
            continue
        engine.set_spec_dir(d)
        e = engine.Engine(m)
        e.set_planning(qidx, base)
        e.set_filter(False)
        got = e.check_edges(qa, qb, 0.01, first_bad=True)
        dv = got[0] != ref[0]
        # an edge the reference rejects at its endpoint (first_bad = waypoint count, by the ABI's convention) or at a waypoint
        out[tag] = {"spec_loaded": e.spec_kind(), "verdicts_differ": int(dv.sum()), "first_bad_differ": int((got[1] != ref[1]).sum()),
                    "said_free_where_reference_says_contact": int((got[0] & ~ref[0].astype(bool)).sum()),
                    "said_contact_where_reference_says_free": int((~got[0].astype(bool) & ref[0].astype(bool)).sum()),
                    "reference_invalid": int((~ref[0].astype(bool)).sum())}
        e.close()
    engine.set_spec_dir(None)
    print(json.dumps(out, indent=1))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/f64_inline_probe.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    build() if sys.argv[1:2] == ["build"] else run()
