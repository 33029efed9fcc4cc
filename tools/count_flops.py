#!/usr/bin/env python3
"""Exact operation count of the float64 arithmetic the CPU oracle executes for the headline workload
("estimate to be replaced by an exact static count emitted by cpu_ref"): runs the counting
build of the oracle (oracle/libmjpl_oracle_count.so, -DORC_COUNT_FLOPS: every routine tallies the additions,
multiplications, divisions and square roots of the statements it executed, sin/cos pairs as calls) over a
seeded sample of bench.py's edges, single-threaded, and writes profiles/flops.json.

    python tools/count_flops.py [--edges 16384]

flops = add + mul + div + sqrt + 55 per sin/cos pair (what the fdlibm restatement of oracle/orc_math.h
executes: 27 additions, 28 multiplications).  The count is of the REFERENCE's algorithm as the oracle restates
it: kinematics of every body and geom, dynamic pair enumeration with a bounding test per filtered pair, the
waypoint recurrence, short-circuit after the first invalid check.  The GPU's float64 kernels fold the static
bodies at create and so execute fewer operations per verdict; bench.py uses this count for
achieved_FP64_fraction as SURVEY.md 8d defines it (edges/s x flops_per_edge / peak)."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", type=int, default=16384)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "flops.json"))
    args = ap.parse_args()
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libmjpl_oracle_count.so"], check=True, stdout=subprocess.DEVNULL)
    os.environ["MJPL_ORACLE_LIB"] = os.path.join(ROOT, "oracle", "libmjpl_oracle_count.so")
    import numpy as np

    import bench
    from mjpl_amd import scenes
    from oracle import pyoracle
    model = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(model, scenes.FRANKA_ARM_JOINTS)
    base = model.keyframe("home").qpos.copy()
    qa, qb = bench.make_edges(model, qidx, bench.EDGES_PER_GPU, seed=2)
    n = args.edges
    orc = pyoracle.Oracle(model, planning_qidx=qidx, qpos_base=base)
    lib = pyoracle.lib()
    out = (C.c_longlong * 5)()

    def tally(fn):
        lib.orc_flops_reset()
        r = fn()
        lib.orc_flops_get(out)
        return r, dict(zip(("add", "mul", "div", "sqrt", "sincos"), [int(x) for x in out]))

    valid, edge_ops = tally(lambda: orc.valid_edges(qa[:n], qb[:n], bench.STEP, nthreads=1))
    _, cfg_ops = tally(lambda: orc.valid_configs(qb[:n], nthreads=1))

    def flops(o):
        return o["add"] + o["mul"] + o["div"] + o["sqrt"] + 55 * o["sincos"]

    rec = {"source": "tools/count_flops.py: oracle/libmjpl_oracle_count.so over the first %d edges of bench.py's rank-0 batch "
                     "(seed 2), single thread" % n,
           "edges_sampled": n, "valid_fraction_of_sample": float(np.mean(valid)),
           "flops_per_edge": flops(edge_ops) / n, "ops_per_edge": {k: v / n for k, v in edge_ops.items()},
           "flops_per_config": flops(cfg_ops) / n, "ops_per_config": {k: v / n for k, v in cfg_ops.items()},
           "sincos_weight": 55,
           "note": "operations of the reference's algorithm as the oracle executes it (all bodies and geoms, dynamic pair "
                   "enumeration, short-circuit after the first invalid check); comparisons, negations, fabs, fmin / fmax and "
                   "copies are not counted"}
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
