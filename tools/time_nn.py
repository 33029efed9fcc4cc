#!/usr/bin/env python3
"""Brute-force nearest neighbour (mjpl_nearest_dev) at planner sizes: M queries against n tree
nodes (SoA slab), checked against NumPy on a sample, timed with the engine's stream synchronised.
    python tools/time_nn.py            -> gpurun_out/nn.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjpl_amd import engine as eng_mod  # noqa: E402
from mjpl_amd import scenes  # noqa: E402


def main():
    m = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    e = eng_mod.Engine(m)
    e.set_planning(qidx, m.keyframe("home").qpos.copy())
    rng = np.random.default_rng(0)
    out = {}
    cells = int(os.environ.get("NN_CELLS", "1"))  # (0: the full scan of round 5 instead of the cell-ordered one)
    e.set_option("nn_cells", cells)
    for n, M in ((4096, 512), (65536, 4096), (1 << 20, 131072), (1 << 21, 131072)):
        nodes = rng.uniform(-2.5, 2.5, size=(7, n))
        qs = rng.uniform(-2.5, 2.5, size=(7, M))
        dn, dq, di = e.alloc(nodes.nbytes).upload(nodes), e.alloc(qs.nbytes).upload(qs), e.alloc(4 * M)
        e.nearest_dev(dn.ptr, n, n, dq.ptr, M, di.ptr)
        e.sync()
        reps = 3 if n >= 1 << 20 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            e.nearest_dev(dn.ptr, n, n, dq.ptr, M, di.ptr)
        e.sync()
        dt = (time.perf_counter() - t0) / reps
        got = di.download(np.int32, M)
        for j in range(0, M, max(1, M // 64)):  # sample check: sequential-sum squared norms, lowest index wins
            d = nodes - qs[:, j:j + 1]
            s = np.zeros(n)
            for c in range(7):
                s = s + d[c] * d[c]
            assert got[j] == int(np.argmin(s)), (n, M, j)
        out[f"{n}x{M}"] = {"ms": dt * 1e3, "pair_distances_per_s": n * M / dt, "cell_ordered": bool(e.get_option("nn_last_cells")),
                           "candidate_fraction": e.get_option("nn_last_candidate_fraction") if e.get_option("nn_last_cells") else None}
        print(n, M, out[f"{n}x{M}"], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/nn_cells%d.json" % cells, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
