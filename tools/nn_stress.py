#!/usr/bin/env python3
"""Differential stress of the cell-ordered nearest-neighbour scan against the full scan (both exact by construction):
random sizes, planning sets of 2 ... 7 columns, uniform / clustered / duplicated / chain-like nodes, queries on nodes, near
nodes, far away, ranged look-ups behind an earlier answer -- indices and distances must be equal on EVERY query.
    python tools/nn_stress.py [cases] -> gpurun_out/nn_stress.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjpl_amd import engine as eng_mod  # noqa: E402
from mjpl_amd import scenes  # noqa: E402


def make_nodes(rng, kind, nplan, n):
    if kind == "uniform":
        return rng.uniform(-2.9, 2.9, size=(nplan, n))
    if kind == "clusters":  # a few hundred tight clusters (near-ties far below the screen's resolution) + background
        k = int(rng.integers(50, 400))
        c = rng.uniform(-2.5, 2.5, size=(nplan, k))
        x = c[:, rng.integers(0, k, n)] + rng.normal(scale=10.0 ** -rng.integers(2, 9), size=(nplan, n))
        x[:, : n // 4] = rng.uniform(-2.9, 2.9, size=(nplan, n // 4))
        return x
    if kind == "duplicates":  # every node several times
        base = rng.uniform(-2.9, 2.9, size=(nplan, max(1, n // 5)))
        return base[:, rng.integers(0, base.shape[1], n)]
    # chains: random walks of small steps from a few roots (what a tree looks like)
    roots = rng.uniform(-2, 2, size=(nplan, 64))
    x = np.empty((nplan, n))
    per = n // 64 + 1
    for r in range(64):
        steps = rng.normal(scale=0.02, size=(nplan, per))
        x[:, r * per:(r + 1) * per] = (roots[:, r:r + 1] + np.cumsum(steps, axis=1))[:, : max(0, min(per, n - r * per))]
    return np.clip(x, -3.0, 3.0)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    m = scenes.franka_p()
    rng = np.random.default_rng(2026)
    lines = []
    for case in range(cases):
        nplan = int(rng.integers(2, 8))
        n = int(rng.integers(262144, 800000))  # (whole look-ups take the screened paths from 16 x 16 384 nodes on)
        M = int(rng.integers(4096, 30000))
        kind = ("uniform", "clusters", "duplicates", "chains")[case % 4]
        e = eng_mod.Engine(m)
        e.set_planning(scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)[:nplan], m.keyframe("home").qpos)
        nodes = np.ascontiguousarray(make_nodes(rng, kind, nplan, n))
        pick = rng.integers(0, n, M)
        qs = nodes[:, pick].copy()
        third = M // 3
        qs[:, third: 2 * third] += rng.normal(scale=10.0 ** -rng.integers(1, 7), size=(nplan, third))
        qs[:, 2 * third:] = rng.uniform(-3.5, 3.5, size=(nplan, M - 2 * third))
        if case % 5 == 0:
            nodes[:, int(rng.integers(0, n))] = np.inf  # a sink
        dn, dq = e.alloc(nodes.nbytes).upload(nodes), e.alloc(qs.nbytes).upload(qs)
        res = {}
        n0 = int(rng.integers(1, n - 70000)) if case % 3 == 0 else 0
        for cells in (1, 0):
            e.set_option("nn_cells", cells)
            e.set_option("nn_cells_min_nodes", 16384)
            di, dd = e.alloc(4 * M), e.alloc(8 * M)
            if n0:
                di0, dd0 = e.alloc(4 * M), e.alloc(8 * M)
                e.nearest_dev(dn.ptr, n0, n, dq.ptr, M, di0.ptr, dd0.ptr)
                e.nearest_range_dev(dn.ptr, n0, n, n, dq.ptr, M, di.ptr, dd.ptr, di0.ptr, dd0.ptr)
            else:
                e.nearest_dev(dn.ptr, n, n, dq.ptr, M, di.ptr, dd.ptr)
            res[cells] = (di.download(np.int32, M), dd.download(np.float64, M), int(e.get_option("nn_last_cells")))
        same = np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1])
        # ... and a few queries against NumPy (the kernel's sum order; lowest index among equals)
        ok_np = True
        for j in rng.integers(0, M, 6):
            d = nodes - qs[:, j:j + 1]
            with np.errstate(invalid="ignore", over="ignore"):
                s = np.zeros(n)
                for c in range(nplan):
                    s = s + d[c] * d[c]
            s = np.where(np.isnan(s), np.inf, s)
            ok_np = ok_np and int(np.argmin(s)) == int(res[1][0][j]) and s.min() == res[1][1][j]
        line = f"case {case}: {kind} nplan {nplan} n {n} M {M} range_from {n0} cells_taken {res[1][2]} equal {same} numpy {ok_np}"
        print(line, flush=True)
        lines.append(line)
        assert same and ok_np and res[0][2] == 0 and (res[1][2] == 1 or (n0 and n - n0 < 16384)), line
        e.close()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/nn_stress.txt", "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
