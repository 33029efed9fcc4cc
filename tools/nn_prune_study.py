"""How much of a tree a cell-ordered nearest-neighbour scan has to read (CPU, NumPy + scipy), on trees dumped by
tools/dump_trees.py: nodes and queries sorted by a Morton key over their cells, bounding boxes per sub-chunk of nodes and
per wave of 128 queries, a sub-chunk scanned iff the box-to-box distance is within the wave's largest bound.
    python3 tools/nn_prune_study.py gpurun_out/trees_r5.npz [sub] [keybits]"""
import sys
import numpy as np
from scipy.spatial import cKDTree


def bit_plan(lo, hi, bits):
    """greedy: every bit halves the column whose cells are widest; -> list of columns, most significant first"""
    w = (hi - lo).astype(float).copy()
    plan = []
    for _ in range(bits):
        c = int(np.argmax(w))
        plan.append(c)
        w[c] *= 0.5
    return plan


def keys(X, lo, hi, plan):
    nb = np.bincount(plan, minlength=X.shape[1])
    u = [np.clip(((X[:, c] - lo[c]) / (hi[c] - lo[c]) * (1 << nb[c])).astype(np.int64), 0, (1 << nb[c]) - 1) if nb[c] else None
         for c in range(X.shape[1])]
    left = nb.copy()
    k = np.zeros(len(X), np.uint64)
    for c in plan:
        left[c] -= 1
        k = (k << np.uint64(1)) | ((u[c] >> left[c]) & 1).astype(np.uint64)
    return k


def study(nodes, queries, sub, bits, label):
    lo, hi = nodes.min(0), nodes.max(0)
    plan = bit_plan(lo, hi, bits)
    kn = keys(nodes, lo, hi, plan)
    on = np.argsort(kn, kind="stable")
    ns = nodes[on]
    kq = keys(queries, lo, hi, plan)
    oq = np.argsort(kq, kind="stable")
    qs = queries[oq]
    nsub = (len(ns) + sub - 1) // sub
    pad = nsub * sub - len(ns)
    nsp = np.concatenate([ns, np.repeat(ns[-1:], pad, 0)]).reshape(nsub, sub, -1)
    nlo, nhi = nsp.min(1), nsp.max(1)
    W = 128
    nw = len(qs) // W
    qw = qs[: nw * W].reshape(nw, W, -1)
    qlo, qhi = qw.min(1), qw.max(1)
    # bounds: exact distance to the nearest node of a strided sample of 65 536 (what the sample pass gives), and the true one
    samp = ns[:: max(1, len(ns) // 65536)]
    ds, _ = cKDTree(samp).query(qs[: nw * W], workers=8)
    sel = np.random.default_rng(0).choice(nw, size=min(nw, 128), replace=False)
    dt, _ = cKDTree(ns).query(qw[sel].reshape(-1, qs.shape[1]), workers=8)
    dt = dt.reshape(len(sel), W)
    ds = ds.reshape(nw, W)
    out = {}
    for name, r in (("sample bound", ds[sel].max(1)), ("true distance", dt.max(1))):
        frac = []
        for i, w in enumerate(sel):
            gap = np.maximum(np.maximum(qlo[w][None] - nhi, nlo - qhi[w][None]), 0.0)
            lb2 = (gap * gap).sum(1)
            frac.append(float((lb2 <= r[i] ** 2).mean()))
        out[name] = float(np.mean(frac))
    print(f"{label}: {len(ns)} nodes, {len(qs)} queries, sub {sub}, {bits} key bits, plan {plan[:14]}...; "
          f"median bound {np.median(ds):.3f} (true {np.median(dt):.3f}); wave box side {np.median(qhi - qlo):.2f}, "
          f"sub-chunk box side {np.median(nhi - nlo):.2f}; scanned fraction: "
          + ", ".join(f"{k} {v:.4f}" for k, v in out.items()), flush=True)


def main():
    d = np.load(sys.argv[1])
    sub = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    bits = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    Q0, Q1, T = d["Q0"].astype(np.float64), d["Q1"].astype(np.float64), d["T"].astype(np.float64)
    rng = np.random.default_rng(1)
    study(Q1, T, sub, bits, "targets of the next round in the tree that grows")
    study(Q0, Q1[rng.choice(len(Q1), 131072, replace=False)], sub, bits, "nodes of one tree looked up in the other (connect phase)")


if __name__ == "__main__":
    main()
