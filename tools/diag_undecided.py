"""Which pairs does the float32 filter leave to the exact kernel?  The generated library against the interpreter on
one batch of a tests/spec_models.py model: undecided pairs by (geom a, geom b), those only one of the two reports."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from mjpl_amd import engine
from spec_models import spec_models

case = int(sys.argv[1]) if len(sys.argv) > 1 else 4
configs = len(sys.argv) > 2 and sys.argv[2] == "configs"  # a configuration launch over the edges' ends instead
name, m, allowed, qidx, base = spec_models()[case]
rng = np.random.default_rng(100 + case)
lo, hi = m.jnt_range[qidx, 0], m.jnt_range[qidx, 1]
n = 40000
qa = rng.uniform(lo, hi, size=(n, len(qidx)))
d = rng.normal(size=qa.shape)
qb = np.clip(qa + rng.choice([0.05, 0.05, 0.3], size=(n, 1)) * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)
res = {}
for tag, env in (("library", {}), ("interpreter", {"MJPL_SPEC": "0"})):
    os.environ.pop("MJPL_SPEC", None)
    os.environ.update(env)
    e = engine.Engine(m, allowed)
    e.set_planning(qidx, base)
    os.environ.pop("MJPL_SPEC", None)
    if configs:
        valid = e.check_configs(qb)
        fbad = np.where(valid, -1, 0)
    else:
        valid, fbad = e.check_edges(qa, qb, 0.01, first_bad=True)
    tot, edge, idx, ga, gb = e.undecided_pairs()
    print(tag, "spec" if e.spec_loaded() else "interp", "undecided", e.last_undecided(), "pairs", tot)
    recs = list(zip(edge.tolist(), idx.tolist(), ga.tolist(), gb.tolist()))
    res[tag] = set(recs)
    dup = collections.Counter(recs)
    dpairs = collections.Counter()
    for r, c in dup.items():
        if c > 1:
            dpairs[(r[2], r[3])] += c - 1
    print("   records handed off more than once:", sum(dpairs.values()), dict(dpairs.most_common(8)))
    where = collections.defaultdict(list)
    for pos, r in enumerate(recs):
        if dup[r] > 1:
            where[r].append(pos)
    print("   positions of the first of them in the list:", [(r, p) for r, p in list(where.items())[:6]])
    e.close()
both = res["library"] & res["interpreter"]
print("in both:", len(both), "at or behind the first bad waypoint:", sum(1 for x in both if 0 <= fbad[x[0]] <= x[1]))
names = [m.geom(i).name or f"g{i}:{int(m.geom_type[i])}" for i in range(m.ngeom)]
for a, b in (("library", "interpreter"), ("interpreter", "library")):
    only = res[a] - res[b]
    c = collections.Counter((names[x[2]] if x[2] >= 0 else "-", names[x[3]] if x[3] >= 0 else "-") for x in only)
    moot = sum(1 for x in only if 0 <= fbad[x[0]] <= x[1])  # the waypoint (or an earlier one) is in contact whatever this pair says
    print(f"only {a}: {len(only)}, of them at or behind the edge's first bad waypoint: {moot}")
    for k, v in c.most_common(25):
        print("   ", k, v)
