"""Diagnostic (GPU box): verdicts of a model's specialised kernels against the interpreter and the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
from mjpl_amd import engine as eng_mod, scenes
from oracle import pyoracle
m = scenes.franka_p(obstacles=True)
qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
base = m.keyframe("home").qpos.copy()
e = eng_mod.Engine(m); e.set_planning(qidx, base)
print("spec loaded:", e.spec_loaded())
orc = pyoracle.Oracle(m, planning_qidx=qidx, qpos_base=base)
rng = np.random.default_rng(3)
Q = rng.uniform(m.jnt_range[qidx, 0], m.jnt_range[qidx, 1], size=(30000, 7))
want = orc.valid_configs(Q, nthreads=8)
got = e.check_configs(Q)
bad = np.flatnonzero(got != want)
print("mismatches", len(bad), "undecided", e.last_undecided(), "valid frac", want.mean(), got.mean())
full = pyoracle.Oracle(m)
from collections import Counter
c = Counter()
for i in bad[:200]:
    q = base.copy(); q[qidx] = Q[i]
    con = full.contacts(q)
    for a, b in con:
        c[(m.geom(int(a)).name, m.geom(int(b)).name, int(got[i]))] += 1
    if not len(con):
        c[("none", "none", int(got[i]))] += 1
print(c.most_common(15))
g = Counter()
for i in bad:
    q = base.copy(); q[qidx] = Q[i]
    names = set()
    for a, b in full.contacts(q):
        names.add(m.geom(int(a)).name); names.add(m.geom(int(b)).name)
    moving = sorted(n for n in names if n.endswith("_c") and n != "link0_c")
    g[tuple(moving)] += 1
print(g.most_common(20))
print("got=1,want=0:", int(((got == 1) & (want == 0)).sum()), " got=0,want=1:", int(((got == 0) & (want == 1)).sum()))
for i in bad[:6]:
    one = np.repeat(Q[i][None], 64, axis=0)
    r1 = e.check_configs(one)
    mix = Q[i - 32: i + 32].copy()
    r2 = e.check_configs(mix)
    solo = e.check_configs(Q[i][None])
    print(i, "replicated x64 ->", r1[:4], " in its neighbourhood ->", r2[32], " alone ->", solo, " want", want[i])
# which lanes of a wave mismatch?
print("lane (i % 64) histogram of mismatches:", np.bincount(bad % 64, minlength=64))
