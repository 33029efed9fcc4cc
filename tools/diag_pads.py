"""Diagnostic (GPU box): mismatches of the finger-pad (moving boxes) model against the oracle, per edge."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
from mjpl_amd import engine as eng_mod, scenes
from oracle import pyoracle
from helpers import random_edges
m = scenes.franka_p(obstacles=True, pads=True)
base = m.keyframe("home").qpos.copy()
e = eng_mod.Engine(m)
rng = np.random.default_rng(3)
joints = scenes.FRANKA_ARM_JOINTS; qidx = scenes.planning_index(m, joints); b = base.copy(); b[7:] = rng.uniform(0, 0.04, size=2)
joints = scenes.FRANKA_ARM_JOINTS + ["finger_joint1", "finger_joint2"]
qidx = scenes.planning_index(m, joints); b = base.copy(); b[7:] = rng.uniform(0, 0.04, size=2)
e.set_planning(qidx, b)
orc = pyoracle.Oracle(m, planning_qidx=qidx, qpos_base=b)
qa, qb = random_edges(m, qidx, 30000, seed=len(joints))
Q = np.concatenate([qa, qb])
want = orc.valid_configs(Q, nthreads=8)
got = e.check_configs(Q)
bad = np.flatnonzero(got != want)
print("config mismatches", len(bad), "undecided", e.last_undecided(), e.info())
full = pyoracle.Oracle(m)
for i in bad[:8]:
    q = b.copy(); q[qidx] = Q[i]
    con = full.contacts(q)
    print(i, "got", got[i], "want", want[i], "contacts", [(m.geom(int(a)).name, m.geom(int(c)).name) for a, c in con][:6])
for env in ("MJPL_EXPAND", "MJPL_TWO_PASS"):
    os.environ[env] = "0"
    e2 = eng_mod.Engine(m); e2.set_planning(qidx, b)
    w2, _, _ = orc.valid_edges(qa, qb, 0.01, nthreads=8, info=True)
    g2 = e2.check_edges(qa, qb, 0.01)
    print(env, "=0 edge mismatches", int((g2 != w2).sum()))
    del os.environ[env]
e.set_filter(False)
got64 = e.check_configs(Q)
print("filter off mismatches", int((got64 != want).sum()), "same set as filter on:", np.array_equal(np.flatnonzero(got64 != want), bad))
with pyoracle.portable_trig():
    want_p = orc.valid_configs(Q, nthreads=8)
print("vs portable-trig oracle: filter on", int((got != want_p).sum()), "off", int((got64 != want_p).sum()))
fk = full.fk(np.array([np.r_[b[:0], b].copy() for _ in range(1)]))
for i in bad[:6]:
    q = b.copy(); q[qidx] = Q[i]
    f = full.fk(q[None])
    ga, gb = m.geom("left_finger_pad2").id, m.geom("right_finger_pad3").id
    d = f["geom_xpos"][0, gb] - f["geom_xpos"][0, ga]
    R1 = f["geom_xmat"][0, ga].reshape(3, 3); R2 = f["geom_xmat"][0, gb].reshape(3, 3)
    t = R1.T @ d
    R = R1.T @ R2
    s1, s2 = m.geom_size[ga], m.geom_size[gb]
    gaps = np.abs(t) - (s1 + np.abs(R) @ s2)
    print(i, "face gaps pad2/pad3", gaps, "R diag", np.diag(R))
