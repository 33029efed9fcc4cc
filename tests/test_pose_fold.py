"""The generated chain (mjpl_amd/fold.py, specialise.generate_pose) on the CPU: the folded straight-line code, compiled
for the host with the library's floating-point flags, against (1) the same generator with folding off -- every operation
of pose_chain's statement executed -- and (2) the oracle's chain (oracle/mjpl_oracle_pose.c: chain_kinematics).
Equality is by value (`==`): folding products with an exact zero may change the sign of an exact zero, nothing else."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import mjpl_amd as mjpl
from mjpl_amd import build as _build
from mjpl_amd import scenes
from mjpl_amd import specialise as sp
from mjpl_amd.fold import Fold, unit_interval

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

_HARNESS = r"""
#include <cmath>
#include <cstdint>
#include "mjpl_trig.h"
namespace mjpl {
struct PoseChainOut { double site_xpos[3], site_xmat[9]; };
enum : int { PT_SITE_POS = 0, PT_SITE_QUAT = 3 };
}
namespace folded {
%(folded)s
}
namespace plain {
%(plain)s
}
template <class PS>
static void run(const double *q_, const double *tail, double *xpos, double *xmat, double *jx_) {
  double q[PS::kNQ], sn[PS::kNJ > 0 ? PS::kNJ : 1], cs[PS::kNJ > 0 ? PS::kNJ : 1], jx[PS::kNJ > 0 ? PS::kNJ : 1][6];
  for (int k = 0; k < PS::kNQ; k++) q[k] = q_[k];
  for (int k = 0; k < PS::kNJ; k++) {
    sn[k] = 0; cs[k] = 1;
    if (PS::jtype(k) == 3) mjpl::sincos_pi2(PS::half_angle(k, q), &sn[k], &cs[k]);
  }
  mjpl::PoseChainOut o;
  PS::chain(q, sn, cs, jx, o, tail);
  for (int k = 0; k < 3; k++) xpos[k] = o.site_xpos[k];
  for (int k = 0; k < 9; k++) xmat[k] = o.site_xmat[k];
  for (int k = 0; k < PS::kNJ; k++) for (int r = 0; r < 6; r++) jx_[6 * k + r] = jx[k][r];
}
extern "C" void chain_folded(const double *q, const double *tail, double *xpos, double *xmat, double *jx) { run<folded::PoseSpec0>(q, tail, xpos, xmat, jx); }
extern "C" void chain_plain(const double *q, const double *tail, double *xpos, double *xmat, double *jx) { run<plain::PoseSpec0>(q, tail, xpos, xmat, jx); }
"""


def _compile(model, site_body, tmp_path, tag):
    pi, pd, h = sp.dump_pose_chain(model, site_body)
    kw = dict(qualifier="static inline")
    src = _HARNESS % dict(folded=sp.generate_pose(pi, pd, h, 0, **kw), plain=sp.generate_pose(pi, pd, h, 0, fold=False, **kw))
    cpp, so = tmp_path / f"chain_{tag}.cpp", tmp_path / f"chain_{tag}.so"
    cpp.write_text(src)
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-shared", "-fPIC", f"-I{_build.CSRC}", "-o", str(so), str(cpp)],
                   check=True)
    lib = C.CDLL(str(so))
    nj = int(pi[sp.PH_NJOINT])
    P = C.POINTER(C.c_double)

    def call(fn, q, tail):
        q, tail = np.ascontiguousarray(q, np.float64), np.ascontiguousarray(tail, np.float64)
        xpos, xmat, jx = np.zeros(3), np.zeros(9), np.zeros((max(nj, 1), 6))
        getattr(lib, fn)(q.ctypes.data_as(P), tail.ctypes.data_as(P), xpos.ctypes.data_as(P), xmat.ctypes.data_as(P), jx.ctypes.data_as(P))
        return xpos, xmat, jx[:nj]
    return call, nj


def _cases():
    from test_gpu_models import random_model  # (no GPU needed to build a model)
    yield "franka", scenes.franka_p(obstacles=False), "ee_site"
    yield "ur5e", scenes.ur5e(), "attachment_site"
    yield "ball", scenes.two_dof_ball(), "ball_site"
    import dataclasses
    for seed in (3, 8):
        model, _ = random_model(seed, moving_boxes=False)
        q = np.array([0.5, 0.5, -0.5, 0.5])
        model = dataclasses.replace(model, nsite=1, site_bodyid=np.array([model.nbody - 1], np.int32),
                                    site_pos=np.array([[0.05, -0.02, 0.08]]), site_quat=q[None] / np.linalg.norm(q), site_names=["tip"])
        yield f"random{seed}", model, "tip"


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_folded_chain_equals_the_statement_and_the_oracle(case, oracle_mod, tmp_path):
    tag, model, site = case
    sid = model.site_names.index(site) if hasattr(model, "site_names") else model.site(site).id
    body = int(np.asarray(model.site_bodyid).reshape(-1)[sid])
    call, nj = _compile(model, body, tmp_path, tag)
    tail = np.concatenate([np.asarray(model.site_pos, np.float64).reshape(-1, 3)[sid], np.asarray(model.site_quat, np.float64).reshape(-1, 4)[sid]])
    ident = (np.array([1.0, 0, 0, 0]), np.zeros(3))
    po = oracle_mod.PoseOracle(model, site, ident, [(-np.inf, np.inf)] * 6)
    rng = np.random.default_rng(5)
    lo, hi = np.asarray(model.jnt_range)[:, 0], np.asarray(model.jnt_range)[:, 1]
    Q = rng.uniform(lo, hi, size=(400, model.nq))
    Q[0] = np.asarray(model.qpos0, np.float64)
    Q[1] = 0.0
    Q[2], Q[3] = lo, hi
    Q[4:40] = np.asarray(model.qpos0, np.float64) + rng.normal(scale=1e-3, size=(36, model.nq))
    with oracle_mod.portable_trig():
        for q in Q:
            a, b = call("chain_folded", q, tail), call("chain_plain", q, tail)
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (tag, q)
            p, R = po.site_pose(q)
            assert np.array_equal(a[0], np.asarray(p)) and np.array_equal(a[1], np.asarray(R).reshape(-1)), (tag, q)


def test_unit_interval_is_the_statements_condition():
    import math
    lo, hi = unit_interval(1e-15)

    def inside(s):
        n = math.sqrt(s)
        return not (n < 1e-15) and not (abs(n - 1.0) > 1e-15)
    x = lo
    while x <= hi:
        assert inside(x)
        x = math.nextafter(x, math.inf)
    for edge, d in ((lo, -math.inf), (hi, math.inf)):
        x = edge
        for _ in range(500):
            x = math.nextafter(x, d)
            assert not inside(x)
    for s in (0.0, 1e-31, 0.5, 2.0, math.inf):
        assert not inside(s) and not (lo <= s <= hi)


def test_fold_rules():
    f = Fold()
    a, b = f.v("a"), f.v("b")
    assert f.mul(a, f.c(0.0)) == ("c", 0.0) and f.mul(f.c(-0.0), b) == ("c", 0.0)
    assert f.mul(a, f.c(1.0)) == a and f.mul(f.c(-1.0), a) == f.neg(a)
    assert f.add(a, f.c(0.0)) == a and f.sub(a, f.c(0.0)) == a and f.sub(f.c(0.0), a) == f.neg(a)
    assert f.mul(f.c(3.0), f.c(0.1)) == ("c", 3.0 * 0.1)
    assert not f.lines
    t = f.sub(f.mul(f.neg(a), b), f.mul(a, f.c(-2.5)))  # -(a b) + 2.5 a
    assert f.text(t) == "t2"
    assert f.lines == ["    const double t0 = a * b;", "    const double t1 = a * 0x1.4000000000000p+1;", "    const double t2 = t1 - t0;"]


_EXACT_HARNESS = r"""
#include <cmath>
#include <cstdint>
#include <cstddef>
#include "mjpl_trig.h"
#define __device__
#define __forceinline__ inline
#define __ballot(x) ((x) ? 1ull : 0ull)
namespace mjpl {
template <class T> struct GeomT { T pos[3], m[9]; };
static inline void sincos_half(double x, double *s, double *c) { sincos_pi2(x, s, c); }
}
namespace folded {
%(folded)s
}
namespace plain {
%(plain)s
}
template <class ES>
static void run(const double *q, double *save, int ga, int gb, double *out) {
  mjpl::GeomT<double> A = {}, B = {};
  ES::fk_pair(q, 1, save, 1, true, ga, gb, A, B);
  for (int k = 0; k < 3; k++) { out[k] = A.pos[k]; out[12 + k] = B.pos[k]; }
  for (int k = 0; k < 9; k++) { out[3 + k] = A.m[k]; out[15 + k] = B.m[k]; }
}
extern "C" void pair_folded(const double *q, double *save, int ga, int gb, double *out) { run<folded::ExactSpec>(q, save, ga, gb, out); }
extern "C" void pair_plain(const double *q, double *save, int ga, int gb, double *out) { run<plain::ExactSpec>(q, save, ga, gb, out); }
"""


@pytest.mark.parametrize("name", ["franka_p", "pads", "ur5e"])
def test_folded_exact_fk_equals_the_statement_and_the_oracle(name, oracle_mod, tmp_path):
    """ExactSpec::fk_pair (the float64 FK of the pair re-check) with the constants folded in, against the same generator
    with folding off and against the oracle's FK: geom positions and frames of random pairs, equal value by value."""
    model = {"franka_p": lambda: scenes.franka_p(True), "pads": lambda: scenes.franka_p(True, True), "ur5e": scenes.ur5e}[name]()
    if name == "ur5e":
        qidx, base = None, None
    else:
        qidx, base = scenes.planning_index(model, scenes.FRANKA_ARM_JOINTS), model.keyframe("home").qpos.copy()
    ip, fp, dp, info = sp.dump_program(model, (), qidx, base)
    src = _EXACT_HARNESS % dict(folded=sp.generate_exact(ip, dp, info), plain=sp.generate_exact(ip, dp, info, fold=False))
    cpp, so = tmp_path / f"exact_{name}.cpp", tmp_path / f"exact_{name}.so"
    cpp.write_text(src)
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-shared", "-fPIC", "-Wno-attributes", f"-I{_build.CSRC}",
                    "-o", str(so), str(cpp)], check=True)
    lib = C.CDLL(str(so))
    P = C.POINTER(C.c_double)
    moving = [g for g in range(model.ngeom) if int(model.body_weldid[model.geom_bodyid[g]]) != 0]
    rng = np.random.default_rng(9)
    nplan = model.nq if qidx is None else len(qidx)
    lo = model.jnt_range[:, 0] if qidx is None else model.jnt_range[qidx, 0]
    hi = model.jnt_range[:, 1] if qidx is None else model.jnt_range[qidx, 1]
    orc = oracle_mod.Oracle(model, planning_qidx=qidx, qpos_base=base)
    save = np.zeros(7 * 64)
    with oracle_mod.portable_trig():
        for trial in range(120):
            q = rng.uniform(lo, hi) if trial > 1 else (np.zeros(nplan) if trial else 0.5 * (lo + hi))
            q = np.ascontiguousarray(q, np.float64)
            ga, gb = (int(x) for x in rng.choice(moving, 2, replace=False))
            a, b = np.zeros(24), np.zeros(24)
            lib.pair_folded(q.ctypes.data_as(P), save.ctypes.data_as(P), ga, gb, a.ctypes.data_as(P))
            lib.pair_plain(q.ctypes.data_as(P), save.ctypes.data_as(P), ga, gb, b.ctypes.data_as(P))
            assert np.array_equal(a, b), (name, trial, ga, gb)
            fk = orc.fk(q[None])
            gx, gm = np.asarray(fk["geom_xpos"]).reshape(-1, 3), np.asarray(fk["geom_xmat"]).reshape(-1, 9)
            assert np.array_equal(a[0:3], gx[ga]) and np.array_equal(a[12:15], gx[gb]), (name, trial)
            assert np.array_equal(a[3:12], gm[ga]) and np.array_equal(a[15:24], gm[gb]), (name, trial)


def test_certificate_levers_bound_every_geoms_motion(oracle_mod, monkeypatch):
    """The edge certificate of the fused kernel (mjpl_fused.h; specialise._generate: cert_levers) rests on one claim: while
    the planning joints travel from QA to QB in a straight line, no point of moving geom g moves farther than
    sum_j |dq_j| rho_j(g).  Checked against the oracle's FK along densely sampled edges: the path length of every geom's
    centre plus what its orientation can add (bounding radius x the rotation its frame turned by), and the lever arms
    themselves (no joint anchor is ever farther from a geom's centre than rho - bounding radius)."""
    monkeypatch.setenv("MJPL_SPEC_CERT", "1")
    model = scenes.franka_p(True)
    qidx, base = scenes.planning_index(model, scenes.FRANKA_ARM_JOINTS), model.keyframe("home").qpos.copy()
    ip, fp, dp, info = sp.dump_program(model, (), qidx, base)
    rep = {}
    sp.generate(ip, fp, dp, info, report=rep)
    assert rep["cert_ok"] and len(rep["cert_levers"]) == 10
    orc = oracle_mod.Oracle(model, planning_qidx=qidx, qpos_base=base)
    rng = np.random.default_rng(17)
    lo, hi = model.jnt_range[qidx, 0], model.jnt_range[qidx, 1]
    jbody = [int(model.jnt_bodyid[j]) for j in range(len(qidx))]
    T = 65
    for trial in range(60):
        qa = rng.uniform(lo, hi)
        d = rng.normal(size=len(qidx))
        qb = np.clip(qa + rng.choice([0.05, 0.3, 1.0]) * d / np.linalg.norm(d), lo, hi)
        path = qa[None] + np.linspace(0.0, 1.0, T)[:, None] * (qb - qa)[None]
        fk = orc.fk(path)
        gx = np.asarray(fk["geom_xpos"]).reshape(T, -1, 3)
        gm = np.asarray(fk["geom_xmat"]).reshape(T, -1, 3, 3)
        xpos = np.asarray(fk["xpos"]).reshape(T, -1, 3)
        adq = np.abs(qb - qa)
        for geom_id, levers, rbound in rep["cert_levers"]:
            bound = sum(adq[qs] * rho for qs, rho in levers)
            centre = np.linalg.norm(np.diff(gx[:, geom_id], axis=0), axis=1).sum()
            # the frame's total turning angle along the path: no point of the geom is farther than rbound from its centre
            rel = np.einsum("tij,tkj->tik", gm[1:, geom_id], gm[:-1, geom_id])
            ang = np.arccos(np.clip((np.trace(rel, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0)).sum()
            assert centre + rbound * ang <= bound * (1 + 1e-9) + 1e-12, (trial, geom_id, centre, ang, bound)
            for qs, rho in levers:
                arm = np.linalg.norm(gx[:, geom_id] - xpos[:, jbody[qs]], axis=1).max()
                assert arm + rbound <= rho * (1 + 1e-9), (trial, geom_id, qs, arm, rho)
