"""Per-model specialised filter kernels (mjpl_amd/specialise.py): an engine
that finds the library of its program's hash must return, bit for bit, what the interpreting kernels
and the CPU oracle return -- configurations, edges with first-bad indices, layouts, every
interior-pass mode -- on every prebuilt (model, planning set) of tests/spec_models.py."""
import os

import numpy as np
import pytest

from mjpl_amd import engine as eng_mod
from spec_models import spec_models

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


@pytest.mark.parametrize("case", range(len(spec_models())))
def test_specialised_kernels_equal_interpreter_and_oracle(oracle_mod, case):
    name, m, allowed, qidx, base = spec_models()[case]
    rng = np.random.default_rng(100 + case)
    lo, hi = m.jnt_range[qidx, 0], m.jnt_range[qidx, 1]
    n = 40000
    qa = rng.uniform(lo, hi, size=(n, len(qidx)))
    d = rng.normal(size=qa.shape)
    qb = np.clip(qa + rng.choice([0.05, 0.05, 0.3], size=(n, 1)) * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)
    qb[::97] = qa[::97]  # zero-length edges
    with oracle_mod.portable_trig():  # clipping puts joints exactly on their limits: see test_gpu_regressions.py
        orc = oracle_mod.Oracle(m, allowed, planning_qidx=qidx, qpos_base=base)
        want, wfb, _ = orc.valid_edges(qa, qb, 0.01, nthreads=8, info=True)
        wantc = orc.valid_configs(qb, nthreads=8)
    spec = eng_mod.Engine(m, allowed)
    spec.set_planning(qidx, base)
    assert spec.spec_loaded(), f"{name}: no specialised library for this program (run __graft_entry__.build())"
    with _Env(MJPL_SPEC="0"):
        interp = eng_mod.Engine(m, allowed)
        interp.set_planning(qidx, base)
    assert not interp.spec_loaded()
    for e in (spec, interp):
        got, gfb = e.check_edges(qa, qb, 0.01, first_bad=True)
        np.testing.assert_array_equal(got, want, err_msg=name)
        np.testing.assert_array_equal(gfb, wfb, err_msg=name)
        np.testing.assert_array_equal(e.check_configs(qb), wantc, err_msg=name)
        np.testing.assert_array_equal(e.check_configs(np.ascontiguousarray(qb.T), layout=eng_mod.SOA), wantc, err_msg=name)
    assert 0.02 < want.mean() < 0.98, name
    # the filter decides about as much with either code (same culls, same narrowphase).  Counted on a configuration
    # launch: on edges the count also holds pairs of waypoints behind an edge's first bad one, and how many of those are
    # looked at depends on the order the waves take waypoints in (waves per workgroup differ between the two).
    spec.check_configs(qb)
    u_spec = spec.last_undecided()
    interp.check_configs(qb)
    u_int = interp.last_undecided()
    assert abs(u_spec - u_int) <= 0.1 * max(u_spec, u_int) + 20, (u_spec, u_int)
    # changing the planning set changes the program: the engine falls back to the interpreter
    if len(qidx) > 2:
        spec.set_planning(qidx[:-1], base)
        assert not spec.spec_loaded()
        spec.set_planning(qidx, base)
        assert spec.spec_loaded()
    spec.close()
    interp.close()


def test_specialised_kernels_in_every_interior_pass_mode(oracle_mod):
    name, m, allowed, qidx, base = spec_models()[0]
    rng = np.random.default_rng(5)
    lo, hi = m.jnt_range[qidx, 0], m.jnt_range[qidx, 1]
    qa = rng.uniform(lo, hi, size=(20000, len(qidx)))
    d = rng.normal(size=qa.shape)
    qb = np.clip(qa + 0.05 * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)
    with oracle_mod.portable_trig():
        want, wfb, _ = oracle_mod.Oracle(m, allowed, planning_qidx=qidx, qpos_base=base).valid_edges(qa, qb, 0.01, nthreads=8, info=True)
    # (MJPL_PERSIST: the endpoint / item kernels as persistent grids of waves with tile queues -- the default
    # with a specialised library -- or as ordinary grids; with the interpreter it is the other way round)
    # ... and since round 4 one fused kernel in their place (MJPL_FUSED=0 restores them), whose own switches follow
    F0 = {"MJPL_FUSED": "0"}
    for env in ({}, {"MJPL_EXPAND": "0"}, {"MJPL_TWO_PASS": "0"}, {"MJPL_UC_CAP": "16"}, {"MJPL_SPEC": "0"},
                {"MJPL_FUSED_POOL": "832"}, {"MJPL_FUSED_POLICY": "1"}, {"MJPL_FUSED_SINGLE": "100000000"}, {"MJPL_FUSED_SINGLE": "0"},
                {"MJPL_FUSED_KMAX": "3", "MJPL_UC_CAP": "16"}, {"MJPL_FUSED_SINGLE": "100000000", "MJPL_SPEC": "0"},
                F0, dict(F0, MJPL_ITEM_CAP="3000"), dict(F0, MJPL_UC_CAP="16"),
                dict(F0, MJPL_PERSIST="0"), dict(F0, MJPL_PERSIST="0", MJPL_ITEM_CAP="3000"), dict(F0, MJPL_PERSIST="1", MJPL_UC_CAP="16"),
                dict(F0, MJPL_PERSIST="1", MJPL_SPEC="0"), dict(F0, MJPL_PERSIST="1", MJPL_SPEC="0", MJPL_ITEM_CAP="3000")):
        with _Env(**env):
            e = eng_mod.Engine(m, allowed)
            e.set_planning(qidx, base)
        assert e.spec_loaded() == (env.get("MJPL_SPEC") != "0")
        got, gfb = e.check_edges(qa, qb, 0.01, first_bad=True)
        np.testing.assert_array_equal(got, want, err_msg=str(env))
        np.testing.assert_array_equal(gfb, wfb, err_msg=str(env))
        e.close()


def test_a_program_with_other_float64_constants_does_not_load_this_library():
    """Finger opening 0.04 vs 0.04 + 1e-10: the same float32 image, one float64 constant apart.  The
    library of the first program bakes ITS float64 constants into the exact pair re-check, so the
    second program must not find it (the hash once covered the float32 image only) --
    and its verdicts, from the interpreting kernels, equal the float64 path's."""
    name, m, allowed, qidx, base = spec_models()[0]
    e = eng_mod.Engine(m, allowed)
    e.set_planning(qidx, base)
    assert e.spec_loaded()
    base2 = base.copy()
    base2[7] += 1e-10
    e.set_planning(qidx, base2)
    assert not e.spec_loaded()
    rng = np.random.default_rng(5)
    Q = rng.uniform(m.jnt_range[qidx, 0], m.jnt_range[qidx, 1], size=(20000, len(qidx)))
    got = e.check_configs(Q)
    e.set_filter(False)
    np.testing.assert_array_equal(got, e.check_configs(Q))
    e.set_planning(qidx, base)
    e.set_spec(False)
    assert not e.spec_loaded()
    e.set_spec(True)
    assert e.spec_loaded()
    e.close()
