"""GPU parity: libmjpl_hip.so (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bar (BASELINE.json north_star): per-configuration / per-edge verdict bit-exact,
FK positions within 1e-6."""
import numpy as np
import pytest

from mjpl_amd import engine as eng_mod
from mjpl_amd import scenes

pytestmark = pytest.mark.gpu

FK_TOL = 1e-6  # north_star tolerance on FK positions


def _uniform_configs(model, n, seed, fingers=0.04):
    rng = np.random.default_rng(seed)
    Q = rng.uniform(model.jnt_range[:, 0], model.jnt_range[:, 1], size=(n, model.nq))
    if model.nq == 9:
        Q[:, 7:] = fingers
    return Q


def _edges(model, qidx, n, seed, eps=0.05):
    rng = np.random.default_rng(seed)
    lo, hi = model.jnt_range[qidx, 0], model.jnt_range[qidx, 1]
    qa = rng.uniform(lo, hi, size=(n, len(qidx)))
    d = rng.normal(size=(n, len(qidx)))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    qb = np.clip(qa + eps * d, lo, hi)
    return qa, qb


@pytest.fixture(scope="module")
def franka_obs():
    return scenes.franka_p(obstacles=True)


def test_kat_two_dof_ball(oracle_mod):
    # test/test_collision_constraint.py:16-33
    m = scenes.two_dof_ball()
    e = eng_mod.Engine(m)
    v = e.check_configs(np.array([[0.0, 0.0], [0.6, 0.0]]))
    assert v.tolist() == [1, 0]
    orc = oracle_mod.Oracle(m)
    rng = np.random.default_rng(0)
    Q = rng.uniform(-2, 2, size=(4096, 2))
    np.testing.assert_array_equal(e.check_configs(Q), orc.valid_configs(Q))


def test_kat_interval_one_dof(oracle_mod):
    # test/test_planning_utils.py:321-344
    m = scenes.one_dof_ball()
    e = eng_mod.Engine(m)
    qa = np.array([[0.8], [0.8], [0.0]])
    qb = np.array([[1.5], [1.5], [0.2]])
    assert e.check_edges(qa[:1], qb[:1], 0.1, interior_only=True).tolist() == [0]
    assert e.check_edges(qa[1:2], qb[1:2], 0.2, interior_only=True).tolist() == [1]
    assert e.check_edges(qa[2:], qb[2:], 0.01, interior_only=True).tolist() == [1]
    with pytest.raises(eng_mod.MjplError, match="step_dist"):
        e.check_edges(qa, qb, 0.0)


def test_franka_self_collision_64k(oracle_mod):
    """BASELINE config 2: 64k uniform configs, self-collision only, verdict bit-exact + FK."""
    m = scenes.franka_p()
    e = eng_mod.Engine(m)
    orc = oracle_mod.Oracle(m)
    Q = _uniform_configs(m, 65536, seed=1)
    got = e.check_configs(Q)
    want = orc.valid_configs(Q, nthreads=8)
    assert 0.05 < want.mean() < 0.95
    np.testing.assert_array_equal(got, want)
    # FK parity on a slice
    fk_g = e.fk(Q[:4096])
    fk_o = orc.fk(Q[:4096])
    for k in ("xpos", "xquat", "geom_xpos", "geom_xmat"):
        err = np.abs(fk_g[k] - fk_o[k]).max()
        assert err < FK_TOL, (k, err)
        assert err < 1e-12, (k, err)  # what the arithmetic actually delivers


def test_franka_planning_columns_layouts(oracle_mod, franka_obs):
    m = franka_obs
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    e = eng_mod.Engine(m)
    e.set_planning(qidx, base)
    orc = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=base)
    Q = _uniform_configs(m, 30000, seed=3)[:, qidx]
    want = orc.valid_configs(Q, nthreads=8)
    np.testing.assert_array_equal(e.check_configs(Q, layout=eng_mod.AOS), want)
    np.testing.assert_array_equal(e.check_configs(np.ascontiguousarray(Q.T), layout=eng_mod.SOA), want)
    assert 0.05 < want.mean() < 0.95


def test_franka_edges_vs_oracle(oracle_mod, franka_obs):
    """BASELINE config 3 at oracle-sized E: verdict and first-bad index bit-exact."""
    m = franka_obs
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    e = eng_mod.Engine(m)
    e.set_planning(qidx, base)
    orc = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=base)
    qa, qb = _edges(m, qidx, 20000, seed=2)
    want, want_fb, _ = orc.valid_edges(qa, qb, 0.01, nthreads=8, info=True)
    got, got_fb = e.check_edges(qa, qb, 0.01, first_bad=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got_fb, want_fb)
    assert 0.05 < want.mean() < 0.95
    # ragged: long edges with different waypoint counts per lane
    qa2, qb2 = _edges(m, qidx, 4096, seed=5, eps=0.4)
    qb2[::7] = qa2[::7]  # zero-length edges
    want2, fb2, _ = orc.valid_edges(qa2, qb2, 0.013, nthreads=8, info=True)
    got2, gfb2 = e.check_edges(qa2, qb2, 0.013, first_bad=True)
    np.testing.assert_array_equal(got2, want2)
    np.testing.assert_array_equal(gfb2, fb2)


def test_empty_and_tail(franka_obs):
    m = franka_obs
    e = eng_mod.Engine(m)
    assert e.check_configs(np.zeros((0, m.nq))).shape == (0,)
    Q = np.tile(m.keyframe("home").qpos, (257, 1))
    assert e.check_configs(Q).tolist() == [1] * 257
    info = e.info()
    assert info["arch"].startswith("gfx950")


def test_filter_and_exact_paths_agree_at_full_size(franka_obs):
    """BASELINE config 3 at full size (262 144 edges): the float32 filter + exact re-run and the
    pure float64 kernels must return identical verdicts and first-bad indices, and the filter
    must decide almost everything by itself."""
    m = franka_obs
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    e = eng_mod.Engine(m)
    e.set_planning(qidx, m.keyframe("home").qpos.copy())
    qa, qb = _edges(m, qidx, 262144, seed=2)
    e.set_filter(True, 1e-4)
    v1, f1 = e.check_edges(qa, qb, 0.01, first_bad=True)
    undecided = e.last_undecided()
    e.set_filter(False)
    v0, f0 = e.check_edges(qa, qb, 0.01, first_bad=True)
    np.testing.assert_array_equal(v1, v0)
    np.testing.assert_array_equal(f1, f0)
    assert 0 < undecided < 0.02 * len(qa)
    # size-independent property: an edge is valid iff its endpoint and every interior waypoint
    # config is valid -> re-check a slice through the configuration kernel
    sl = slice(0, 4096)
    end_ok = e.check_configs(qb[sl])
    assert np.all(v0[sl] <= end_ok)  # a valid edge has a valid endpoint
    assert np.all((f0[sl] == 0) == (end_ok == 0))
    # a much tighter tolerance still gives the same answers (the band only moves work)
    e.set_filter(True, 2e-6)
    v2 = e.check_edges(qa[:65536], qb[:65536], 0.01)
    np.testing.assert_array_equal(v2, v0[:65536])


def test_filter_configs_64k_and_info(oracle_mod):
    m = scenes.franka_p()
    e = eng_mod.Engine(m)
    info = e.info()
    assert info["filter_enabled"] == 1 and abs(info["filter_tol"] - 1e-4) < 1e-9
    Q = _uniform_configs(m, 65536, seed=7)
    want = oracle_mod.Oracle(m).valid_configs(Q, nthreads=8)
    np.testing.assert_array_equal(e.check_configs(Q), want)
    e.set_filter(False)
    np.testing.assert_array_equal(e.check_configs(Q), want)
    with pytest.raises(eng_mod.MjplError, match="tolerance"):
        e.set_filter(True, 0.0)


def test_two_pass_and_one_pass_filter_agree(monkeypatch):
    """The endpoint-first split of the edge filter and the lane-per-waypoint interior pass change
    scheduling only: valid and first_bad are those of the one-pass filter, of the walking interior
    pass, of an overflowing item buffer and of the pure float64 kernels, on clean edges, on edges
    that end in an obstacle and on zero-length edges."""
    from mjpl_amd import engine, scenes
    import bench as _bench
    m = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    qa, qb = _bench.make_edges(m, qidx, 20000, seed=77)
    qb[:50] = qa[:50]  # waypoints == [start]: nothing interior
    out = {}
    # ("two": the fused kernel, one launch for endpoints and waypoints; "kernels2": the two persistent kernels it
    # replaced; the MJPL_FUSED_* switches: a ring of pool entries that wraps, waypoint tiles first, endpoints as items
    # of their own -- the mode of small batches -- and not, a waypoint limit that sends most edges to the walking list)
    keys = ("MJPL_TWO_PASS", "MJPL_FILTER", "MJPL_EXPAND", "MJPL_ITEM_CAP", "MJPL_FUSED", "MJPL_FUSED_POOL",
            "MJPL_FUSED_POLICY", "MJPL_FUSED_SINGLE", "MJPL_FUSED_KMAX")
    for tag, env in (("two", {"MJPL_TWO_PASS": "1", "MJPL_FUSED_SINGLE": "0"}), ("auto", {}), ("one", {"MJPL_TWO_PASS": "0"}), ("f64", {"MJPL_FILTER": "0"}),
                     ("f64_lane_per_edge", {"MJPL_FILTER": "0", "MJPL_FUSED": "0"}), ("f64_items_first", {"MJPL_FILTER": "0", "MJPL_FUSED_POLICY": "1"}),
                     ("walk", {"MJPL_EXPAND": "0"}), ("kernels2", {"MJPL_FUSED": "0"}),
                     ("tight", {"MJPL_ITEM_CAP": "3000", "MJPL_FUSED": "0"}),
                     ("pool", {"MJPL_FUSED_POOL": "832"}), ("items_first", {"MJPL_FUSED_POLICY": "1"}),
                     ("single", {"MJPL_FUSED_SINGLE": "100000000"}), ("rounds2", {"MJPL_FUSED_SINGLE": "0"}),
                     ("kmax3", {"MJPL_FUSED_KMAX": "3"}), ("single_kmax3", {"MJPL_FUSED_SINGLE": "100000000", "MJPL_FUSED_KMAX": "3"})):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = engine.Engine(m)
        e.set_planning(qidx, base)
        out[tag] = e.check_edges(qa, qb, 0.01, first_bad=True)
        if tag in ("two", "kernels2"):
            assert bool(e.info()["fused_edges"]) == (tag == "two")
            n_int = e.last_interior_edges()
            assert 0 < n_int < len(qa)
            assert e.last_items() > n_int  # one lane per interior waypoint
        if tag == "tight":  # the item buffer overflows: the rest of the edges take the walking kernel
            assert e.last_items() > 3000
        if tag == "kmax3":  # edges of four waypoints take the walking list
            assert e.last_items() < out["two"][2]
        out[tag] = out[tag] + ((e.last_items() if e.info()["filter_enabled"] else 0),)
        e.close()
    for tag in [t for t in out if t != "two"]:
        np.testing.assert_array_equal(out["two"][0], out[tag][0])
        np.testing.assert_array_equal(out["two"][1], out[tag][1])
    assert 0 < out["two"][0].sum() < len(qa)


def test_waypoint_recurrence_corner_cases(oracle_mod):
    """Edges whose length is zero, tiny, below one step, an exact multiple of the step (where
    rounding decides whether one more waypoint appears) or many steps long: the waypoint items the
    endpoint kernel emits, the walking kernel and the oracle must agree on valid AND first_bad,
    which is sensitive to every single waypoint (a wall in the middle of the 1-DoF scene)."""
    m = scenes.one_dof_ball()
    orc = oracle_mod.Oracle(m)
    rng = np.random.default_rng(0)
    step = 0.01
    starts, ends = [], []
    for a in np.concatenate([np.linspace(0.0, 0.84, 64), rng.uniform(0.0, 0.84, 192)]):
        for length in (0.0, 1e-150, 1e-12, 0.5 * step, step, np.nextafter(step, 0), np.nextafter(step, 1),
                       3 * step, np.nextafter(3 * step, 1), 7 * step, 0.05, 0.1, 0.17, 0.3, 0.7, 1.3):
            starts.append([a])
            ends.append([a + length])
    qa, qb = np.array(starts), np.array(ends)
    want, wfb, ncheck = orc.valid_edges(qa, qb, step, nthreads=4, info=True)
    assert 0 < want.sum() < len(want) and ncheck.max() > 50
    import os
    mode_keys = ("MJPL_EXPAND", "MJPL_TWO_PASS", "MJPL_ITEM_CAP", "MJPL_FUSED", "MJPL_FUSED_SINGLE", "MJPL_FUSED_KMAX", "MJPL_FUSED_POOL")
    for env in ({}, {"MJPL_EXPAND": "0"}, {"MJPL_TWO_PASS": "0"}, {"MJPL_FUSED": "0"}, {"MJPL_ITEM_CAP": "500", "MJPL_FUSED": "0"},
                {"MJPL_FUSED_SINGLE": "0"}, {"MJPL_FUSED_KMAX": "5"}, {"MJPL_FUSED_KMAX": "5", "MJPL_FUSED_SINGLE": "0"},
                {"MJPL_FUSED_POOL": "832"}):
        old = {k: os.environ.pop(k, None) for k in mode_keys}
        os.environ.update(env)
        try:
            e = eng_mod.Engine(m)
            got, gfb = e.check_edges(qa, qb, step, first_bad=True)
            gi = e.check_edges(qa, qb, step, interior_only=True)
        finally:
            for k in mode_keys:
                os.environ.pop(k, None)
                if old[k] is not None:
                    os.environ[k] = old[k]
        np.testing.assert_array_equal(got, want, err_msg=str(env))
        np.testing.assert_array_equal(gfb, wfb, err_msg=str(env))
        wi = np.array([orc.valid_collision_interval(a, b, step) for a, b in zip(qa[::5], qb[::5])])
        np.testing.assert_array_equal(gi[::5].astype(bool), wi, err_msg=str(env))
        e.close()


def test_underflowing_edge_is_reported_not_walked(oracle_mod):
    """An edge so short that its squared length underflows makes the reference's recurrence
    divide by zero and never end; the oracle and the engine both report it as a non-finite edge
    (and the engine does so at once, in every interior-pass mode)."""
    import os
    import time
    m = scenes.one_dof_ball()
    qa = np.array([[0.1], [0.2], [0.3]])
    qb = np.array([[0.1 + 1e-3], [0.2 + 1e-170], [0.3]])
    qb[1, 0] = np.nextafter(0.2, 1.0)  # one ulp: (2.8e-17)^2 is fine ...
    qa2 = np.array([[0.0], [1e-170], [0.3]])  # ... but 1e-170 next to zero is not
    qb2 = np.array([[1e-3], [2e-170], [0.3]])
    orc = oracle_mod.Oracle(m)
    with pytest.raises(RuntimeError):
        orc.valid_edges(qa2, qb2, 0.01)
    for env in ({}, {"MJPL_EXPAND": "0"}, {"MJPL_TWO_PASS": "0"}, {"MJPL_FILTER": "0"}, {"MJPL_FUSED": "0"}, {"MJPL_FUSED_SINGLE": "0"}):
        old = {k: os.environ.pop(k, None) for k in ("MJPL_EXPAND", "MJPL_TWO_PASS", "MJPL_FILTER", "MJPL_FUSED", "MJPL_FUSED_SINGLE")}
        os.environ.update(env)
        try:
            e = eng_mod.Engine(m)
            assert e.check_edges(qa, qb, 0.01).tolist() == [1, 1, 1]
            t0 = time.time()
            with pytest.raises(eng_mod.MjplError, match="NaN/inf"):
                e.check_edges(qa2, qb2, 0.01)
            assert time.time() - t0 < 5.0
            e.close()
        finally:
            for k in ("MJPL_EXPAND", "MJPL_TWO_PASS", "MJPL_FILTER", "MJPL_FUSED", "MJPL_FUSED_SINGLE"):
                os.environ.pop(k, None)
                if old[k] is not None:
                    os.environ[k] = old[k]


def test_absurdly_long_edge_is_refused_at_once():
    """More than 2^20 waypoints: an error, not a kernel that walks them all."""
    import time
    m = scenes.one_dof_ball()
    e = eng_mod.Engine(m)
    t0 = time.time()
    with pytest.raises(eng_mod.MjplError, match="waypoints"):
        e.check_edges(np.array([[0.0]]), np.array([[1.5]]), 1e-7)
    assert time.time() - t0 < 5.0
    assert e.check_edges(np.array([[0.0]]), np.array([[0.5]]), 1e-4).tolist() == [1]  # 5 000 waypoints are fine
