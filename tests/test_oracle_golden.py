"""The oracle against (1) the reference tests' own known answers and (2) independent
closed-form FK fixtures.  CPU only.  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest

from mjpl_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def _num(x):
    return float("inf") if x == "inf" else x


KAT = _load("kat_reference.json")
SCENES = {"one_dof_ball": scenes.one_dof_ball, "two_dof_ball": scenes.two_dof_ball}


@pytest.mark.parametrize("case", KAT["valid_config"], ids=lambda c: str(c["q"]))
def test_kat_valid_config(oracle_mod, case):
    orc = oracle_mod.Oracle(SCENES[case["scene"]]())
    assert orc.valid_config(case["q"]) is case["valid"]


@pytest.mark.parametrize("case", KAT["valid_collision_interval"], ids=lambda c: f"{c['start']}-{c['end']}@{c['step']}")
def test_kat_interval(oracle_mod, case):
    orc = oracle_mod.Oracle(SCENES[case["scene"]]())
    assert orc.valid_collision_interval(case["start"], case["end"], case["step"]) is case["valid"]


def test_kat_interval_bad_step(oracle_mod):
    orc = oracle_mod.Oracle(scenes.one_dof_ball())
    for bad in (0.0, -1.0):
        with pytest.raises(ValueError, match="step_dist"):
            orc.valid_collision_interval([0.0], [0.2], bad)


@pytest.mark.parametrize("case", KAT["step"], ids=lambda c: str(c["max_step"]))
def test_kat_step(oracle_mod, case):
    got = oracle_mod.step(case["start"], case["target"], _num(case["max_step"]))
    np.testing.assert_allclose(got, case["expect"], rtol=0, atol=case["atol"])
    with pytest.raises(ValueError, match="`max_step_dist` must be > 0.0"):
        oracle_mod.step(case["start"], case["target"], 0.0)


def test_kat_ruleset(oracle_mod):
    """CollisionRuleset truth table on the oracle's C restatement AND the host class."""
    from mjpl_amd.constraint import CollisionRuleset
    from mjpl_amd.model import ModelBuilder
    table = KAT["ruleset"]
    mb = ModelBuilder()
    mb.add_geom("world", "plane", (1, 1, 0.1))
    parent = "world"
    for b, ngeom in enumerate((2, 1, 2, 1, 1), start=1):  # chain b1..b5, some with two geoms
        mb.add_body(f"b{b}", parent, pos=(0, 0, 0.3))
        mb.add_joint(f"b{b}", f"j{b}", "hinge", axis=(0, 1, 0), range=(-1, 1))
        for _ in range(ngeom):
            mb.add_geom(f"b{b}", "sphere", (0.05,))
        parent = f"b{b}"
    model = mb.compile()
    assert model.geom_bodyid.tolist() == table["geom_bodyid"]
    for case in table["cases"]:
        names = [(model.body(a).name, model.body(b).name) for a, b in case["allowed"]]
        orc = oracle_mod.Oracle(model, names)
        contacts = np.asarray(case["contacts"], dtype=np.int32).reshape(-1, 2)
        assert orc.obeys_ruleset(contacts) is case["obeys"], case
        assert CollisionRuleset(model, names).obeys_ruleset(contacts) is case["obeys"], case
    for bad in (np.zeros((3,)), np.zeros((2, 3)), np.zeros((2, 2, 2))):
        with pytest.raises(ValueError, match="nx2"):
            oracle_mod.Oracle(model).obeys_ruleset(bad)
        with pytest.raises(ValueError, match="nx2"):
            CollisionRuleset(model).obeys_ruleset(bad)


@pytest.mark.parametrize("name,factory", [("franka_p", lambda: scenes.franka_p(obstacles=True)),
                                          ("ur5e_c", scenes.ur5e), ("two_dof_ball", scenes.two_dof_ball)])
def test_fk_against_independent_fixture(oracle_mod, name, factory):
    model = factory()
    orc = oracle_mod.Oracle(model)
    for case in _load(f"fk_{name}.json")["cases"]:
        k = orc.kinematics(case["qpos"])
        for key in ("xpos", "xmat", "geom_xpos", "geom_xmat"):
            np.testing.assert_allclose(k[key], np.asarray(case[key]), rtol=0, atol=1e-12, err_msg=key)


def test_franka_home_is_valid_and_floor_contact_detected(oracle_mod):
    model = scenes.franka_p(obstacles=True)
    orc = oracle_mod.Oracle(model)
    home = model.keyframe("home").qpos
    assert orc.valid_config(home)
    assert len(orc.contacts(home)) == 0
    q = home.copy()
    q[1], q[3] = 1.7, -0.1  # arm swung down and stretched: the hand goes through the floor
    con = orc.contacts(q)
    floor = model.geom("floor").id
    assert any(floor in pair for pair in con.tolist())
    assert not orc.valid_config(q)
    # allowed pairs only silence the named bodies (a6): fingers touching each other
    q2 = home.copy()
    q2[7:] = 0.0
    assert orc.valid_config(q2)  # 3 mm gap when closed


def test_operation_count_of_the_oracle(tmp_path):
    """tools/count_flops.py: the counting build of the oracle (every routine tallies the float64 operations it
    executed) gives the same verdicts as the ordinary build and a count that only depends on the inputs.  One
    configuration of Franka-P + 16 obstacles costs seven sin/cos pairs (its seven hinges away from their
    reference) and a few thousand operations; the committed profiles/flops.json is this tool's output."""
    import subprocess
    import sys
    out = tmp_path / "flops.json"
    tool = os.path.join(ROOT, "tools", "count_flops.py")
    recs = []
    for _ in range(2):
        subprocess.run([sys.executable, tool, "--edges", "512", "--out", str(out)], check=True, capture_output=True)
        recs.append(json.load(open(out)))
    assert recs[0]["ops_per_edge"] == recs[1]["ops_per_edge"]
    r = recs[0]
    assert r["ops_per_config"]["sincos"] == 7.0
    assert 3000 < r["flops_per_config"] < 12000 and r["flops_per_edge"] > 2 * r["flops_per_config"]
    from mjpl_amd import scenes
    import bench
    from oracle import pyoracle
    m = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    qa, qb = bench.make_edges(m, qidx, bench.EDGES_PER_GPU, seed=2)
    v = pyoracle.Oracle(m, planning_qidx=qidx, qpos_base=m.keyframe("home").qpos.copy()).valid_edges(qa[:512], qb[:512], bench.STEP, nthreads=2)
    assert abs(float(np.mean(v)) - r["valid_fraction_of_sample"]) < 1e-12
    committed = json.load(open(os.path.join(ROOT, "profiles", "flops.json")))
    assert abs(committed["flops_per_config"] - r["flops_per_config"]) / r["flops_per_config"] < 0.05


# ---- fixtures made by REAL MuJoCo, whenever somebody has run tools/crosscheck_mujoco.py --write ----------------
def _mujoco_fixtures():
    import glob
    return sorted(glob.glob(os.path.join(GOLDEN, "mujoco_*.json")))


def _crosscheck():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import crosscheck_mujoco
    return crosscheck_mujoco


@pytest.mark.parametrize("path", _mujoco_fixtures() or [None], ids=lambda p: os.path.basename(p)[:-5] if p else "none-committed")
def test_oracle_against_fixtures_written_by_mujoco(oracle_mod, path):
    """Every tests/golden/mujoco_<model>.json present (tools/crosscheck_mujoco.py --write on a machine that has
    the mujoco wheel; none can be made in the build container) pins the oracle to MuJoCo itself: verdicts of the
    seeded configurations, and FK of the first rows to 1e-12.  Capsule-box / box-box disagreements -- the two
    routines restated by another algorithm -- are told apart from any other."""
    if path is None:
        pytest.skip("no MuJoCo-written fixture is committed: the oracle is pinned by the reference's known answers, "
                    "the scipy FK fixtures and the independent narrowphase fixtures only ('parity unpinned' against MuJoCo)")
    rep = _crosscheck().check_fixture(path, oracle_mod)
    assert not rep["stale"], "the model this fixture was made with has changed: regenerate it"
    assert rep["fk_max_abs_err"] < 1e-12, rep
    assert rep["mismatches_other"] == 0, rep
    assert rep["mismatches_capsule_box_or_box_box_only"] == 0, rep  # (reported apart: DESIGN.md section 2's two deviations)


def test_the_fixture_consumer_itself(oracle_mod, tmp_path):
    """The consumer on a fixture written through the same writer from the oracle's own outputs (labelled so: it
    pins nothing) -- it must pass, notice a flipped verdict and a shifted FK number, and name the pair types."""
    cm = _crosscheck()
    name = "pair_capsule_box"
    model = cm.model_by_name(name)
    k, seed = 300, 11
    Q = np.random.default_rng(seed).uniform(model.jnt_range[:, 0], model.jnt_range[:, 1], size=(k, model.nq))
    orc = oracle_mod.Oracle(model)
    valid = orc.valid_configs(Q, nthreads=2).astype(bool)
    contacts = [[[int(a), int(b)] for a, b in orc.contacts(q)] for q in Q]
    fk_o = orc.fk(Q[:16])
    fk = {key: np.asarray(fk_o[key]).reshape(16, -1).tolist() for key in ("xpos", "xquat", "geom_xpos", "geom_xmat")}
    fix = cm.make_fixture(name, model, seed, k, valid, contacts, fk, "SELF-TEST: the oracle's own outputs")
    path = tmp_path / f"mujoco_{name}.json"
    json.dump(fix, open(path, "w"))
    rep = cm.check_fixture(str(path), oracle_mod)
    assert not rep["stale"] and rep["mismatches"] == 0 and rep["fk_max_abs_err"] == 0.0
    # a configuration whose only contact is the capsule-box pair, reported the other way round
    row = next(i for i in range(k) if not valid[i] and all(sorted(c) == [1, 2] for c in contacts[i]))
    bad = dict(fix)
    v = valid.copy()
    v[row] = True
    bad["valid_bits"] = np.packbits(v).tobytes().hex()
    bad["contacts"] = [c if i != row else [] for i, c in enumerate(contacts)]
    bad["fk_first_rows"] = {key: [list(r) for r in rows] for key, rows in fk.items()}
    bad["fk_first_rows"]["xpos"][3][4] += 1e-9
    json.dump(bad, open(path, "w"))
    rep = cm.check_fixture(str(path), oracle_mod)
    assert rep["mismatches"] == 1 and rep["mismatches_capsule_box_or_box_box_only"] == 1 and rep["mismatches_other"] == 0
    assert rep["examples"][0]["pairs"] == ["box-capsule"] and 0.5e-9 < rep["fk_max_abs_err"] < 2e-9
    bad["mjcf_sha256"] = "0" * 64
    json.dump(bad, open(path, "w"))
    assert cm.check_fixture(str(path), oracle_mod)["stale"]
