"""Scene-generic specialised kernels: ONE library per robot -- its forward kinematics, geom
poses and self-pair culls as literals, every static partner a row of the engine's scene table -- serves the
robot in any scene of up to 32 static geoms: obstacles change with mjpl_create alone, no compiler.  Three
obstacle sets (and the bare robot) on engines that must report the generic library (spec_loaded() == 2),
verdicts bit-identical to the CPU oracle and to the interpreting kernels; the scene's own literal library, where
one exists, still wins the lookup.  The library is built by __graft_entry__.build() (hipcc, no GPU)."""
import numpy as np
import pytest

from mjpl_amd import engine as eng_mod
from mjpl_amd import scenes
from spec_models import generic_scenes

pytestmark = pytest.mark.gpu


def _edges(m, qidx, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = m.jnt_range[qidx, 0], m.jnt_range[qidx, 1]
    qa = rng.uniform(lo, hi, size=(n, len(qidx)))
    d = rng.normal(size=qa.shape)
    qb = np.clip(qa + rng.choice([0.05, 0.05, 0.25], size=(n, 1)) * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)
    return qa, qb


@pytest.mark.parametrize("which", range(len(generic_scenes())))
def test_one_robot_library_serves_every_scene(oracle_mod, which):
    name, m, expect = generic_scenes()[which]
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    e = eng_mod.Engine(m)
    e.set_planning(qidx, base)
    assert e.lib.mjpl_spec_loaded(e.h) == expect, name  # 2: the robot's generic library; 1: this scene's own
    qa, qb = _edges(m, qidx, 30000, 40 + which)
    orc = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=base)
    want, wfb, _ = orc.valid_edges(qa, qb, 0.01, nthreads=8, info=True)
    got, gfb = e.check_edges(qa, qb, 0.01, first_bad=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(gfb, wfb)
    assert 0.02 < want.mean() < 0.98
    Q = np.concatenate([qa, qb])
    np.testing.assert_array_equal(e.check_configs(Q), orc.valid_configs(Q, nthreads=8))
    und = e.last_undecided()
    e.set_spec(False)  # the interpreting kernels on the same engine
    assert e.lib.mjpl_spec_loaded(e.h) == 0
    np.testing.assert_array_equal(e.check_edges(qa, qb, 0.01), want)
    e.set_spec(True)
    assert e.lib.mjpl_spec_loaded(e.h) == expect
    # both launch layouts
    for env in ({"MJPL_FUSED": "0"}, {"MJPL_FUSED": "0", "MJPL_PERSIST": "0"}, {"MJPL_TAIL": "0"}, {"MJPL_FUSED_SINGLE": "0"}):
        import os
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            e2 = eng_mod.Engine(m)
            e2.set_planning(qidx, base)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
        assert e2.lib.mjpl_spec_loaded(e2.h) == expect
        np.testing.assert_array_equal(e2.check_edges(qa, qb, 0.01), want)
        e2.close()
    assert und >= 0
    e.close()


def test_scene_beyond_the_table_falls_back_to_the_interpreter(oracle_mod):
    """33 static geoms do not fit the 32 rows a generic library reads: the engine runs the interpreter."""
    m = scenes.franka_p_scene(n_boxes=16, n_spheres=16, seed=9)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    e = eng_mod.Engine(m)
    e.set_planning(qidx, base)
    assert e.info()["nstatic_geoms"] > 32 and e.lib.mjpl_spec_loaded(e.h) == 0
    qa, qb = _edges(m, qidx, 8000, 3)
    np.testing.assert_array_equal(e.check_edges(qa, qb, 0.01),
                                  oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=base).valid_edges(qa, qb, 0.01, nthreads=8))
    e.close()
