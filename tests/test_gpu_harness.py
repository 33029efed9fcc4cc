"""examples/benchmark.py -- the planning-time harness in the shape of the reference's
(/root/reference/examples/benchmark.py:28-48, 83-91; run by its CI, .github/workflows/ci.yml:44) -- under
test: 15 attempts of plan_to_pose, epsilon 0.05, seed 42, goal bias 0.1; every attempt must succeed and
every path must pass the CPU oracle (collision of every waypoint and of the interval waypoints between
them where an interval check was asked for), stay inside the joint limits and step no farther than
epsilon."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def _checker(oracle_mod, obstacles, interval):
    from mjpl_amd import scenes
    model = scenes.franka_p(obstacles=obstacles)
    orc = oracle_mod.Oracle(model)

    def check(path):
        P = np.stack(path)
        ok = orc.valid_configs(P, nthreads=4).all()
        ok = ok and np.all((P >= model.jnt_range[:, 0]) & (P <= model.jnt_range[:, 1]))
        ok = ok and np.linalg.norm(np.diff(P, axis=0), axis=1).max() <= 0.05 + 1e-9
        if interval:
            ok = ok and all(orc.valid_collision_interval(a, b, interval) for a, b in zip(P[:-1], P[1:]))
        return bool(ok)

    return check


@pytest.mark.parametrize("planner,obstacles,interval", [("device", True, 0.0), ("device", False, 0.0), ("device", True, 0.01),
                                                         ("rrt", True, 0.0)])
def test_fifteen_attempts_in_the_reference_shape(oracle_mod, planner, obstacles, interval):
    import benchmark
    res = benchmark.run(planner=planner, attempts=15, obstacles=obstacles, interval=interval, seed=42, quiet=True,
                        check_path=_checker(oracle_mod, obstacles, interval))
    assert res["successes"] == 15 and len(res["planning_times"]) == 15
    assert res["paths_valid"]
    q_init = res["q_init"]
    for path in res["paths"]:
        np.testing.assert_array_equal(path[0], q_init)
        assert len(path) >= 2
    assert np.median(res["planning_times"]) < 2.0  # (measured: 0.004 - 0.05 s; the reference allows 10 s per plan)


def test_attempts_with_their_own_seeds(oracle_mod):
    import benchmark
    res = benchmark.run(planner="device", attempts=8, obstacles=True, seed=42, vary_seed=True, quiet=True,
                        check_path=_checker(oracle_mod, True, 0.0))
    assert res["successes"] == 8 and res["paths_valid"]
