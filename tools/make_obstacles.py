"""Generate mjpl_amd/models/obstacles16.json: the 16 seeded world-fixed obstacles of BASELINE
configs 3-5: 8 boxes + 8 spheres, np.random.default_rng(0), centres
uniform in x,y in [-0.8,0.8], z in [0.1,1.0]; box half-extents U[0.02,0.15] with a random
orientation; sphere radii U[0.03,0.12]; any obstacle touching Franka-P at its home keyframe
(or reaching below the floor) is rejected and redrawn.  Uses the CPU oracle for the
rejection test, so it is a build-time tool, not product code.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mjpl_amd import scenes  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    base = scenes.franka_p_builder()
    home = base.compile().keyframe("home").qpos
    obstacles = []
    kinds = ["box"] * 8 + ["sphere"] * 8
    for k, kind in enumerate(kinds):
        while True:
            pos = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0.1, 1.0)])
            if kind == "box":
                size = rng.uniform(0.02, 0.15, size=3)
                quat = rng.normal(size=4)
                quat /= np.linalg.norm(quat)
                reach = float(np.linalg.norm(size))
            else:
                size = np.array([rng.uniform(0.03, 0.12)])
                quat = np.array([1.0, 0, 0, 0])
                reach = float(size[0])
            if pos[2] - reach <= 0.0:
                continue
            cand = dict(name=f"obstacle_{k}", type=kind, size=[float(x) for x in size],
                        pos=[float(x) for x in pos], quat=[float(x) for x in quat])
            model = scenes.franka_p_builder([cand]).compile()
            if pyoracle.Oracle(model).valid_config(home):
                obstacles.append(cand)
                break
    out = os.path.join(ROOT, "mjpl_amd", "models", "obstacles16.json")
    with open(out, "w") as f:
        json.dump(dict(seed=0, generator="tools/make_obstacles.py", obstacles=obstacles), f, indent=1)
    model = scenes.franka_p(obstacles=True)
    orc = pyoracle.Oracle(model)
    assert orc.valid_config(home)
    rng = np.random.default_rng(1)
    Q = rng.uniform(model.jnt_range[:, 0], model.jnt_range[:, 1], size=(20000, model.nq))
    Q[:, 7:] = 0.04
    print("wrote", out, "| valid fraction of uniform configs:", orc.valid_configs(Q, nthreads=8).mean())


if __name__ == "__main__":
    main()
