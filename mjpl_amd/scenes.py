"""Scenes used by the tests and the benchmark.

* :func:`one_dof_ball` / :func:`two_dof_ball` -- the analytic scenes the reference's tests use
  (test/models/one_dof_ball.xml:4-12, test/models/two_dof_ball.xml:4-14), rebuilt through
  :class:`~mjpl_amd.model.ModelBuilder` from their physical content.
* :func:`franka_p` -- Franka-P (mjpl_amd/models/franka_p.xml), optionally with the 16 seeded
  box/sphere obstacles of BASELINE configs 3-5 (mjpl_amd/models/obstacles16.json, written by
  tools/make_obstacles.py).
* :func:`ur5e` -- UR5e-C (mjpl_amd/models/ur5e_c.xml) for BASELINE config 1.
"""
from __future__ import annotations

import json
import os

import numpy as np

from .model import Model, ModelBuilder, parse_mjcf

_MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")

FRANKA_ARM_JOINTS = [f"joint{i}" for i in range(1, 8)]  # examples/benchmark.py:29-37


def one_dof_ball() -> Model:
    """Sphere r=0.01 sliding along x at z=1; wall box x in [0.85, 0.95]; floor plane z=0."""
    mb = ModelBuilder()
    mb.add_geom("world", "plane", (2, 2, 0.1))
    mb.add_body("ball", pos=(0, 0, 1))
    mb.add_joint("ball", "ball_slide_x", "slide", axis=(1, 0, 0), range=(-2, 2))
    mb.add_geom("ball", "sphere", (0.01,))
    mb.add_geom("world", "box", (0.05, 0.5, 0.5), pos=(0.9, 0, 1), name="wall_obstacle")
    return mb.compile()


def two_dof_ball() -> Model:
    """Sphere r=0.1 translating in the xy-plane at z=1; wall box x in [0.5,0.7], |y|<=0.5."""
    mb = ModelBuilder()
    mb.add_geom("world", "plane", (2, 2, 0.1))
    mb.add_body("ball", pos=(0, 0, 1))
    mb.add_joint("ball", "ball_slide_x", "slide", axis=(1, 0, 0), range=(-2, 2))
    mb.add_joint("ball", "ball_slide_y", "slide", axis=(0, 1, 0), range=(-2, 2))
    mb.add_geom("ball", "sphere", (0.1,))
    mb.add_site("ball", "ball_site")
    mb.add_geom("world", "box", (0.1, 0.5, 0.5), pos=(0.6, 0, 1), name="wall_obstacle")
    return mb.compile()


def load_obstacles(name: str = "obstacles16.json") -> list[dict]:
    with open(os.path.join(_MODELS, name)) as f:
        return json.load(f)["obstacles"]


# the ten fingertip pad boxes of the reference Panda, five per finger body:
# (half-sizes, position in the finger frame), examples/models/franka_emika_panda/panda.xml:20-33,225-241
FINGER_PADS = (((0.0085, 0.004, 0.0085), (0.0, 0.0055, 0.0445)),
               ((0.003, 0.002, 0.003), (0.0055, 0.002, 0.05)),
               ((0.003, 0.002, 0.003), (-0.0055, 0.002, 0.05)),
               ((0.003, 0.002, 0.0035), (0.0055, 0.002, 0.0395)),
               ((0.003, 0.002, 0.0035), (-0.0055, 0.002, 0.0395)))


def franka_p_builder(obstacles=False, pads=False) -> ModelBuilder:
    mb = parse_mjcf(os.path.join(_MODELS, "franka_p.xml"))
    if pads:
        for finger in ("left_finger", "right_finger"):
            for k, (size, pos) in enumerate(FINGER_PADS):
                mb.add_geom(finger, "box", size, pos=pos, name=f"{finger}_pad{k + 1}")
    if obstacles:
        obs = load_obstacles() if obstacles is True else obstacles
        for k, o in enumerate(obs):
            mb.add_geom("world", o["type"], o["size"], pos=o["pos"], quat=o.get("quat", (1, 0, 0, 0)),
                        name=o.get("name", f"obstacle_{k}"))
    return mb


def franka_p(obstacles=False, pads=False) -> Model:
    """Franka-P; ``obstacles=True`` adds the 16 committed box/sphere obstacles, ``pads=True`` the
    ten fingertip pad boxes the reference Panda carries on its two finger bodies (moving boxes)."""
    return franka_p_builder(obstacles, pads).compile()


def random_obstacles(n_boxes: int, n_spheres: int, seed: int, n_capsules: int = 0) -> list[dict]:
    """Seeded obstacles in the spirit of examples/models/franka_emika_panda/scene_with_obstacles.xml:23-30: centres
    uniform in x, y in [-0.8, 0.8], z in [0.1, 1.0], kept 0.3 m away from the robot's column; box half-extents
    U[0.02, 0.15], sphere radii U[0.03, 0.12], capsules r U[0.02, 0.06] x half-length U[0.05, 0.2] at random attitudes."""
    rng = np.random.default_rng(seed)
    out = []
    kinds = ["box"] * n_boxes + ["sphere"] * n_spheres + ["capsule"] * n_capsules
    for k, kind in enumerate(kinds):
        while True:
            pos = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0.1, 1.0)])
            if np.hypot(pos[0], pos[1]) >= 0.3:
                break
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        if kind == "box":
            size = tuple(float(x) for x in rng.uniform(0.02, 0.15, 3))
        elif kind == "sphere":
            size, q = (float(rng.uniform(0.03, 0.12)),), np.array([1.0, 0, 0, 0])
        else:
            size = (float(rng.uniform(0.02, 0.06)), float(rng.uniform(0.05, 0.2)))
        out.append({"type": kind, "size": size, "pos": tuple(float(x) for x in pos), "quat": tuple(float(x) for x in q),
                    "name": f"obstacle_{seed}_{k}"})
    return out


def franka_p_scene(n_boxes: int, n_spheres: int, seed: int, n_capsules: int = 0) -> Model:
    """Franka-P among seeded random obstacles (another scene per seed: what a scene-generic library is for)."""
    return franka_p_builder(random_obstacles(n_boxes, n_spheres, seed, n_capsules)).compile()


def ur5e() -> Model:
    return parse_mjcf(os.path.join(_MODELS, "ur5e_c.xml")).compile()


def planning_index(model: Model, joints: list[str]) -> np.ndarray:
    """qpos indices of 1-DoF ``joints`` (the reference's utils.qpos_idx, src/mjpl/utils.py:22-38)."""
    return np.array([model.jnt_qposadr[model.joint(j).id] for j in joints], dtype=np.int32)
