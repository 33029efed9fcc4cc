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


def ur5e() -> Model:
    return parse_mjcf(os.path.join(_MODELS, "ur5e_c.xml")).compile()


def planning_index(model: Model, joints: list[str]) -> np.ndarray:
    """qpos indices of 1-DoF ``joints`` (the reference's utils.qpos_idx, src/mjpl/utils.py:22-38)."""
    return np.array([model.jnt_qposadr[model.joint(j).id] for j in joints], dtype=np.int32)
