"""The (model, planning set) combinations whose specialised filter kernels are prebuilt by
``__graft_entry__.build()`` (hipcc, no GPU) and exercised by tests/test_gpu_spec.py on the GPU box,
where nothing may be compiled from a process that has touched the GPU."""
import numpy as np

from mjpl_amd import scenes


def spec_models():
    out = []
    m = scenes.franka_p(obstacles=True)
    arm = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    out.append(("franka_p+16obs, arm planned (bench.py, BASELINE configs[2])", m, (), arm, m.keyframe("home").qpos.copy()))
    out.append(("franka_p+16obs, all nine joints planned", m, (), np.arange(m.nq, dtype=np.int32), np.asarray(m.qpos0, float).copy()))
    out.append(("franka_p+16obs, fingers allowed to touch", m, (("left_finger", "right_finger"),), arm,
                m.keyframe("home").qpos.copy()))
    m0 = scenes.franka_p(obstacles=False)
    out.append(("franka_p self-collision (BASELINE configs[1])", m0, (), np.arange(m0.nq, dtype=np.int32),
                np.asarray(m0.qpos0, float).copy()))
    # ... with the reference Panda's ten finger-pad boxes (panda.xml:134-241): moving boxes -- whole frames in the slot file
    # and in the box queue's records, kernels built for two waves per SIMD (the five pads of a finger share frames)
    mp = scenes.franka_p(True, True)
    out.append(("franka_p+16obs+10 pad boxes (moving boxes)", mp, (), scenes.planning_index(mp, scenes.FRANKA_ARM_JOINTS),
                mp.keyframe("home").qpos.copy()))
    u = scenes.ur5e()
    out.append(("ur5e_c", u, (), np.arange(u.nq, dtype=np.int32), np.asarray(u.qpos0, float).copy()))
    # the reference's KAT model (test/models/two_dof_ball.xml): two SLIDE joints under a site -- the slide branch of the
    # generated PoseConstraint chain (tests/test_gpu_pose.py: test_translation_limit_kat_on_gpu)
    tb = scenes.two_dof_ball()
    out.append(("two_dof_ball", tb, (), np.arange(tb.nq, dtype=np.int32), np.asarray(tb.qpos0, float).copy()))
    from test_gpu_models import random_model
    for seed in (1002, 1005):  # seeded random trees with slides, off-centre hinges, branching, static boxes
        rm, allowed = random_model(seed, moving_boxes=False)
        out.append((f"random_model({seed})", rm, tuple(allowed), np.arange(rm.nq, dtype=np.int32), np.asarray(rm.qpos0, float).copy()))
    for seed in (1003, 1004):  # ... and with boxes of any orientation on the moving bodies (seeds whose configurations are neither all free nor all in contact)
        rm, allowed = random_model(seed, moving_boxes=True)
        out.append((f"random_model({seed}, moving boxes)", rm, tuple(allowed), np.arange(rm.nq, dtype=np.int32),
                    np.asarray(rm.qpos0, float).copy()))
    return out


def generic_scenes():
    """Scenes of ONE robot (Franka-P, arm joints planned from the home keyframe) for the scene-generic library
    __graft_entry__.build() compiles from the first of them: (name, model, expected mjpl_spec_loaded --
    1 where the scene's own literal library exists and wins, 2 where the robot's generic one serves)."""
    return [("franka_p + the 16 committed obstacles (has its own library)", scenes.franka_p(obstacles=True), 1),
            ("franka_p alone (floor + base)", scenes.franka_p(obstacles=False), 2),
            ("franka_p + 6 boxes + 4 spheres, seed 1", scenes.franka_p_scene(6, 4, 1), 2),
            ("franka_p + 3 boxes + 3 spheres + 5 capsules, seed 2", scenes.franka_p_scene(3, 3, 2, n_capsules=5), 2),
            ("franka_p + 14 boxes + 12 spheres, seed 3 (30 static geoms)", scenes.franka_p_scene(14, 12, 3), 2),
            ("franka_p + a wall (two planes) + 4 boxes + 3 spheres, seed 5", _with_wall(scenes.random_obstacles(4, 3, 5)), 2),
            ("franka_p + a wall (two planes) + 3 boxes, seed 6 (odd number of bounded rows)", _with_wall(scenes.random_obstacles(3, 0, 6)), 2),
            # ... and the robot WITH the Panda's ten finger-pad boxes (moving boxes): a second generic library
            ("franka_p with pads + the 16 committed obstacles (has its own library)", scenes.franka_p(True, True), 1),
            ("franka_p with pads alone (floor + base)", scenes.franka_p(False, True), 2),
            ("franka_p with pads + 6 boxes + 4 spheres, seed 1", scenes.franka_p_builder(scenes.random_obstacles(6, 4, 1), pads=True).compile(), 2),
            ("franka_p with pads + 2 boxes + 3 spheres + 4 capsules, seed 7",
             scenes.franka_p_builder(scenes.random_obstacles(2, 3, 7, n_capsules=4), pads=True).compile(), 2)]


def _with_wall(obstacles):
    """... and a second plane: a wall behind the robot at x = -0.55, its normal along +x."""
    wall = {"type": "plane", "size": (0.0, 0.0, 0.05), "pos": (-0.55, 0.0, 0.0), "quat": (np.sqrt(0.5), 0.0, np.sqrt(0.5), 0.0),
            "name": "wall"}
    return scenes.franka_p_builder([wall] + list(obstacles)).compile()


def generic_robot():
    """(model, allowed, planning indices, base) the generic library is generated from."""
    m = scenes.franka_p(obstacles=True)
    return m, (), scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS), m.keyframe("home").qpos.copy()


def generic_robots():
    """Every robot __graft_entry__.build() makes a scene-generic library for: Franka-P, and Franka-P with the finger pads."""
    mp = scenes.franka_p(True, True)
    return [generic_robot(), (mp, (), scenes.planning_index(mp, scenes.FRANKA_ARM_JOINTS), mp.keyframe("home").qpos.copy())]
