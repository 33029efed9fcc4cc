"""The C-ABI library loads and exports every symbol include/mjpl_hip.h declares (no compute
without a GPU), the product fails loudly without a device, and the model compiler behaves."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

import mjpl_amd as mjpl
from mjpl_amd import build, engine, scenes
from mjpl_amd.model import GEOM_CAPSULE, JNT_HINGE, JNT_SLIDE, ModelBuilder, load_mjcf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mjpl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mjpl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    build.build_hip()
    lib = ctypes.CDLL(build.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/mjpl_hip.h but not exported"
    assert set(names) == set(engine.ABI), "mjpl_amd.engine.ABI must bind exactly the declared entry points"
    assert engine.load_library().mjpl_version().startswith(b"mjpl_hip")


def test_product_has_no_cpu_fallback():
    if engine.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(engine.MjplError, match="no HIP device"):
        engine.Engine(scenes.two_dof_ball())
    with pytest.raises(engine.MjplError):
        mjpl.CollisionConstraint(scenes.two_dof_ball())
    with pytest.raises(FileNotFoundError, match="no CPU fallback"):
        engine.load_library("/nonexistent/libmjpl_hip.so")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mjpl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in src and "libmjpl_oracle" not in src and "mjpl_oracle.h" not in src, f


def test_model_compile_conventions():
    m = scenes.franka_p(obstacles=True)
    assert (m.nq, m.njnt, m.nbody) == (9, 9, 12)
    assert m.ngeom == 12 + 16
    assert m.body_names[:3] == ["world", "link0", "link1"]
    # link0 and the hand carry no joint: welded to their parents
    assert m.body_weldid[m.body("link0").id] == 0
    assert m.body_weldid[m.body("hand").id] == m.body("link7").id
    # geoms are numbered by body id: every world geom (floor + obstacles) first
    assert (m.geom_bodyid[:17] == 0).all() and m.geom_bodyid[17] == 1
    np.testing.assert_allclose(m.body_quat[m.body("link2").id], [np.sqrt(0.5), -np.sqrt(0.5), 0, 0], atol=1e-15)
    np.testing.assert_allclose(m.jnt_range[m.joint("joint4").id], [-3.0718, -0.0698])
    assert m.jnt_type[m.joint("finger_joint1").id] == JNT_SLIDE and m.jnt_type[0] == JNT_HINGE
    g = m.geom("link1_c").id  # fromto capsule -> pos / half-length / rbound
    np.testing.assert_allclose(m.geom_pos[g], [0, 0, -0.075], atol=1e-15)
    np.testing.assert_allclose(m.geom_size[g][:2], [0.055, 0.055], atol=1e-15)
    np.testing.assert_allclose(m.geom_rbound[g], 0.11, atol=1e-15)
    assert m.geom_type[g] == GEOM_CAPSULE
    np.testing.assert_allclose(m.keyframe("home").qpos[:7], [0, 0, 0, -1.57079, 0, 1.57079, -0.7853])
    with pytest.raises(KeyError):
        m.body("no_such_body")


def test_mjcf_reader_features(tmp_path):
    (tmp_path / "inc.xml").write_text('<mujoco><worldbody><geom name="floor" type="plane" size="1 1 .1"/></worldbody></mujoco>')
    xml = tmp_path / "m.xml"
    xml.write_text("""
    <mujoco>
      <include file="inc.xml"/>
      <default><geom type="capsule" size="0.05 0.1"/><default class="s"><joint type="slide" axis="1 0 0" range="-1 1"/></default></default>
      <worldbody>
        <body name="a" pos="0 0 1" euler="0 0 90">
          <joint name="h" range="-90 90"/>
          <geom name="vis" type="mesh" contype="0" conaffinity="0"/>
          <geom name="ca"/>
          <body name="b" childclass="s" pos="0.5 0 0">
            <joint name="sl"/>
            <geom name="cb" type="sphere" size="0.02"/>
          </body>
        </body>
      </worldbody>
    </mujoco>""")
    m = load_mjcf(str(xml))
    assert m.geom_names == ["floor", "ca", "cb"]          # non-colliding mesh dropped
    assert m.jnt_type.tolist() == [JNT_HINGE, JNT_SLIDE]
    np.testing.assert_allclose(m.jnt_range[0], [-np.pi / 2, np.pi / 2])   # degrees by default
    np.testing.assert_allclose(m.jnt_range[1], [-1, 1])
    np.testing.assert_allclose(m.body_quat[1], [np.sqrt(0.5), 0, 0, np.sqrt(0.5)], atol=1e-15)
    with pytest.raises(ValueError, match="primitive path"):
        load_mjcf('<mujoco><worldbody><body><geom type="mesh"/></body></worldbody></mujoco>')
    with pytest.raises(ValueError, match="1-DoF"):
        mb = ModelBuilder()
        mb.add_body("x")
        mb.add_joint("x", "j", "ball")


def test_mjcf_emitter_round_trips_through_the_parser(tmp_path):
    """mjpl_amd.model.to_mjcf (what tools/crosscheck_mujoco.py feeds to real MuJoCo) must describe
    the same model: parsing the emitted MJCF gives back every table bit for bit."""
    import subprocess
    import sys

    from mjpl_amd import scenes
    from mjpl_amd.model import load_mjcf, to_mjcf
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_models import random_model
    cases = {"franka_p": scenes.franka_p(True), "pads": scenes.franka_p(True, True), "ur5e": scenes.ur5e(),
             "two_dof": scenes.two_dof_ball(), "rand3": random_model(3)[0], "rand1009": random_model(1009)[0]}
    for name, m in cases.items():
        back = load_mjcf(to_mjcf(m, name))
        for f in ("nq", "njnt", "nbody", "ngeom", "nsite"):
            assert getattr(back, f) == getattr(m, f), (name, f)
        for f in ("body_parentid", "body_weldid", "body_jntadr", "body_jntnum", "body_pos", "body_quat", "jnt_type",
                  "jnt_qposadr", "jnt_axis", "jnt_pos", "jnt_range", "qpos0", "geom_type", "geom_bodyid", "geom_contype",
                  "geom_conaffinity", "geom_pos", "geom_quat", "geom_rbound", "geom_margin", "site_bodyid", "site_pos",
                  "site_quat", "key_qpos"):
            np.testing.assert_array_equal(getattr(back, f), getattr(m, f), err_msg=f"{name}.{f}")
        planes = m.geom_type == 0
        np.testing.assert_array_equal(back.geom_size[~planes], m.geom_size[~planes])
        assert back.body_names == m.body_names and back.joint_names == m.joint_names
    # the cross-check tool itself: without mujoco it says so and checks nothing; --emit needs no mujoco
    tool = os.path.join(ROOT, "tools", "crosscheck_mujoco.py")
    r = subprocess.run([sys.executable, tool, "--emit", str(tmp_path), "--models", "franka_p,pairs"], capture_output=True, text=True)
    assert r.returncode == 0 and "MJCF files" in r.stdout, r.stderr
    assert len(list(tmp_path.glob("*.xml"))) == 7
    try:
        import mujoco  # noqa: F401
    except ImportError:
        r = subprocess.run([sys.executable, tool, "--n", "10", "--models", "ur5e"], capture_output=True, text=True)
        assert r.returncode == 0 and "NOTHING WAS CHECKED" in r.stdout


def test_program_dump_and_code_generation_need_no_gpu():
    """mjpl_program_dump compiles a model on the host; mjpl_amd/specialise.py turns the program into
    HIP source.  Hashes are stable, depend on the planning set, and the generated source has one
    stage per moving geom and one bounding cull per enabled pair."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mjpl_amd import scenes, specialise
    from spec_models import spec_models
    seen = set()
    for name, m, allowed, qidx, base in spec_models():
        ip, fp, dp, info = specialise.dump_program(m, allowed, qidx, base)
        ip2, fp2, _, info2 = specialise.dump_program(m, allowed, qidx, base)
        assert info.hash == info2.hash and np.array_equal(ip, ip2) and np.array_equal(fp.view(np.uint32), fp2.view(np.uint32))
        assert info.hash not in seen, name
        seen.add(info.hash)
        assert not info.immediate and info.filter_usable and info.spec_abi == 14
        np.testing.assert_allclose(fp[np.isfinite(fp)], dp[np.isfinite(fp)].astype(np.float32), rtol=2e-5, atol=2e-4)
        src = specialise.generate(ip, fp, dp, info)
        culls = (src.count("MJPL_SPEC_CULLX(") + 2 * src.count("MJPL_SPEC_CULLX2(") + src.count("MJPL_SPEC_CULL(") +
                 2 * src.count("MJPL_SPEC_CULL2(") + src.count("MJPL_SPEC_SLOTCULL(") + src.count("MJPL_SPEC_HIT("))
        assert culls >= 10
        assert "struct Spec" in src and f"{info.hash:016x}" in src
        if info.mbox:
            # moving boxes: the slot file is plain scalars (a literal index everywhere), whole frames go through the box
            # queue, and boxes of one body with one orientation keep their x / y axes once
            assert "struct SpecSlots" in src and "SlotFile<float" not in src
            assert "queue_push<float, true, true>" in src and "cur6b, t6b" in src
            nbox = int(((m.geom_type == 6) & (m.geom_bodyid > 0)).sum())
            stored_axes = src.count(", cur6b); break;")
            assert 0 < stored_axes <= nbox
            if "pad" in name:
                assert stored_axes == 3  # (five pads per finger, three distinct orientations among the stored finger's)
            gsrc = specialise.generate(ip, fp, dp, info, generic=True) if info.scene_ok else None
            assert gsrc is None or ("struct SpecSlots" in gsrc and "MJPL_SCENE_PAIR" in gsrc)
    # sixteen moving boxes in a chain, 28 slots to hold at once: beyond the 24 of the queued kernels -- the immediate
    # interpreter, nothing to specialise
    from mjpl_amd.model import ModelBuilder
    mb = ModelBuilder()
    mb.add_geom("world", "plane", (1, 1, 0.1), pos=(0, 0, -0.4))
    for b in range(16):
        mb.add_body(f"b{b}", f"b{b - 1}" if b else "world", pos=(0.12, 0, 0))
        mb.add_joint(f"b{b}", f"j{b}", "hinge", axis=(0, 0, 1) if b % 2 else (0, 1, 0), range=(-2.5, 2.5))
        mb.add_geom(f"b{b}", "box", (0.04, 0.03, 0.02))
    m = mb.compile()
    _, _, _, info = specialise.dump_program(m)
    assert info.immediate and info.nslots == 28
    with pytest.raises(ValueError, match="immediate"):
        specialise.generate(*specialise.dump_program(m))
    assert specialise.build(m) is None


def test_pose_chain_dump_and_generated_projection_need_no_gpu():
    """mjpl_pose_chain_dump compiles the chain program of a (model, site body) on the host; specialise.generate_pose
    turns it into straight-line code (constants folded, mjpl_amd/fold.py): one block per chain body, the hinges' half-angle
    sines and cosines as ARGUMENTS (the row kernels compute them, one joint per lane of a row's group or one after the
    other), the hash a handle looks its projection up by -- which depends on the chain's constants and not on anything
    else of the model."""
    from mjpl_amd import scenes, specialise
    m = scenes.franka_p(obstacles=True)
    bodies = specialise.pose_site_bodies(m)
    assert bodies == [int(m.site_bodyid[m.site("ee_site").id])]
    pi, pd, h = specialise.dump_pose_chain(m, bodies[0])
    pi2, pd2, h2 = specialise.dump_pose_chain(scenes.franka_p(obstacles=False), bodies[0])
    assert h == h2 and np.array_equal(pi, pi2) and np.array_equal(pd, pd2)  # (obstacles are no part of the chain)
    nb, nj, nq = (int(pi[k]) for k in (specialise.PH_NBODY, specialise.PH_NJOINT, specialise.PH_NQ))
    assert (nj, nq) == (7, 9) and nb >= 8 and len(pd) == 7 * (nb + nj)
    src = specialise.generate_pose(pi, pd, h, 0)
    assert "struct PoseSpec0" in src and f"kNQ = {nq}, kNJ = {nj}" in src and f"{h:016x}" in src
    assert src.count("// chain body") == nb and "sincos" not in src and src.count(": return q[") == nj  # (half_angle: one case per hinge)
    assert "const double (&sn)[7], const double (&cs)[7]" in src and "pd[" not in src and "pi[" not in src
    folded, plain = (specialise.generate_pose(pi, pd, h, 0, fold=f).count(" * ") for f in (True, False))
    assert folded < 0.5 * plain  # (more than half of the statement's products have a factor the constants decide)
    m2 = scenes.franka_p(obstacles=True)
    m2.body_pos[3, 2] += 1e-9  # one constant of the chain, far below anything a test would notice
    assert specialise.dump_pose_chain(m2, bodies[0])[2] != h
    u = scenes.ur5e()
    assert specialise.dump_pose_chain(u, specialise.pose_site_bodies(u)[0])[2] != h
    sec = specialise.generate_pose_section(m)
    for sym in ("mjpl_spec_pose_count", "mjpl_spec_pose_hash", "mjpl_spec_launch_pose_apply", "mjpl_spec_launch_gen_project"):
        assert sym in sec


def test_program_hash_covers_the_float64_constants_and_the_shared_headers():
    """A specialised library carries its program's float64 constants as literals (the exact pair
    re-check, ExactSpec::fk_pair) and is built from the same headers as libmjpl_hip.so.  Two programs
    whose constants differ by less than a binary32 ulp share their float32 image -- they must not
    share a hash, or one would be re-checked against the other's constants; and the digest of the
    shared headers is part of both the library (flag MJPL_SRC_STAMP) and the hash."""
    from mjpl_amd import build, scenes, specialise
    m = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    ip, fp, dp, info = specialise.dump_program(m, (), qidx, base)
    base2 = base.copy()
    base2[7] += 1e-10  # a finger opening: a non-planning joint's constant, far below a binary32 ulp of 0.04
    ip2, fp2, dp2, info2 = specialise.dump_program(m, (), qidx, base2)
    assert np.array_equal(ip, ip2) and np.array_equal(fp.view(np.uint32), fp2.view(np.uint32))
    assert (dp != dp2).sum() == 1
    assert info.hash != info2.hash
    # the stamp: a digest of the three shared headers, passed to both compilations
    stamp = build.src_stamp()
    assert stamp != 0 and f"-DMJPL_SRC_STAMP=0x{stamp:016x}ull" in build.hipcc_flags()
    src = specialise.translation_unit("", "", info.hash, info)
    assert "mjpl_spec_src_stamp" in src and "MJPL_SRC_STAMP" in src


def test_robot_hash_and_scene_generic_generation():
    """The hash a scene-generic library is named by covers the robot only: the same for Franka-P alone, among the
    16 committed obstacles and among seeded random ones -- and another for another planning set, base
    configuration or tolerance.  Generated generic code carries no static geom: no literal static cull, a loop
    over the scene table instead, geoms numbered from the first moving one in the exact pair re-check."""
    from mjpl_amd import scenes, specialise
    arm = None
    hashes, programs = set(), {}
    for m in (scenes.franka_p(False), scenes.franka_p(True), scenes.franka_p_scene(5, 5, 4), scenes.franka_p_scene(2, 1, 5, n_capsules=3)):
        arm = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
        base = m.keyframe("home").qpos.copy()
        ip, fp, dp, info = specialise.dump_program(m, (), arm, base)
        assert info.scene_ok == 1 and info.scene_rows == 32
        hashes.add(info.robot_hash)
        programs[info.hash] = (ip, fp, dp, info)
    assert len(hashes) == 1 and len(programs) == 4
    m = scenes.franka_p(True)
    base = m.keyframe("home").qpos.copy()
    other = base.copy()
    other[7] = 0.02
    for kw in (dict(qidx=np.arange(m.nq, dtype=np.int32), qpos_base=base), dict(qidx=arm, qpos_base=other),
               dict(qidx=arm, qpos_base=base, filter_tol=2e-4)):
        assert specialise.dump_program(m, (), **kw)[3].robot_hash not in hashes
    assert specialise.dump_program(scenes.franka_p_scene(16, 16, 9), (), arm, base)[3].scene_ok == 0  # 34 static geoms
    ip, fp, dp, info = next(iter(programs.values()))
    src = specialise.generate(ip, fp, dp, info, generic=True)
    assert "MJPL_SPEC_CULLX" not in src.split("struct Spec")[1] and "MJPL_SCENE_PAIR(ra + 0, cc, cc, 0, 1)" in src and "acc_a, acc_b, 30, 31)" in src
    assert src.count("MJPL_SPEC_SLOTCULL(") == 33 + 1  # Franka-P's 33 self pairs (and the macro itself), in either kind of library
    srcs = {specialise.generate(*p, generic=True) for p in programs.values()}
    assert len({s.split("program hash")[1].split("\n", 1)[1] for s in srcs}) == 1, "generic code must not depend on the scene"
    ex = specialise.generate_exact(ip, dp, info, generic=True)
    assert "kRelative = 1" in ex and "ga == 0 || gb == 0" in ex


def test_options_table_is_enumerable_without_a_device_and_documented():
    """include/mjpl_hip.h: mjpl_option_count / mjpl_option_name need no engine (and no GPU); every name the table
    holds is in tools/README.md's options section, and the library's sources read the environment in four places only
    (the MJPL_DEBUG gate and its loop, the debug file of -DMJPL_FUSED_DEBUG builds, one A/B print)."""
    import re
    from mjpl_amd import engine
    names = engine.option_names()
    assert len(names) == len(set(names)) >= 45
    for must in ("filter", "fused", "fused_cert_min_edges", "pose_spec", "rows_g", "rrt_early_nn", "nn_cells", "kernel_timer"):
        assert must in names
    writable = engine.option_names(writable_only=True)
    assert "nn_last_cells" in names and "nn_last_cells" not in writable
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tools", "README.md")) as f:
        doc = f.read()
    missing = [n for n in names if f"`{n}`" not in doc]
    assert not missing, missing
    sites = 0
    for fn in os.listdir(os.path.join(root, "mjpl_amd", "csrc")):
        if fn.endswith((".h", ".hip")):
            with open(os.path.join(root, "mjpl_amd", "csrc", fn)) as f:
                sites += len(re.findall(r"\bgetenv\(", f.read()))
    assert sites <= 5, sites
