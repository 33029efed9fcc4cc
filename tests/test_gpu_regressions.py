"""GPU regression tests for defects found by review: buffers sized for an
earlier planning set, the edge-level undecided list overflowing when one edge contributes several
undecided waypoint items, the sticky status of the device-pointer edge entry point, handle
lifetimes.  Every verdict is compared with the CPU oracle."""
import os

import numpy as np
import pytest

from mjpl_amd import engine as eng_mod
from mjpl_amd import scenes
from mjpl_amd.model import ModelBuilder

from helpers import random_edges

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


def test_replanning_with_more_columns_keeps_item_rows_apart(oracle_mod):
    """set_planning(5 columns) -> check_edges -> set_planning(7 columns) -> check_edges with the
    same E: the waypoint item buffer allocated for the first call must hold the wider rows."""
    m = scenes.franka_p(obstacles=True)
    base = m.keyframe("home").qpos.copy()
    e = eng_mod.Engine(m)
    E = 20000
    for joints in (scenes.FRANKA_ARM_JOINTS[:5], scenes.FRANKA_ARM_JOINTS, scenes.FRANKA_ARM_JOINTS[:3],
                   scenes.FRANKA_ARM_JOINTS + ["finger_joint1", "finger_joint2"]):
        qidx = scenes.planning_index(m, joints)
        e.set_planning(qidx, base)
        orc = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=base)
        qa, qb = random_edges(m, qidx, E, seed=len(joints))
        want, wfb, _ = orc.valid_edges(qa, qb, 0.01, nthreads=8, info=True)
        got, gfb = e.check_edges(qa, qb, 0.01, first_bad=True)
        np.testing.assert_array_equal(got, want, err_msg=str(joints))
        np.testing.assert_array_equal(gfb, wfb, err_msg=str(joints))
        assert e.last_items() > E  # the lane-per-waypoint pass ran
    e.close()


def test_undecided_overflow_lists_every_edge_once(oracle_mod):
    """A tolerance band of half a metre makes nearly every culled-in pair undecided; with a
    64-entry hand-over buffer the items fall back to the edge-level list, several per edge.  The
    list holds E entries and the exact re-run has one lane per entry: duplicates would overflow
    it and leave edges at the provisional `valid` of the endpoint pass."""
    m = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    orc = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=base)
    qa, qb = random_edges(m, qidx, 6000, seed=11)
    want, wfb, _ = orc.valid_edges(qa, qb, 0.01, nthreads=8, info=True)
    with _Env(MJPL_UC_CAP="64"):
        e = eng_mod.Engine(m)
    e.set_planning(qidx, base)
    e.set_filter(True, 0.5)
    got, gfb = e.check_edges(qa, qb, 0.01, first_bad=True)
    assert e.last_undecided() > 1000  # the overflow path really ran
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(gfb, wfb)
    # configurations through the same tiny buffer
    Q = np.concatenate([qa, qb])
    np.testing.assert_array_equal(e.check_configs(Q), orc.valid_configs(Q, nthreads=8))
    e.close()


def _long_slide() -> "scenes.Model":
    """A ball on a 200 m rail with walls inside and outside the float32 filter's range."""
    mb = ModelBuilder()
    mb.add_geom("world", "plane", (0, 0, 0.1))
    mb.add_body("ball", pos=(0, 0, 1))
    mb.add_joint("ball", "rail", "slide", axis=(1, 0, 0), range=(-100, 100))
    mb.add_geom("ball", "sphere", (0.01,))
    for k, x in enumerate((0.9, 63.5, 64.5, 70.0)):
        mb.add_geom("world", "box", (0.05, 0.5, 0.5), pos=(x, 0, 1), name=f"wall{k}")
    return mb.compile()


def test_edges_straddling_the_filter_range(oracle_mod):
    """Edges that start beyond the range the float32 filter decides and end inside it: the
    endpoint is decided by the filter, every far interior waypoint is an undecided item of the
    same edge.  Verdict and first_bad must be the oracle's, in every interior-pass mode."""
    m = _long_slide()
    orc = oracle_mod.Oracle(m)
    rng = np.random.default_rng(3)
    n = 3000
    far = rng.uniform(62.0, 72.0, size=(n, 1))
    near = rng.uniform(61.0, 66.0, size=(n, 1))
    qa, qb = np.concatenate([far, near]), np.concatenate([near, far])
    want, wfb, _ = orc.valid_edges(qa, qb, 0.05, nthreads=8, info=True)
    assert 0.05 < want.mean() < 0.95
    for env in ({}, {"MJPL_EXPAND": "0"}, {"MJPL_TWO_PASS": "0"}, {"MJPL_UC_CAP": "16"}, {"MJPL_FUSED": "0"},
                {"MJPL_FUSED": "0", "MJPL_UC_CAP": "16"}, {"MJPL_FUSED_SINGLE": "0"}, {"MJPL_FUSED_POLICY": "1", "MJPL_UC_CAP": "16"}):
        with _Env(**env):
            e = eng_mod.Engine(m)
        got, gfb = e.check_edges(qa, qb, 0.05, first_bad=True)
        np.testing.assert_array_equal(got, want, err_msg=str(env))
        np.testing.assert_array_equal(gfb, wfb, err_msg=str(env))
        e.close()


def test_device_path_status_is_sticky_until_taken():
    m = scenes.one_dof_ball()
    e = eng_mod.Engine(m)
    qa = np.array([[0.0], [np.nan], [0.1]])
    qb = np.array([[0.2], [0.3], [0.3]])
    da, db = e.alloc(qa.nbytes).upload(qa), e.alloc(qb.nbytes).upload(qb)
    dv, dfb = e.alloc(3), e.alloc(12)
    assert e.take_status() == 0
    e.check_edges_dev(da.ptr, db.ptr, 3, 0.01, eng_mod.AOS, dv.ptr, dfb.ptr)
    assert dv.download(np.uint8, 3).tolist() == [1, 0, 1]
    assert dfb.download(np.int32, 3).tolist() == [-1, -2, -1]
    assert e.take_status() == -7  # MJPL_E_NONFINITE
    assert e.take_status() == 0   # taken
    e.close()


def test_engine_close_destroys_its_pose_handles():
    import mjpl_amd as mjpl
    m = scenes.franka_p()
    q = m.keyframe("home").qpos.copy()
    e = eng_mod.Engine(m)
    frame = mjpl.site_pose(m, q, "ee_site", engine=e)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), engine=e)
    assert pc.valid_config(q)
    e.close()          # engine first ...
    assert pc._proj.h is None
    pc._proj.close()   # ... then the projector: nothing left to free, no crash
    # site_pose without an engine builds and closes its own
    assert np.allclose(mjpl.site_pose(m, q, "ee_site").translation(), frame.translation())
    assert mjpl.CollisionConstraint.__doc__.startswith("Batched collision validation")


@pytest.mark.parametrize("immediate", [False, True])
@pytest.mark.parametrize("allowed", [(), (("left_finger", "right_finger"),)])
def test_franka_with_the_ten_finger_pad_boxes(oracle_mod, allowed, immediate):
    """The reference Panda carries ten pad boxes on its finger bodies (panda.xml:20-33,225-241):
    moving boxes, 17 geoms held in the slot file at once.  Such models run the queued interpreter's
    24-slot build, box pairs through its box queue with full frames (and, forced here, the
    immediate interpreter that models with more than 24 stored geoms get), with the
    lane-per-waypoint interior pass; verdicts
    and first-bad indices must be the oracle's, filter on and off, fingers moving or not.
    Clipping the random edges to the joint ranges puts both fingers at exactly 0 in some rows, where
    opposite pads touch with a gap of +-1e-17: there the verdict hangs on the last bit of sin/cos, so
    the oracle runs with its bit-reproducible trig (see test_gpu_filter_adversarial.py)."""
    with oracle_mod.portable_trig(), _Env(MJPL_FORCE_IMMEDIATE="1" if immediate else "0"):
        _finger_pads(oracle_mod, allowed, immediate)


def _finger_pads(oracle_mod, allowed, immediate):
    m = scenes.franka_p(obstacles=True, pads=True)
    base = m.keyframe("home").qpos.copy()
    e = eng_mod.Engine(m, allowed)
    info = e.info()
    assert info["nmoving_geoms"] == 20 and info["filter_enabled"] == 1
    assert info["filter_interpreter"] == (2 if immediate else 1)
    assert (info["nslots"] > 16) == (not allowed)  # pad-against-pad pairs keep the left pads in the slot file
    rng = np.random.default_rng(3)
    for joints in (scenes.FRANKA_ARM_JOINTS, scenes.FRANKA_ARM_JOINTS + ["finger_joint1", "finger_joint2"]):
        qidx = scenes.planning_index(m, joints)
        b = base.copy()
        b[7:] = rng.uniform(0.0, 0.04, size=2)
        e.set_planning(qidx, b)
        orc = oracle_mod.Oracle(m, allowed, planning_qidx=qidx, qpos_base=b)
        qa, qb = random_edges(m, qidx, 30000, seed=len(joints))
        want, wfb, _ = orc.valid_edges(qa, qb, 0.01, nthreads=8, info=True)
        assert 0.05 < want.mean() < 0.95
        for filt in (True, False):
            e.set_filter(filt, 1e-4) if filt else e.set_filter(False)
            got, gfb = e.check_edges(qa, qb, 0.01, first_bad=True)
            np.testing.assert_array_equal(got, want, err_msg=f"{joints[-1]} filter={filt}")
            np.testing.assert_array_equal(gfb, wfb, err_msg=f"{joints[-1]} filter={filt}")
            if filt:
                assert e.last_items() > len(qa)  # one lane per interior waypoint, also for this model
        Q = np.concatenate([qa, qb])
        np.testing.assert_array_equal(e.check_configs(Q), orc.valid_configs(Q, nthreads=8))
    e.close()


def test_item_count_is_the_reference_walk(oracle_mod):
    """The endpoint kernel counts the interior waypoints of an edge by walking the reference's
    recurrence (planning/utils.py:182-185) -- with its divisions sharing one refined reciprocal.
    Where the step divides the edge length the count hangs on the last bit of every intermediate
    value: the number of items of a batch must equal the oracle's step-by-step count."""
    m = scenes.franka_p(obstacles=False)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    orc = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=base)
    with _Env(MJPL_FUSED_SINGLE="0"):  # (endpoint tiles and waypoint tiles: an edge's items are its interior waypoints)
        e = eng_mod.Engine(m)
    e.set_planning(qidx, base)
    e1 = eng_mod.Engine(m)  # ... a batch this small by default: the endpoint is an item, too (one more per edge)
    e1.set_planning(qidx, base)
    rng = np.random.default_rng(77)
    E, step = 6000, 0.01
    # around the home pose nothing is near contact: every endpoint survives, nothing is undecided,
    # and the number of items is the sum of the counts
    qa = base[qidx] + rng.uniform(-0.25, 0.25, size=(E, len(qidx)))
    u = rng.normal(size=(E, len(qidx)))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    # lengths: exact multiples of the step (as an RRT extension makes them), near-multiples, anything
    length = np.where(rng.random(E) < 0.6, step * rng.integers(1, 12, E), rng.uniform(0.001, 0.12, E))
    length = np.where(rng.random(E) < 0.1, length * (1 + rng.uniform(-4, 4, E) * 2.0 ** -52), length)
    qb = qa + length[:, None] * u
    if rng.random() < 2:  # a few edges along one axis only (zero components in the direction)
        qb[:200] = qa[:200]
        qb[:200, 3] += step * rng.integers(1, 9, 200)
    assert orc.valid_configs(qb).all()
    want = 0
    for i in range(E):
        w, k = qa[i].copy(), 0
        while True:
            w = oracle_mod.step(w, qb[i], step)
            if np.array_equal(w, qb[i]):
                break
            k += 1
            assert k < 64
        want += k
    got_valid = e.check_edges(qa, qb, step)
    np.testing.assert_array_equal(got_valid, orc.valid_edges(qa, qb, step, nthreads=8))
    assert e.last_undecided() == 0
    assert e.last_items() == want
    e.close()
    np.testing.assert_array_equal(e1.check_edges(qa, qb, step), got_valid)
    assert e1.last_items() == want + E
    e1.close()


@pytest.mark.parametrize("fused", ["0", "1"])
def test_long_edges_rebuild_undecided_waypoints_from_checkpoints(oracle_mod, fused):
    """A few long edges (path shortcutting) become hundreds of lane-per-waypoint items each; the
    items the filter cannot decide are rebuilt for the exact re-check by the reference's
    recurrence -- by the two persistent kernels from the nearest stored checkpoint (every 32nd
    waypoint), by the fused kernel from the start of the edge."""
    m = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    orc = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=base)
    with _Env(MJPL_FUSED=fused):
        e = eng_mod.Engine(m)
    e.set_planning(qidx, base)
    e.set_filter(True, 2e-2)  # a wide tolerance band: many undecided pairs, at every index
    rng = np.random.default_rng(9)
    lo, hi = m.jnt_range[qidx, 0], m.jnt_range[qidx, 1]
    for E in (48, 700):
        qa = rng.uniform(lo, hi, size=(E, len(qidx)))
        qb = np.clip(qa + rng.uniform(-1.2, 1.2, size=qa.shape), lo, hi)
        want, wfb, _ = orc.valid_edges(qa, qb, 0.005, nthreads=8, info=True)
        got, gfb = e.check_edges(qa, qb, 0.005, first_bad=True)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(gfb, wfb)
        assert e.last_items() > 40 * E and e.last_undecided() > 0
    e.close()


@pytest.mark.parametrize("cap", ["3000", "40000"])
def test_item_regions_overflow_to_the_walking_kernel(oracle_mod, cap):
    """The item space is split into regions with their own fill counters; an edge whose items do not
    fit its workgroup's region goes to the walking kernel, and the slots it had reserved are void.
    With a small item space several regions overflow at once."""
    m = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    orc = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=base)
    with _Env(MJPL_ITEM_CAP=cap, MJPL_FUSED="0"):  # (the two persistent kernels: the fused one has no item space)
        e = eng_mod.Engine(m)
    e.set_planning(qidx, base)
    qa, qb = random_edges(m, qidx, 20000, seed=11, eps=0.08)
    want, wfb, _ = orc.valid_edges(qa, qb, 0.01, nthreads=8, info=True)
    got, gfb = e.check_edges(qa, qb, 0.01, first_bad=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(gfb, wfb)
    assert e.last_items() > int(cap)  # more was reserved than fits: the surplus took the walking kernel
    e.close()


def test_host_pointer_paths_agree():
    """The host-pointer entry points have three ways to move a batch: the kernels reading and writing a pinned block
    themselves (up to 256 KiB, the default), one copy each way through that block (MJPL_ZERO_COPY_BYTES=0), and
    pageable copies for large batches.  Same verdicts, first-bad indices and error reports on all three."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys, hashlib
        import numpy as np
        sys.path.insert(0, {root!r})
        import bench
        from mjpl_amd import engine, scenes
        m = scenes.franka_p(obstacles=True)
        qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
        e = engine.Engine(m); e.set_planning(qidx, m.keyframe("home").qpos.copy())
        h = hashlib.sha256()
        for E in (1, 7, 300, 2000, 40000):
            qa, qb = bench.make_edges(m, qidx, E, 5 + E)
            v, fb = e.check_edges(qa, qb, 0.01, first_bad=True)
            c = e.check_configs(qb)
            h.update(v.tobytes()); h.update(fb.tobytes()); h.update(c.tobytes())
        bad = qa.copy(); bad[3, 2] = np.nan
        try:
            e.check_edges(bad, qb, 0.01)
            h.update(b"no error")
        except engine.MjplError as err:
            h.update(str(err.code).encode())
        print(h.hexdigest())
    """)
    outs = []
    for env in ({}, {"MJPL_ZERO_COPY_BYTES": "0"}, {"MJPL_ZERO_COPY_BYTES": "2000"}):
        res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(res.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2], outs


def test_engines_taking_batches_in_turns_return_a_single_engine_s_verdicts(oracle_mod):
    """EngineRing: three engines (three streams) validate six different batches in turns, all enqueued before the
    first synchronisation; every batch's verdicts and first-bad indices are the oracle's."""
    import bench
    m = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    ring = eng_mod.EngineRing(m, engines=3, planning_qidx=qidx, qpos_base=base)
    orc = oracle_mod.Oracle(m, planning_qidx=qidx, qpos_base=base)
    batches, bufs = [], []
    for k in range(6):
        E = 30000 + 777 * k
        qa, qb = bench.make_edges(m, qidx, E, 50 + k)
        e0 = ring.engines[0]
        da, db = e0.alloc(qa.nbytes).upload(qa), e0.alloc(qb.nbytes).upload(qb)
        dv, df = e0.alloc(E), e0.alloc(4 * E)
        batches.append((qa, qb, E))
        bufs.append((da, db, dv, df))
    ring.sync()
    for (qa, qb, E), (da, db, dv, df) in zip(batches, bufs):
        ring.check_edges_dev(da.ptr, db.ptr, E, 0.01, eng_mod.AOS, dv.ptr, df.ptr)
    ring.sync()
    for (qa, qb, E), (da, db, dv, df) in zip(batches, bufs):
        want, wfb, _ = orc.valid_edges(qa, qb, 0.01, nthreads=8, info=True)
        np.testing.assert_array_equal(dv.download(np.uint8, E), want)
        np.testing.assert_array_equal(df.download(np.int32, E), wfb)
    ring.close()
