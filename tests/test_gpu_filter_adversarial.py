"""Adversarial parity tests of the float32 filter.

Random inputs almost never land within 1e-4 m of a contact threshold, so these tests CONSTRUCT
such configurations: for every pair routine a moving geom is driven along a slide joint towards
its partner, the joint value at which the ORACLE's verdict flips is found by bisection (to one
ulp), and configurations are placed at that value plus {0 (the two adjacent doubles), +-1e-9,
+-1e-7, +-1e-5, +-tol/2, +-0.9 tol, +-1.1 tol, +-2 tol}.  The engine (float32 filter + float64 re-check, through the
C ABI) must return the oracle's verdict for every one of them, and the near-threshold ones must
show up in mjpl_filter_last_undecided.  Variants: chains of six one-joint bodies, both geoms moving
or the partner static; geometry 60 m from the origin; a 300 m static box; sub-millimetre
capsules; a tolerance set just above the model's derived error floor.
"""
import numpy as np
import pytest

from mjpl_amd import engine as eng_mod
from mjpl_amd.model import ModelBuilder

pytestmark = pytest.mark.gpu

TOL = 1e-4
OFFSETS = np.array([0.0, 1e-9, 1e-7, 1e-5, 0.5 * TOL, 0.9 * TOL, 1.1 * TOL, 2 * TOL])


def _rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def _size(rng, gtype, scale):
    if gtype == "sphere":
        return (scale * rng.uniform(0.5, 1.5),)
    if gtype == "capsule":
        return (scale * rng.uniform(0.3, 1.0), scale * rng.uniform(0.5, 2.0))
    return tuple(scale * rng.uniform(0.4, 1.6, size=3))


def _pod(mb, name, gtype, size, origin, link=0.07):
    """Six one-joint bodies (slides x, y, z from `origin`, then hinges x, y, z with `link`-long
    offsets between them) carrying one geom at the tip: any pose is reachable, the chain is as
    deep as an arm's."""
    parent = "world"
    axes = [("slide", (1, 0, 0)), ("slide", (0, 1, 0)), ("slide", (0, 0, 1)),
            ("hinge", (1, 0, 0)), ("hinge", (0, 1, 0)), ("hinge", (0, 0, 1))]
    for k, (jt, ax) in enumerate(axes):
        pos = origin if k == 0 else ((0, 0, 0) if k < 3 else (link * (k % 2), link * ((k + 1) % 2), 0.0))
        body = f"{name}{k}"
        mb.add_body(body, parent=parent, pos=pos)
        mb.add_joint(body, f"{name}_j{k}", jt, axis=ax, range=(-1e3, 1e3))
        parent = body
    mb.add_geom(parent, gtype, size, pos=(0.01, -0.02, 0.015), quat=(0.9, 0.1, -0.3, 0.2), name=f"{name}_geom")


def _inradius(gtype, size):
    return size[0] if gtype in ("sphere", "capsule") else min(size)


def _bisect(orc, make_q, lo, hi):
    """Per-row joint value where the oracle's verdict flips: valid (no contact) at lo, invalid at hi."""
    lo, hi = lo.copy(), hi.copy()
    assert orc.valid_configs(make_q(lo), nthreads=8).all(), "bracket: far end must be contact-free"
    assert not orc.valid_configs(make_q(hi), nthreads=8).any(), "bracket: near end must be in contact"
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        done = (mid == lo) | (mid == hi)
        if done.all():
            break
        v = orc.valid_configs(make_q(mid), nthreads=8).astype(bool)
        lo = np.where(v, mid, lo)
        hi = np.where(v, hi, mid)
    return lo, hi  # adjacent doubles: lo valid, hi in contact


def _seed(*names):
    return sum(ord(c) * (k + 1) for k, c in enumerate("/".join(names)))


def _attack(oracle_mod, model, make_q, lo, hi, tol=None, expect_filter=True, offsets=OFFSETS, near_below=1e-5,
            steps_aside=False):
    """Twice: against the oracle with its bit-reproducible sin/cos (every offset, the 1-ulp
    neighbours of the threshold included), and against the oracle with libm's sin/cos as MuJoCo
    uses it (offsets from 1e-9 up: two libm's differ in the last bit of some sin/cos themselves)."""
    with oracle_mod.portable_trig():
        out = _attack_once(oracle_mod, model, make_q, lo, hi, tol, expect_filter, offsets, near_below, steps_aside)
    _attack_once(oracle_mod, model, make_q, lo, hi, tol, expect_filter, offsets[offsets > 0], near_below, steps_aside)
    return out


def _attack_once(oracle_mod, model, make_q, lo, hi, tol, expect_filter, offsets, near_below, steps_aside):
    orc = oracle_mod.Oracle(model)
    e = eng_mod.Engine(model)
    if tol is not None:
        e.set_filter(True, tol)
    info = e.info()
    assert bool(info["filter_enabled"]) == expect_filter, info
    qv, qc = _bisect(orc, make_q, lo, hi)
    rows, near = [], []
    for off in offsets:
        for base, sgn in ((qv, -1.0), (qc, +1.0)):
            rows.append(base + sgn * off * np.sign(hi - lo))
            near.append(np.full(len(base), off <= near_below))
    x = np.concatenate(rows)
    near = np.concatenate(near)
    Q = make_q(x, reps=len(rows))
    want = orc.valid_configs(Q, nthreads=8)
    got = e.check_configs(Q)
    undecided = e.last_undecided()
    np.testing.assert_array_equal(got, want)
    assert 0 < want.sum() < len(want)
    if expect_filter:
        # every configuration within `near_below` of the threshold lies inside the band: undecided
        assert undecided >= near.sum(), (undecided, int(near.sum()))
        if steps_aside:   # nothing here is for float32 to decide
            assert undecided >= len(want)
        else:             # the filter still decides the ones two bands away by itself
            assert undecided < len(want)
    # the same through the edge entry point: zero-length-free edges ending on the probes
    QA = Q.copy()
    QA[:, 0] -= 0.003 * np.sign(np.tile(hi - lo, len(rows)))
    e_got, e_fb = e.check_edges(QA, Q, 0.001, first_bad=True)
    e_want, e_wfb, _ = orc.valid_edges(QA, Q, 0.001, nthreads=8, info=True)
    np.testing.assert_array_equal(e_got, e_want)
    np.testing.assert_array_equal(e_fb, e_wfb)
    e.close()
    return info, undecided


PAIRS = [("sphere", "sphere"), ("sphere", "capsule"), ("capsule", "capsule"), ("sphere", "box"),
         ("capsule", "box"), ("box", "sphere"), ("box", "capsule"), ("box", "box")]


def _two_pod_case(rng, ta, tb, scale, centre, n):
    mb = ModelBuilder()
    sa, sb = _size(rng, ta, scale), _size(rng, tb, scale)
    _pod(mb, "a", ta, sa, origin=(centre[0] - 0.5, centre[1], centre[2]), link=0.7 * scale)
    _pod(mb, "b", tb, sb, origin=centre, link=0.7 * scale)
    model = mb.compile()
    ang = rng.uniform(-np.pi, np.pi, size=(n, 6))
    shift = rng.uniform(-0.3, 0.3, size=(n, 2)) * _inradius(ta, sa)

    # world position of each tip geom at slide values 0 comes from the oracle's FK; the A pod is
    # then translated so that at slide value x = 0 ... easier: put A's geom centre ON B's geom centre
    # at the "near" end of the bracket, and 2 * (rbound sum) away at the "far" end.
    from oracle import pyoracle
    orc = pyoracle.Oracle(model)

    def full(xa, reps=1):
        m = len(xa)
        Q = np.zeros((m, 12))
        a = np.tile(ang, (reps, 1))
        Q[:, 3:6] = a[:, :3]
        Q[:, 9:12] = a[:, 3:]
        Q[:, 0] = xa
        return Q

    fk = orc.fk(full(np.zeros(n)))
    ga, gb = model.geom("a_geom").id, model.geom("b_geom").id
    delta = fk["geom_xpos"][:, gb] - fk["geom_xpos"][:, ga]  # slides of A translate it in world axes

    def make_q(xa, reps=1):
        Q = full(xa, reps)
        Q[:, 1] = np.tile(delta[:, 1] + shift[:, 0], reps)
        Q[:, 2] = np.tile(delta[:, 2] + shift[:, 1], reps)
        return Q

    reach = model.geom_rbound[ga] + model.geom_rbound[gb]
    return model, make_q, delta[:, 0] - 2.5 * reach, delta[:, 0].copy()


@pytest.mark.parametrize("ta,tb", PAIRS)
def test_moving_pairs_at_the_threshold(oracle_mod, ta, tb):
    rng = np.random.default_rng(_seed(ta, tb))
    model, make_q, lo, hi = _two_pod_case(rng, ta, tb, scale=0.06, centre=(0.3, -0.2, 0.5), n=96)
    info, _ = _attack(oracle_mod, model, make_q, lo, hi)
    assert abs(info["filter_tol"] - TOL) < 1e-9 and info["filter_max_coord"] > 2.0


def _static_case(rng, ta, tb, scale, centre, n, big=None, origin=None):
    mb = ModelBuilder()
    sa = _size(rng, ta, scale)
    _pod(mb, "a", ta, sa, origin=origin or (centre[0], centre[1], centre[2] + 1.0), link=0.7 * scale)
    if tb == "plane":
        sb = (0, 0, 0.1)
        quat = _rand_quat(rng) if big is None else (1, 0, 0, 0)
    else:
        sb = _size(rng, tb, scale) if big is None else big
        quat = _rand_quat(rng) if big is None else (1, 0, 0, 0)
    mb.add_geom("world", tb, sb, pos=centre, quat=quat, name="b_geom")
    model = mb.compile()
    ang = rng.uniform(-np.pi, np.pi, size=(n, 3))
    from oracle import pyoracle
    orc = pyoracle.Oracle(model)

    def full(za, reps=1):
        Q = np.zeros((len(za), 6))
        Q[:, 3:6] = np.tile(ang, (reps, 1))
        Q[:, 2] = za
        return Q

    fk = orc.fk(full(np.zeros(n)))
    ga, gb = model.geom("a_geom").id, model.geom("b_geom").id
    delta = fk["geom_xpos"][:, gb] - fk["geom_xpos"][:, ga]
    shift = rng.uniform(-0.3, 0.3, size=(n, 2)) * _inradius(ta, sa)
    if tb == "plane":
        # approach along the plane normal: the slides move A in world axes
        nrm = fk["geom_xmat"][0, gb].reshape(3, 3)[:, 2]
        axis = int(np.argmax(np.abs(nrm)))
    else:
        axis = 2

    def make_q(xa, reps=1):
        Q = full(np.zeros(len(xa)), reps)
        for k in range(3):
            if k != axis:
                Q[:, k] = np.tile(delta[:, k] + (shift[:, 0] if k == (axis + 1) % 3 else shift[:, 1]), reps)
        Q[:, axis] = xa
        return Q

    reach = model.geom_rbound[ga] + (model.geom_rbound[gb] if tb != "plane" else 0.0)
    if big is not None:
        reach = model.geom_rbound[ga] + big[axis]
    far = delta[:, axis] + (2.5 * reach + 0.05) * (np.sign(nrm[axis]) if tb == "plane" else 1.0)
    near = delta[:, axis] - (model.geom_rbound[ga] * (np.sign(nrm[axis]) if tb == "plane" else 0.0))
    return model, make_q, far, near


@pytest.mark.parametrize("ta,tb", [("sphere", "plane"), ("capsule", "plane"), ("box", "plane"), ("sphere", "sphere"),
                                   ("capsule", "sphere"), ("sphere", "capsule"), ("capsule", "capsule"),
                                   ("sphere", "box"), ("capsule", "box"), ("box", "box")])
def test_static_partners_at_the_threshold(oracle_mod, ta, tb):
    rng = np.random.default_rng(_seed(ta, tb, "static"))
    model, make_q, lo, hi = _static_case(rng, ta, tb, scale=0.08, centre=(-0.4, 0.3, 0.2), n=96)
    _attack(oracle_mod, model, make_q, lo, hi)


@pytest.mark.parametrize("ta,tb", [("capsule", "capsule"), ("capsule", "box"), ("sphere", "sphere")])
def test_scene_sixty_metres_out(oracle_mod, ta, tb):
    """Everything 60 m from the origin: one float32 ulp is 4e-6 m there and the derived limit
    (mjpl_info.filter_max_coord) is well below 60 m for a 1e-4 band, so the filter must step aside
    -- and the verdicts must still be the oracle's."""
    rng = np.random.default_rng(5)
    model, make_q, lo, hi = _two_pod_case(rng, ta, tb, scale=0.06, centre=(60.0, -45.0, 20.0), n=64)
    info, undecided = _attack(oracle_mod, model, make_q, lo, hi, steps_aside=True)
    assert info["filter_max_coord"] < 45.0


def test_three_hundred_metre_static_box(oracle_mod):
    """A static box with 300 m half-extents cannot be held in float32 within the band (its size
    alone is off by up to 2e-5 m): its narrowphase rows are poisoned, every configuration that
    passes its bounding cull is re-checked in float64."""
    rng = np.random.default_rng(6)
    model, make_q, lo, hi = _static_case(rng, "capsule", "box", scale=0.08, centre=(0.0, 0.0, -300.0), n=64,
                                         big=(300.0, 300.0, 300.0), origin=(0.0, 0.0, 0.5))
    info, _ = _attack(oracle_mod, model, make_q, lo, hi, steps_aside=True)
    assert info["filter_poisoned_geoms"] == 1


@pytest.mark.parametrize("tb", ["capsule", "sphere", "box"])
def test_sub_millimetre_capsules(oracle_mod, tb):
    """Capsules with half-lengths of 1e-4 m: ma * mc is below the ABSOLUTE 1e-15 threshold at
    which the float64 routine takes its parallel branch, whatever the angle between the axes; the
    filter must not decide those pairs on the general formula."""
    rng = np.random.default_rng(7)
    model, make_q, lo, hi = _two_pod_case(rng, "capsule", tb, scale=1.5e-4, centre=(0.2, 0.1, 0.3), n=96)
    _attack(oracle_mod, model, make_q, lo, hi)


def test_tolerance_just_above_the_error_floor(oracle_mod):
    """The strongest attack on the bound itself: a band barely wider than twice the derived error
    floor A.  If the bound were optimistic, decisions at the edge of the band would flip."""
    rng = np.random.default_rng(8)
    model, make_q, lo, hi = _two_pod_case(rng, "capsule", "capsule", scale=0.06, centre=(0.3, -0.2, 0.5), n=96)
    e = eng_mod.Engine(model)
    a = e.info()["filter_err_a"]
    e.close()
    assert 0 < a < 0.4 * TOL
    tight = 2.6 * a
    offsets = np.array([0.0, 1e-9, 1e-7, 0.2 * tight, 0.5 * tight, 0.9 * tight, 1.1 * tight, 2 * tight, 1e-5])
    info, _ = _attack(oracle_mod, model, make_q, lo, hi, tol=tight, offsets=offsets, near_below=0.2 * tight)
    assert abs(info["filter_tol"] - tight) < 1e-9 * tight + 1e-12
    # a band below the floor switches the filter off for the engine instead of guessing
    _attack(oracle_mod, model, make_q, lo, hi, tol=1.5 * a, expect_filter=False)


def test_large_hinge_angles_go_to_the_exact_path(oracle_mod):
    """float32(q) is off by eps |q|: at |q| = 1000 rad that is 6e-5 rad, metres at arm's length.
    Angles beyond the filter's limit must be decided in float64."""
    rng = np.random.default_rng(9)
    model, make_q0, lo, hi = _two_pod_case(rng, "capsule", "capsule", scale=0.06, centre=(0.3, -0.2, 0.5), n=64)
    turns = 2 * np.pi * rng.integers(100, 160, size=(64, 6))

    def make_q(xa, reps=1):
        Q = make_q0(xa, reps)
        Q[:, 3:6] += np.tile(turns[:, :3], (reps, 1))
        Q[:, 9:12] += np.tile(turns[:, 3:], (reps, 1))
        return Q

    _attack(oracle_mod, model, make_q, lo, hi, steps_aside=True)
