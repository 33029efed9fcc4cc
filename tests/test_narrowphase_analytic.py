"""Analytic / property tests of the oracle's primitive narrowphase (a5).  No reference test
pins capsule or box pairs, so these closed forms and brute-force checks are
what stands behind them.  CPU only."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

PLANE, SPHERE, CAPSULE, BOX = 0, 2, 3, 6
I3 = np.eye(3)


def rand_rot(rng):
    return Rotation.random(random_state=rng).as_matrix()


def brute_capsule_box(p1, R1, s1, p2, R2, s2, n=4001):
    """min over a dense sampling of the capsule segment of the exact point-box distance."""
    t = np.linspace(-1, 1, n)[:, None]
    pts = p1 + t * (R1[:, 2] * s1[1])
    loc = (pts - p2) @ R2
    d = loc - np.clip(loc, -s2, s2)
    return np.sqrt((d * d).sum(1)).min() - s1[0]


def test_sphere_sphere_closed_form(oracle_mod):
    t = oracle_mod.pair_test
    assert t(SPHERE, [0, 0, 0], I3, [0.5, 0, 0], SPHERE, [1.0, 0, 0], I3, [0.5, 0, 0]) == 1   # touching counts
    assert t(SPHERE, [0, 0, 0], I3, [0.5, 0, 0], SPHERE, [1.0 + 1e-9, 0, 0], I3, [0.5, 0, 0]) == 0
    assert t(SPHERE, [0, 0, 0], I3, [0.5, 0, 0], SPHERE, [1.2, 0, 0], I3, [0.5, 0, 0], margin=0.25) == 1


def test_plane_pairs(oracle_mod):
    t = oracle_mod.pair_test
    assert t(PLANE, [0, 0, 0], I3, [0, 0, 0], SPHERE, [3, 4, 0.1], I3, [0.1, 0, 0]) == 1
    assert t(PLANE, [0, 0, 0], I3, [0, 0, 0], SPHERE, [3, 4, 0.1001], I3, [0.1, 0, 0]) == 0
    # capsule tilted 30 deg: lowest point is centre_z - h*cos(30) - r
    R = Rotation.from_euler("y", 30, degrees=True).as_matrix()
    h, r = 0.2, 0.05
    z0 = h * np.cos(np.pi / 6) + r
    assert t(PLANE, [0, 0, 0], I3, [0, 0, 0], CAPSULE, [0, 0, z0 - 1e-9], R, [r, h, 0]) == 1
    assert t(PLANE, [0, 0, 0], I3, [0, 0, 0], CAPSULE, [0, 0, z0 + 1e-9], R, [r, h, 0]) == 0
    # box rotated 45 deg about x: lowest corner at (sy+sz)/sqrt(2)
    Rb = Rotation.from_euler("x", 45, degrees=True).as_matrix()
    s = np.array([0.3, 0.1, 0.2])
    zc = (s[1] + s[2]) / np.sqrt(2)
    assert t(PLANE, [0, 0, 0], I3, [0, 0, 0], BOX, [0, 0, zc - 1e-9], Rb, s) == 1
    assert t(PLANE, [0, 0, 0], I3, [0, 0, 0], BOX, [0, 0, zc + 1e-9], Rb, s) == 0
    # argument order is free (the table is indexed by type)
    assert t(BOX, [0, 0, zc - 1e-9], Rb, s, PLANE, [0, 0, 0], I3, [0, 0, 0]) == 1


def test_sphere_capsule_and_capsule_capsule(oracle_mod):
    t = oracle_mod.pair_test
    cap = (CAPSULE, [0, 0, 0], I3, [0.1, 0.5, 0])  # along z, |z|<=0.5, r=0.1
    assert t(SPHERE, [0.3, 0, 0.2], I3, [0.2, 0, 0], *cap) == 1          # side, dist 0.3 = r1+r2
    assert t(SPHERE, [0.3 + 1e-9, 0, 0.2], I3, [0.2, 0, 0], *cap) == 0
    assert t(SPHERE, [0, 0, 0.8], I3, [0.2, 0, 0], *cap) == 1            # cap end: 0.8-0.5 = 0.3
    assert t(SPHERE, [0, 0, 0.8 + 1e-9], I3, [0.2, 0, 0], *cap) == 0
    # perpendicular capsules crossing at distance d along x
    Rx = Rotation.from_euler("x", 90, degrees=True).as_matrix()
    for d, want in ((0.2, 1), (0.2 + 1e-9, 0)):
        assert t(CAPSULE, [0, 0, 0], I3, [0.1, 0.5, 0], CAPSULE, [d, 0, 0], Rx, [0.1, 0.5, 0]) == want
    # parallel capsules (det == 0 branch), offset sideways and lengthways
    for off, want in (([0.2, 0, 0.3], 1), ([0.2 + 1e-9, 0, 0.3], 0), ([0, 0, 1.2], 1), ([0, 0, 1.2 + 1e-9], 0)):
        assert t(CAPSULE, [0, 0, 0], I3, [0.1, 0.5, 0], CAPSULE, off, I3, [0.1, 0.5, 0]) == want
    # skew: end of one against the side of the other
    assert t(CAPSULE, [0, 0, 0], I3, [0.1, 0.5, 0], CAPSULE, [0.7, 0, 0.5], Rotation.from_euler("y", 90, degrees=True).as_matrix(), [0.1, 0.5, 0]) == 1


def test_capsule_capsule_matches_bruteforce(oracle_mod):
    rng = np.random.default_rng(0)
    t = np.linspace(-1, 1, 401)
    checked = 0
    for _ in range(400):
        p1, p2 = rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.5, 0.5, 3)
        R1, R2 = rand_rot(rng), rand_rot(rng)
        s1, s2 = [rng.uniform(0.03, 0.1), rng.uniform(0.05, 0.4), 0], [rng.uniform(0.03, 0.1), rng.uniform(0.05, 0.4), 0]
        a = p1 + t[:, None] * R1[:, 2] * s1[1]
        b = p2 + t[:, None] * R2[:, 2] * s2[1]
        dist = np.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)).min() - s1[0] - s2[0]
        if abs(dist) < 2e-3:
            continue  # inside the sampling error of the brute force
        checked += 1
        assert oracle_mod.pair_test(CAPSULE, p1, R1, s1, CAPSULE, p2, R2, s2) == int(dist < 0)
    assert checked > 300


def test_sphere_box_and_capsule_box(oracle_mod):
    t = oracle_mod.pair_test
    box = (BOX, [0, 0, 0], I3, [0.1, 0.2, 0.3])
    assert t(SPHERE, [0.05, 0.1, 0.1], I3, [0.01, 0, 0], *box) == 1        # centre inside
    assert t(SPHERE, [0.2, 0, 0], I3, [0.1, 0, 0], *box) == 1              # face, touching
    assert t(SPHERE, [0.2 + 1e-9, 0, 0], I3, [0.1, 0, 0], *box) == 0
    c = np.array([0.1, 0.2, 0.3]) + 0.1 / np.sqrt(3) * np.ones(3)          # corner direction
    assert t(SPHERE, c * (1 - 1e-9), I3, [0.1, 0, 0], *box) == 1
    assert t(SPHERE, c + 1e-9, I3, [0.1, 0, 0], *box) == 0
    rng = np.random.default_rng(1)
    checked = 0
    for _ in range(600):
        p1, p2 = rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.3, 0.3, 3)
        R1, R2 = rand_rot(rng), rand_rot(rng)
        s1 = np.array([rng.uniform(0.02, 0.08), rng.uniform(0.05, 0.4), 0])
        s2 = rng.uniform(0.03, 0.25, 3)
        d = brute_capsule_box(p1, R1, s1, p2, R2, s2)
        if abs(d) < 1e-3:
            continue
        checked += 1
        assert t(CAPSULE, p1, R1, s1, BOX, p2, R2, s2) == int(d < 0), (p1, p2)
    assert checked > 450
    # degenerate: segment parallel to a face (h_k = 0 exactly) and segment through the box
    assert t(CAPSULE, [0, 0, 0.45], Rotation.from_euler("y", 90, degrees=True).as_matrix(), [0.05, 0.5, 0], *box) == 0
    assert t(CAPSULE, [0, 0, 0.35], Rotation.from_euler("y", 90, degrees=True).as_matrix(), [0.05, 0.5, 0], *box) == 1
    assert t(CAPSULE, [0, 0, 0], I3, [0.01, 1.0, 0], *box) == 1


def test_box_box_sat(oracle_mod):
    t = oracle_mod.pair_test
    a = (BOX, [0, 0, 0], I3, [0.1, 0.1, 0.1])
    assert t(*a, BOX, [0.2, 0, 0], I3, [0.1, 0.1, 0.1]) == 1
    assert t(*a, BOX, [0.2 + 1e-9, 0, 0], I3, [0.1, 0.1, 0.1]) == 0
    # independent check: two boxes overlap iff the 12 half-space inequalities are feasible (LP)
    from scipy.optimize import linprog
    rng = np.random.default_rng(2)
    checked = edge_cases = 0
    for _ in range(300):
        p2 = rng.uniform(-0.35, 0.35, 3)
        R1, R2 = rand_rot(rng), rand_rot(rng)
        s1, s2 = rng.uniform(0.05, 0.2, 3), rng.uniform(0.05, 0.2, 3)

        def feasible(scale):
            A = np.vstack([R1.T, -R1.T, R2.T, -R2.T])
            b = np.concatenate([s1 * scale, s1 * scale, s2 * scale + R2.T @ p2, s2 * scale - R2.T @ p2])
            return linprog(np.zeros(3), A_ub=A, b_ub=b, bounds=[(None, None)] * 3).status == 0

        lo, hi = feasible(0.98), feasible(1.02)
        if lo != hi:
            continue  # within 2 % of touching
        checked += 1
        got = t(BOX, [0, 0, 0], R1, s1, BOX, p2, R2, s2)
        assert got == int(lo)
        assert got == t(BOX, p2, R2, s2, BOX, [0, 0, 0], R1, s1)  # symmetric
        # how often is the verdict decided by an edge-edge axis only?
        d = R1.T @ p2
        Rr = R1.T @ R2
        face_sep = any(abs(d[i]) > s1[i] + np.abs(Rr[i]) @ s2 for i in range(3)) or any(
            abs(Rr[:, j] @ d) > s2[j] + np.abs(Rr[:, j]) @ s1 for j in range(3))
        edge_cases += int(not lo and not face_sep)
    assert checked > 200 and edge_cases > 0


def test_rigid_motion_invariance(oracle_mod):
    """A verdict is invariant under a common rigid motion of both geoms (away from the boundary)."""
    rng = np.random.default_rng(3)
    types = [SPHERE, CAPSULE, BOX]
    for _ in range(300):
        ta, tb = rng.choice(types), rng.choice(types)
        pa, pb = rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.3, 0.3, 3)
        Ra, Rb = rand_rot(rng), rand_rot(rng)
        sa, sb = rng.uniform(0.05, 0.2, 3), rng.uniform(0.05, 0.2, 3)
        G, g = rand_rot(rng), rng.uniform(-1, 1, 3)
        base = [oracle_mod.pair_test(ta, pa, Ra, s, tb, pb, Rb, sb) for s in (sa * 0.97, sa * 1.03)]
        if base[0] != base[1]:
            continue  # too close to the boundary for a rounding-robust statement
        moved = oracle_mod.pair_test(ta, G @ pa + g, G @ Ra, sa, tb, G @ pb + g, G @ Rb, sb)
        assert moved == oracle_mod.pair_test(ta, pa, Ra, sa, tb, pb, Rb, sb)
