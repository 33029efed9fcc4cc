"""Seeded poses for the primitive narrowphase routines and an INDEPENDENT statement of
their verdicts -- shared by tools/make_narrowphase_golden.py (which writes tests/golden/narrowphase_*.json)
and tests/test_narrowphase_golden.py (which checks the oracle against those files).

Independent of the oracle in method and arithmetic: nothing here follows MuJoCo's routines.  Every shape
is a core (point, segment, box) inflated by a radius; the distance between two cores is found by brute
force -- golden-section search of a convex function of one segment parameter over exact point-to-segment
/ point-to-box distances, box-box overlap by clipping every edge of either box against the other -- in
x87 extended precision (numpy longdouble, 64-bit mantissa), vectorised over the poses.

How a pose is made: geom 1 at a random place and attitude, geom 2 with a random attitude on a ray from it;
bisection on the overlap predicate finds the ray parameter s* at which the two just touch (the set of s
with overlap is an interval, both shapes being convex), and the pose takes s = s* + delta with delta drawn
from labelled bands: clearly apart, clearly overlapping, and +-10^U[-9, -3] metres around touching.  The
expected verdict is the predicate evaluated on the float64 inputs as stored, and it must agree with the
sign of delta.  Contact criterion as in MuJoCo with margin 0: distance <= 0.
"""
from __future__ import annotations

import hashlib

import numpy as np

LD = np.longdouble
PLANE, SPHERE, CAPSULE, BOX = 0, 2, 3, 6
NAMES = {PLANE: "plane", SPHERE: "sphere", CAPSULE: "capsule", BOX: "box"}
PAIRS = [(PLANE, SPHERE), (PLANE, CAPSULE), (PLANE, BOX), (SPHERE, SPHERE), (SPHERE, CAPSULE), (CAPSULE, CAPSULE),
         (SPHERE, BOX), (CAPSULE, BOX), (BOX, BOX)]
RECIPE = 3  # bump when the sampling below changes: fixtures are tied to it through the digest of the inputs

# delta classes (one hex digit per pose in the fixture)
APART, DEEP = 0, 1          # delta in U[0.01, 0.3]; s = s* U[0, 0.9]
BAND0 = 2                   # class 2 + k: |delta| in [10^-(9-k), 10^-(8-k)), k = 0..5, sign in a separate bit string


# ------------------------------------------------------------------ extended-precision geometry
def _dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def _norm(a):
    return np.sqrt(_dot(a, a))


def quat_mat64(q):
    """float64 rotation matrices [n, 3, 3] of unit quaternions (what the routines are handed)."""
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], 1),
                     np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], 1),
                     np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1)], 1)


def point_seg_dist(p, a, b):
    ab = b - a
    den = _dot(ab, ab)
    t = np.where(den > 0, _dot(p - a, ab) / np.where(den > 0, den, LD(1)), LD(0))
    t = np.clip(t, LD(0), LD(1))
    return _norm(p - (a + t[..., None] * ab))


def point_box_dist(p, c, R, half):
    loc = np.einsum("nji,nj->ni", R, p - c)  # R^T (p - c)
    d = loc - np.clip(loc, -half, half)
    return _norm(d)


def _golden_min(f, n, iters=110):
    """min over t in [0, 1] of a convex f(t) (arrays of n), by golden section."""
    g = (np.sqrt(LD(5)) - 1) / 2
    lo, hi = np.zeros(n, LD), np.ones(n, LD)
    x1, x2 = hi - g * (hi - lo), lo + g * (hi - lo)
    f1, f2 = f(x1), f(x2)
    for _ in range(iters):
        left = f1 < f2
        hi = np.where(left, x2, hi)
        lo = np.where(left, lo, x1)
        nx1, nx2 = hi - g * (hi - lo), lo + g * (hi - lo)
        # reuse one evaluation per side (recomputing both keeps the code simple and exact enough)
        x1, x2 = nx1, nx2
        f1, f2 = f(x1), f(x2)
    return np.minimum(np.minimum(f1, f2), np.minimum(f(lo), f(hi)))


def seg_seg_dist(a1, b1, a2, b2):
    return _golden_min(lambda t: point_seg_dist(a1 + t[..., None] * (b1 - a1), a2, b2), len(a1))


def seg_box_dist(a, b, c, R, half):
    return _golden_min(lambda t: point_box_dist(a + t[..., None] * (b - a), c, R, half), len(a))


def _seg_hits_box_local(p0, p1, half):
    """does the segment p0-p1 (box frame) meet the box [-half, half]^3 ?  Slab clipping."""
    d = p1 - p0
    t0, t1 = np.zeros(len(p0), LD), np.ones(len(p0), LD)
    ok = np.ones(len(p0), bool)
    for k in range(3):
        dk, pk, hk = d[:, k], p0[:, k], half[:, k]
        par = dk == 0
        inv = LD(1) / np.where(par, LD(1), dk)
        ta, tb = (-hk - pk) * inv, (hk - pk) * inv
        tn, tf = np.minimum(ta, tb), np.maximum(ta, tb)
        t0 = np.where(par, t0, np.maximum(t0, tn))
        t1 = np.where(par, t1, np.minimum(t1, tf))
        ok &= ~(par & (np.abs(pk) > hk))
    return ok & (t0 <= t1)


_EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]
_SIGNS = np.array([[1 if (i >> k) & 1 else -1 for k in range(3)] for i in range(8)], dtype=np.float64)


def box_box_overlap(c1, R1, h1, c2, R2, h2):
    out = np.zeros(len(c1), bool)
    for (ca, Ra, ha, cb, Rb, hb) in ((c1, R1, h1, c2, R2, h2), (c2, R2, h2, c1, R1, h1)):
        corners = ca[:, None, :] + np.einsum("nij,nvj->nvi", Ra, _SIGNS[None].astype(LD) * ha[:, None, :])
        loc = np.einsum("nji,nvj->nvi", Rb, corners - cb[:, None, :])  # in b's frame
        for a, b in _EDGES:
            out |= _seg_hits_box_local(loc[:, a], loc[:, b], hb)
        cl = np.einsum("nji,nj->ni", Rb, ca - cb)
        out |= np.all(np.abs(cl) <= hb, axis=1)
    return out


def gap(t1, p1, R1, s1, t2, p2, R2, s2):
    """Signed surface distance where it exists (cores apart: distance of the cores minus the radii), and for
    box-box -- no radius to subtract -- +1 / -1 by the overlap predicate.  Contact iff gap <= 0.
    All arguments longdouble arrays; types with t1 <= t2 in PAIRS order."""
    z1, z2 = R1[:, :, 2], R2[:, :, 2]
    if t1 == PLANE:
        n = z1
        if t2 == SPHERE:
            return _dot(p2 - p1, n) - s2[:, 0]
        if t2 == CAPSULE:
            e = z2 * s2[:, 1:2]
            return np.minimum(_dot(p2 + e - p1, n), _dot(p2 - e - p1, n)) - s2[:, 0]
        corners = p2[:, None, :] + np.einsum("nij,nvj->nvi", R2, _SIGNS[None].astype(LD) * s2[:, None, :])
        return np.min(np.einsum("nvi,ni->nv", corners - p1[:, None, :], n), axis=1)
    if t1 == SPHERE:
        if t2 == SPHERE:
            return _norm(p1 - p2) - s1[:, 0] - s2[:, 0]
        if t2 == CAPSULE:
            e = z2 * s2[:, 1:2]
            return point_seg_dist(p1, p2 - e, p2 + e) - s1[:, 0] - s2[:, 0]
        return point_box_dist(p1, p2, R2, s2) - s1[:, 0]
    if t1 == CAPSULE:
        e1 = z1 * s1[:, 1:2]
        if t2 == CAPSULE:
            e2 = z2 * s2[:, 1:2]
            return seg_seg_dist(p1 - e1, p1 + e1, p2 - e2, p2 + e2) - s1[:, 0] - s2[:, 0]
        return seg_box_dist(p1 - e1, p1 + e1, p2, R2, s2) - s1[:, 0]
    return np.where(box_box_overlap(p1, R1, s1, p2, R2, s2), LD(-1), LD(1))


# ------------------------------------------------------------------ sampling
def _sizes(rng, t, n):
    s = np.zeros((n, 3))
    if t == SPHERE:
        s[:, 0] = rng.uniform(0.02, 0.2, n)
    elif t == CAPSULE:
        s[:, 0], s[:, 1] = rng.uniform(0.02, 0.1, n), rng.uniform(0.05, 0.3, n)
    elif t == BOX:
        s[:] = rng.uniform(0.03, 0.25, (n, 3))
    return s


def _bound(t, s):
    return {PLANE: 0 * s[:, 0], SPHERE: s[:, 0], CAPSULE: s[:, 0] + s[:, 1], BOX: np.linalg.norm(s, axis=1)}[t]


def draw(t1: int, t2: int, n: int, seed: int):
    """Everything of a case set that is a plain draw from the seeded generator (cheap): attitudes, sizes,
    the ray, the delta classes.  The ray parameter s of every pose is what takes work (`solve`), and what a
    fixture stores."""
    rng = np.random.default_rng([seed, t1, t2, RECIPE])
    p1 = rng.uniform(-0.5, 0.5, (n, 3))
    q1, q2 = rng.normal(size=(n, 4)), rng.normal(size=(n, 4))
    near = np.zeros(n, bool)
    if (t1, t2) == (CAPSULE, CAPSULE):
        # a fifth of the capsule pairs nearly parallel (angle 10^U[-9, -3]), a fiftieth exactly parallel
        kind = rng.uniform(size=n)
        near = kind < 0.22
        ang = np.where(kind < 0.02, 0.0, 10.0 ** rng.uniform(-9, -3, n))
        ax = rng.normal(size=(n, 3))
        ax /= np.linalg.norm(ax, axis=1, keepdims=True)
        dq = np.concatenate([np.cos(ang / 2)[:, None], np.sin(ang / 2)[:, None] * ax], axis=1)
        w1, x1, y1, z1 = (q1 / np.linalg.norm(q1, axis=1, keepdims=True)).T
        w2, x2, y2, z2 = dq.T
        prod = np.stack([w2 * w1 - x2 * x1 - y2 * y1 - z2 * z1, w2 * x1 + x2 * w1 + y2 * z1 - z2 * y1,
                         w2 * y1 - x2 * z1 + y2 * w1 + z2 * x1, w2 * z1 + x2 * y1 - y2 * x1 + z2 * w1], 1)
        q2 = np.where(near[:, None], prod, q2)
    R1, R2 = quat_mat64(q1), quat_mat64(q2)
    s1, s2 = _sizes(rng, t1, n), _sizes(rng, t2, n)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    if t1 == PLANE:  # leave the plane on its upper side, not too flat
        nrm = R1[:, :, 2]
        d = np.sum(u * nrm, axis=1, keepdims=True)
        u = u - d * nrm + np.abs(d) * nrm + 0.3 * nrm
        u /= np.linalg.norm(u, axis=1, keepdims=True)
    cls = rng.integers(0, 8, n).astype(np.uint8)   # 0 apart, 1 deep, 2..7 bands
    sign = rng.integers(0, 2, n).astype(bool)      # bands: True = apart side
    mag = 10.0 ** (-(9 - (cls.astype(float) - BAND0)) + rng.uniform(0, 1, n))
    delta = np.where(cls == APART, rng.uniform(0.01, 0.3, n), np.where(sign, mag, -mag))
    frac = rng.uniform(0, 0.9, n)
    return dict(t1=t1, t2=t2, n=n, p1=p1, R1=R1, s1=s1, R2=R2, s2=s2, u=u, cls=cls, sign=sign, delta=delta, frac=frac,
                near_parallel=near)


def solve(d):
    """The ray parameter of every pose (float64 [n]) and the expected verdicts, in extended precision."""
    t1, t2, n = d["t1"], d["t2"], d["n"]
    P1, r1, S1, r2, S2, U = (d[k].astype(LD) for k in ("p1", "R1", "s1", "R2", "s2", "u"))

    def g(s):
        return gap(t1, P1, r1, S1, t2, P1 + s[:, None] * U, r2, S2)

    lo = np.zeros(n, LD)
    hi = (_bound(t1, d["s1"]) + _bound(t2, d["s2"]) + 1.0).astype(LD) / (0.3 if t1 == PLANE else 1.0)
    assert np.all(g(lo) <= 0) and np.all(g(hi) > 0)
    for _ in range(72):
        mid = (lo + hi) / 2
        inside = g(mid) <= 0
        lo, hi = np.where(inside, mid, lo), np.where(inside, hi, mid)
    sstar = ((lo + hi) / 2).astype(np.float64)
    cls = d["cls"]
    s = np.where(cls == DEEP, sstar * d["frac"], sstar + d["delta"])
    s = np.maximum(s, 0.0)
    c = assemble(d, s)
    contact = gap(t1, P1, r1, S1, t2, c["pos2"].astype(LD), r2, S2) <= 0
    want = np.where(cls == APART, False, np.where(cls == DEEP, True, ~d["sign"]))
    # (a band pose whose touching point lies within its own delta of s = 0 may be clipped there)
    assert np.array_equal(contact, want) or np.all(s[contact != want] == 0.0), (t1, t2, int((contact != want).sum()))
    return s, np.asarray(contact, bool)


def assemble(d, s):
    """The float64 inputs of the routines for ray parameters s: what a test hands to the code under test."""
    n = d["n"]
    return dict(pos1=d["p1"], mat1=d["R1"].reshape(n, 9), size1=d["s1"], pos2=d["p1"] + s[:, None] * d["u"],
                mat2=d["R2"].reshape(n, 9), size2=d["s2"])


def make_cases(t1: int, t2: int, n: int, seed: int):
    d = draw(t1, t2, n, seed)
    s, contact = solve(d)
    c = assemble(d, s)
    c.update(cls=d["cls"], sign=d["sign"], near_parallel=d["near_parallel"], contact=contact, s=s)
    return c


def digest(c) -> str:
    h = hashlib.sha256()
    for k in ("pos1", "mat1", "size1", "pos2", "mat2", "size2"):
        h.update(np.ascontiguousarray(c[k], dtype="<f8").tobytes())
    return h.hexdigest()
