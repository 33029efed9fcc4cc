#!/usr/bin/env python3
"""How far can the oracle's capsule-box verdict be from MuJoCo's?  

The oracle minimises segment-to-box distance exactly; mjc_CapsuleBox searches closest-feature candidates (segment
ends over faces, the twelve edges) and tests a sphere at the best one.  oracle/capsule_box_mj.c states that
published STRUCTURE [MJ-recalled] beside the exact routine; this script runs both on
  (1) 1 000 000 seeded pose pairs: 600 000 random ones in the size ranges of the benchmark scene, 400 000 placed by
      bisection within 1e-3 .. 1e-12 m of touching (where a different minimiser could flip the verdict);
  (2) every capsule-box pair of the headline batch's first 16 384 edges -- all their endpoints and interior
      waypoints, geom poses from the oracle's FK -- that passes the bounding-sphere cull,
and writes profiles/r04_capsule_box_deviation.json: counts, disagreements and their direction.
CPU only (gcc, numpy); analysis aid, never loaded by the product, the tests or bench.py."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mjpl_amd import scenes  # noqa: E402
from oracle import pyoracle  # noqa: E402

F64P = C.POINTER(C.c_double)
U8P = C.POINTER(C.c_uint8)


def load():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libmjpl_oracle_cbx.so"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libmjpl_oracle_cbx.so"))
    lib.orc_capsule_box_compare.restype = C.c_int
    lib.orc_capsule_box_gap.restype = C.c_double
    lib.orc_capsule_box_gap_batch.restype = None
    return lib


def compare(lib, cpos, cmat, csize, bpos, bmat, bsize, margin=0.0):
    n = len(cpos)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (cpos, cmat.reshape(n, 9), csize, bpos, bmat.reshape(n, 9), bsize)]
    ve, vs, ts = np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n)
    lib.orc_capsule_box_compare(C.c_long(n), *[a.ctypes.data_as(F64P) for a in arrs], C.c_double(margin),
                                ve.ctypes.data_as(U8P), vs.ctypes.data_as(U8P), ts.ctypes.data_as(F64P))
    return ve.astype(bool), vs.astype(bool), ts


def random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], axis=1).reshape(n, 3, 3)


def summary(ve, vs):
    return {"pairs": int(len(ve)), "contacts_exact": int(ve.sum()), "contacts_structured": int(vs.sum()),
            "exact_contact_structured_free": int((ve & ~vs).sum()), "structured_contact_exact_free": int((~ve & vs).sum())}


def main():
    lib = load()
    rng = np.random.default_rng(8)
    out = {"what": "oracle capsule_box (exact segment-box minimum) vs a routine with the published structure of mjc_CapsuleBox "
                   "(oracle/capsule_box_mj.c); MuJoCo itself is not available here [MJ-recalled]"}
    # ---- (1a) random poses
    n = 600000
    csize = np.stack([rng.uniform(0.03, 0.08, n), rng.uniform(0.02, 0.2, n)], axis=1)
    bsize = rng.uniform(0.02, 0.15, size=(n, 3))
    cmat, bmat = random_rotations(rng, n), random_rotations(rng, n)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    reach = csize.sum(1) + np.linalg.norm(bsize, axis=1)
    bpos = rng.uniform(-0.8, 0.8, size=(n, 3))
    cpos = bpos + d * (rng.uniform(0.0, 1.1, n) * reach)[:, None]
    ve, vs, _ = compare(lib, cpos, cmat, csize, bpos, bmat, bsize)
    out["random_poses"] = summary(ve, vs)
    dis = np.flatnonzero(ve & ~vs)
    if len(dis):  # how deep are the contacts the structured search misses?  (gap = segment-to-box distance minus the radius)
        arrs = [np.ascontiguousarray(a[dis], dtype=np.float64) for a in (cpos, cmat.reshape(n, 9), csize, bpos, bmat.reshape(n, 9), bsize)]
        g = np.zeros(len(dis))
        lib.orc_capsule_box_gap_batch(C.c_long(len(dis)), *[a.ctypes.data_as(F64P) for a in arrs], g.ctypes.data_as(F64P))
        seg = g + csize[dis, 0]  # distance of the capsule's AXIS to the box: 0 = the axis itself enters the box
        out["random_poses"]["missed_contacts"] = {
            "axis_inside_box": int((seg < 1e-12).sum()), "shallowest_penetration_m": float(-g.max()),
            "median_penetration_m": float(-np.median(g)),
            "note": "every contact the structured search misses is one where the capsule's AXIS passes through the box (both ends "
                    "outside, over edges or corners): its candidates -- ends over faces, closest points to the twelve edges -- then "
                    "hold no point of the axis inside the box.  Upstream treats penetration with code this restatement does not "
                    "have; none of these cases is near touching"}
    # ---- (1b) poses near touching: bisection on the exact gap along the approach direction
    m = 400000
    csz, bsz = csize[:m], bsize[:m]
    cm, bm, dirs, bp = cmat[:m], bmat[:m], d[:m], bpos[:m]
    sub = 20000  # poses searched; each is then offset twenty ways: 400 000 pairs

    def gaps(dist):
        c = np.ascontiguousarray(bp[:sub] + dirs[:sub] * dist[:, None])
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (c, cm[:sub].reshape(sub, 9), csz[:sub], bp[:sub], bm[:sub].reshape(sub, 9), bsz[:sub])]
        g = np.zeros(sub)
        lib.orc_capsule_box_gap_batch(C.c_long(sub), *[a.ctypes.data_as(F64P) for a in arrs], g.ctypes.data_as(F64P))
        return g
    s_lo, s_hi = np.zeros(sub), 1.2 * reach[:sub] + 0.1
    for _ in range(45):
        mid = 0.5 * (s_lo + s_hi)
        inside = gaps(mid) <= 0
        s_lo = np.where(inside, mid, s_lo)
        s_hi = np.where(inside, s_hi, mid)
    touch = 0.5 * (s_lo + s_hi)
    offs = np.concatenate([s * 10.0 ** -np.arange(3, 13) for s in (-1.0, 1.0)])  # 20 offsets per pose: 20 000 x 20 = 400 000
    near = {}
    ve_all, vs_all = [], []
    for off in offs:
        c = bp[:sub] + dirs[:sub] * (touch + off)[:, None]
        ve2, vs2, _ = compare(lib, c, cm[:sub], csz[:sub], bp[:sub], bm[:sub], bsz[:sub])
        near[f"{off:+.0e}"] = summary(ve2, vs2)
        ve_all.append(ve2)
        vs_all.append(vs2)
    out["near_touching"] = dict(summary(np.concatenate(ve_all), np.concatenate(vs_all)), by_offset_m=near,
                                note="20 000 poses moved along their approach direction to the exact routine's touching distance "
                                     "(bisection, 45 steps), then offset by +-1e-3 .. +-1e-12 m")
    # ---- (2) the headline batch: capsule-box pairs of Franka-P + 16 obstacles
    import bench
    model = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(model, scenes.FRANKA_ARM_JOINTS)
    base = model.keyframe("home").qpos.copy()
    orc = pyoracle.Oracle(model, planning_qidx=qidx, qpos_base=base)
    E = 16384
    qa, qb = bench.make_edges(model, qidx, 262144, seed=2)
    qa, qb = qa[:E], qb[:E]
    # every configuration the edges' checks could visit: endpoint + interior waypoints by the reference's recurrence
    configs = [qb]
    w = qa.copy()
    alive = np.ones(E, bool)
    for _ in range(8):
        nxt = np.array([pyoracle.step(w[i], qb[i], bench.STEP) if alive[i] else w[i] for i in range(E)])
        alive &= ~np.all(nxt == qb, axis=1)
        if not alive.any():
            break
        configs.append(nxt[alive])
        w = nxt
    Q = np.concatenate(configs)
    fk = orc.fk(Q)
    caps = [g for g in range(model.ngeom) if model.geom_type[g] == 3 and model.body_weldid[model.geom_bodyid[g]] != 0]
    boxes = [g for g in range(model.ngeom) if model.geom_type[g] == 6 and model.geom_bodyid[g] == 0]
    tot = {"pairs": 0, "contacts_exact": 0, "contacts_structured": 0, "exact_contact_structured_free": 0, "structured_contact_exact_free": 0}
    culled = 0
    for gc in caps:
        for gb in boxes:
            cp, cmm = fk["geom_xpos"][:, gc], fk["geom_xmat"][:, gc].reshape(-1, 3, 3)
            bpz, bmm = fk["geom_xpos"][:, gb], fk["geom_xmat"][:, gb].reshape(-1, 3, 3)
            bound = model.geom_rbound[gc] + model.geom_rbound[gb]
            keep = ((cp - bpz) ** 2).sum(1) <= bound * bound  # mj_collideGeoms' bounding test: the narrowphase runs for these
            culled += int((~keep).sum())
            k = int(keep.sum())
            if k == 0:
                continue
            ve3, vs3, _ = compare(lib, cp[keep], cmm[keep], np.repeat(model.geom_size[gc][None, :2], k, 0), bpz[keep], bmm[keep],
                                  np.repeat(model.geom_size[gb][None], k, 0))
            for key, val in summary(ve3, vs3).items():
                tot[key] += val
    out["headline_batch_sample"] = dict(tot, edges=E, configurations=int(len(Q)), capsule_geoms=len(caps), box_geoms=len(boxes),
                                        pairs_culled_by_bounding_spheres=culled,
                                        note="first 16 384 edges of bench.py's rank-0 batch (seed 2): endpoints and all interior waypoints; "
                                             "80 capsule-box pairs per configuration, those passing mj_collideGeoms' bounding test compared")
    path = os.path.join(ROOT, "profiles", "r04_capsule_box_deviation.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: (v if k == "what" else {kk: vv for kk, vv in v.items() if kk not in ("by_offset_m", "note")}) for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
