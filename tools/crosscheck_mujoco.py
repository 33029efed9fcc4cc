#!/usr/bin/env python3
"""Opt-in cross-check of the CPU oracle against REAL MuJoCo (SURVEY.md sections 7 / 8c).

Runs only where ``import mujoco`` succeeds (pip install mujoco; never on the GPU box, never in the
build container, which has no such wheel).  Nothing here touches the reference's Python files: the
models are this package's own (mjpl_amd.scenes + seeded random primitive models), emitted as
primitive MJCF by mjpl_amd.model.to_mjcf and compiled by MuJoCo itself.

For every model it
  1. compiles the emitted MJCF with mujoco.MjModel.from_xml_string and compares the compiled tables
     (body tree, weld ids, joint axes / ranges, geom sizes / poses / rbound) with mjpl_amd's Model;
  2. draws N seeded configurations, runs mj_kinematics + mj_collision exactly as
     CollisionConstraint.valid_config does (reference src/mjpl/constraint/collision_constraint.py:27-30)
     and compares FK (xpos, xquat, geom_xpos, geom_xmat) and the per-configuration verdict with
     oracle/libmjpl_oracle.so;
  3. reports verdict disagreements per geom-type pair -- capsule-box and box-box separately, the two
     routines the oracle does NOT restate op for op -- with the oracle's and
     MuJoCo's contact lists of the first few disagreeing configurations;
  4. with --write regenerates tests/golden/mujoco_<model>.json (seeded inputs, MuJoCo's verdicts and
     FK of a subset), the fixtures that would pin the oracle to MuJoCo itself.

    python tools/crosscheck_mujoco.py [--n 10000] [--models franka_p,franka_p_pads,ur5e,pairs,random] [--write]

Exit status: 0 = every verdict and FK agrees (or mujoco is absent: nothing was checked, says so),
1 = disagreements were found (they are listed).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from mjpl_amd import scenes  # noqa: E402
from mjpl_amd.model import GEOM_BOX, GEOM_CAPSULE, ModelBuilder, to_mjcf  # noqa: E402

TYPE_NAMES = {0: "plane", 2: "sphere", 3: "capsule", 6: "box"}


def pair_models(rng):
    """Two free-moving geoms per model (3 slides + 3 hinges each), one model per pair routine, so
    that capsule-box and box-box verdicts can be examined in isolation."""
    out = {}
    for ta, tb in (("sphere", "sphere"), ("sphere", "capsule"), ("capsule", "capsule"), ("sphere", "box"),
                   ("capsule", "box"), ("box", "box")):
        mb = ModelBuilder()
        mb.add_geom("world", "plane", (0, 0, 0.1), pos=(0, 0, -0.4))
        for name, gt in (("a", ta), ("b", tb)):
            parent = "world"
            for k, (jt, ax) in enumerate([("slide", (1, 0, 0)), ("slide", (0, 1, 0)), ("slide", (0, 0, 1)),
                                          ("hinge", (1, 0, 0)), ("hinge", (0, 1, 0)), ("hinge", (0, 0, 1))]):
                body = f"{name}{k}"
                mb.add_body(body, parent=parent, pos=(0.02 * k, 0, 0))
                rng_j = (-0.25, 0.25) if jt == "slide" else (-3.1, 3.1)
                mb.add_joint(body, f"{name}_j{k}", jt, axis=ax, range=rng_j)
                parent = body
            size = {"sphere": (0.08,), "capsule": (0.04, 0.12), "box": (0.05, 0.09, 0.03)}[gt]
            mb.add_geom(parent, gt, size, pos=(0.01, -0.02, 0.015), quat=(0.9, 0.1, -0.3, 0.2), name=f"{name}_geom")
        out[f"pair_{ta}_{tb}"] = mb.compile()
    return out


def models(which, rng):
    out = {}
    if "franka_p" in which:
        out["franka_p"] = scenes.franka_p(obstacles=True)
    if "franka_p_pads" in which:
        out["franka_p_pads"] = scenes.franka_p(obstacles=True, pads=True)
    if "ur5e" in which:
        out["ur5e_c"] = scenes.ur5e()
    if "pairs" in which:
        out.update(pair_models(rng))
    if "random" in which:
        from test_gpu_models import random_model
        for seed in (3, 8, 1004, 1009):
            out[f"random_{seed}"] = random_model(seed)[0]
    return out


def compare_tables(model, mj):
    """Compiled-model comparison: mjpl_amd.Model vs mujoco.MjModel (same field names)."""
    bad = []
    for f in ("nq", "njnt", "nbody", "ngeom"):
        if getattr(model, f) != getattr(mj, f):
            bad.append(f"{f}: {getattr(model, f)} vs {getattr(mj, f)}")
    if bad:
        return bad
    for f, tol in (("body_parentid", 0), ("body_weldid", 0), ("body_jntnum", 0), ("jnt_type", 0), ("jnt_qposadr", 0),
                   ("geom_type", 0), ("geom_bodyid", 0), ("geom_contype", 0), ("geom_conaffinity", 0),
                   ("body_pos", 1e-15), ("body_quat", 1e-15), ("jnt_axis", 1e-15), ("jnt_pos", 1e-15), ("qpos0", 0),
                   ("geom_pos", 1e-15), ("geom_quat", 1e-15), ("geom_rbound", 1e-15), ("geom_margin", 0)):
        a, b = np.asarray(getattr(model, f), float), np.asarray(getattr(mj, f), float).reshape(np.shape(getattr(model, f)))
        err = float(np.abs(a - b).max()) if a.size else 0.0
        if err > tol:
            bad.append(f"{f}: max |diff| {err:.3g}")
    # geom_size: MuJoCo stores all three numbers; unused entries are zero in both
    a, b = np.asarray(model.geom_size), np.asarray(mj.geom_size)
    planes = np.asarray(model.geom_type) == 0
    if np.abs(a[~planes] - b[~planes]).max(initial=0.0) > 0:
        bad.append("geom_size differs")
    return bad


def make_fixture(name, model, seed, k, valid, contacts, fk, engine):
    """tests/golden/mujoco_<name>.json: the verdicts, contact lists and (first rows') FK an engine -- MuJoCo --
    gave the first k configurations of default_rng(seed).uniform(jnt_range) on the model `name` of models();
    the digest of the emitted MJCF ties the file to the model it was made with."""
    import hashlib
    return {"model": name, "seed": int(seed), "n": int(k), "engine": engine,
            "generator": "tools/crosscheck_mujoco.py --write (inputs: default_rng(seed).uniform(jnt_range), first n rows)",
            "mjcf_sha256": hashlib.sha256(to_mjcf(model, name).encode()).hexdigest(),
            "valid_bits": np.packbits(np.asarray(valid, bool)).tobytes().hex(),
            "contacts": contacts, "fk_first_rows": fk}


def model_by_name(name, seed=20250523):
    """The model a fixture names (same construction as main(): seeded)."""
    for group in ("franka_p", "franka_p_pads", "ur5e", "pairs", "random"):
        ms = models({group}, np.random.default_rng(seed))
        if name in ms:
            return ms[name]
    raise KeyError(name)


def check_fixture(path, oracle_mod=None):
    """Compare the CPU oracle with one tests/golden/mujoco_*.json.  -> report dict: verdict mismatches in all,
    split into those whose differing contacts involve ONLY capsule-box / box-box pairs (the two routines the
    oracle does not restate op for op) and the others, and the largest FK difference."""
    import hashlib
    if oracle_mod is None:
        from oracle import pyoracle as oracle_mod
    fix = json.load(open(path))
    model = model_by_name(fix["model"])
    if hashlib.sha256(to_mjcf(model, fix["model"]).encode()).hexdigest() != fix["mjcf_sha256"]:
        return {"model": fix["model"], "stale": True}
    k = fix["n"]
    Q = np.random.default_rng(fix["seed"]).uniform(model.jnt_range[:, 0], model.jnt_range[:, 1], size=(k, model.nq))
    want = np.unpackbits(np.frombuffer(bytes.fromhex(fix["valid_bits"]), np.uint8))[:k].astype(bool)
    orc = oracle_mod.Oracle(model)
    got = orc.valid_configs(Q, nthreads=2).astype(bool)
    deviating, other, examples = 0, 0, []
    for i in np.flatnonzero(got != want):
        mine = {tuple(sorted(int(x) for x in c)) for c in orc.contacts(Q[i])}
        theirs = {tuple(sorted(c)) for c in fix["contacts"][i]}
        kinds = {"-".join(sorted((TYPE_NAMES.get(int(model.geom_type[a]), "?"), TYPE_NAMES.get(int(model.geom_type[b]), "?"))))
                 for a, b in mine ^ theirs}
        if kinds and kinds <= {"box-capsule", "box-box"}:
            deviating += 1
        else:
            other += 1
        if len(examples) < 5:
            examples.append({"row": int(i), "pairs": sorted(kinds)})
    nfk = len(fix["fk_first_rows"]["xpos"])
    fk_err = 0.0
    if nfk:
        fk_o = orc.fk(Q[:nfk])
        for key in ("xpos", "xquat", "geom_xpos", "geom_xmat"):
            ref = np.asarray(fix["fk_first_rows"][key], float)
            fk_err = max(fk_err, float(np.abs(ref - np.asarray(fk_o[key]).reshape(nfk, -1)).max()))
    return {"model": fix["model"], "engine": fix["engine"], "stale": False, "n": k, "mismatches": int((got != want).sum()),
            "mismatches_capsule_box_or_box_box_only": deviating, "mismatches_other": other, "fk_max_abs_err": fk_err,
            "examples": examples}


def check_model(name, model, n, seed, mujoco, write_dir):
    from oracle import pyoracle
    xml = to_mjcf(model, name)
    mj = mujoco.MjModel.from_xml_string(xml)
    data = mujoco.MjData(mj)
    report = {"model": name, "tables": compare_tables(model, mj)}
    rng = np.random.default_rng(seed)
    lo, hi = model.jnt_range[:, 0], model.jnt_range[:, 1]
    Q = rng.uniform(lo, hi, size=(n, model.nq))
    orc = pyoracle.Oracle(model)
    want = orc.valid_configs(Q, nthreads=os.cpu_count() or 1).astype(bool)
    fk_o = orc.fk(Q[: min(n, 512)])
    got = np.zeros(n, bool)
    fk_err = {"xpos": 0.0, "xquat": 0.0, "geom_xpos": 0.0, "geom_xmat": 0.0}
    pairs = {}
    examples = []
    for i in range(n):
        data.qpos[:] = Q[i]
        mujoco.mj_kinematics(mj, data)
        mujoco.mj_collision(mj, data)
        got[i] = data.ncon == 0
        if i < len(fk_o["xpos"]):
            for k, arr in (("xpos", data.xpos), ("xquat", data.xquat), ("geom_xpos", data.geom_xpos),
                           ("geom_xmat", data.geom_xmat)):
                fk_err[k] = max(fk_err[k], float(np.abs(np.asarray(arr).reshape(fk_o[k][i].shape) - fk_o[k][i]).max()))
        if got[i] != want[i]:
            mine = [tuple(int(x) for x in c) for c in orc.contacts(Q[i])]
            theirs = [tuple(int(x) for x in data.contact.geom[c]) for c in range(data.ncon)]
            for g1, g2 in (set(map(tuple, map(sorted, mine))) ^ set(map(tuple, map(sorted, theirs)))):
                key = "-".join(sorted((TYPE_NAMES.get(int(model.geom_type[g1]), "?"), TYPE_NAMES.get(int(model.geom_type[g2]), "?"))))
                pairs[key] = pairs.get(key, 0) + 1
            if len(examples) < 5:
                examples.append({"q": Q[i].tolist(), "oracle_contacts": mine, "mujoco_contacts": theirs})
    report.update(n=n, valid_fraction=float(want.mean()), verdict_mismatches=int((got != want).sum()),
                  mismatching_pairs_by_type=pairs, fk_max_abs_err=fk_err, examples=examples,
                  mujoco_version=mujoco.__version__)
    if write_dir:
        k = min(n, 2000)
        contacts, fk = [], {"xpos": [], "xquat": [], "geom_xpos": [], "geom_xmat": []}
        for i in range(k):  # MuJoCo's own outputs, configuration by configuration
            data.qpos[:] = Q[i]
            mujoco.mj_kinematics(mj, data)
            mujoco.mj_collision(mj, data)
            contacts.append([[int(x) for x in data.contact.geom[c]] for c in range(data.ncon)])
            if i < 16:
                for key, arr in (("xpos", data.xpos), ("xquat", data.xquat), ("geom_xpos", data.geom_xpos), ("geom_xmat", data.geom_xmat)):
                    fk[key].append(np.asarray(arr, float).reshape(-1).tolist())
        fix = make_fixture(name, model, seed, k, got[:k], contacts, fk, "mujoco " + mujoco.__version__)
        with open(os.path.join(write_dir, f"mujoco_{name}.json"), "w") as f:
            json.dump(fix, f)
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=20250523)
    ap.add_argument("--models", default="franka_p,franka_p_pads,ur5e,pairs,random")
    ap.add_argument("--write", action="store_true", help="regenerate tests/golden/mujoco_<model>.json")
    ap.add_argument("--emit", default="", help="only write the emitted MJCF files into this directory (no mujoco needed)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ms = models(set(args.models.split(",")), rng)
    if args.emit:
        os.makedirs(args.emit, exist_ok=True)
        for name, m in ms.items():
            with open(os.path.join(args.emit, f"{name}.xml"), "w") as f:
                f.write(to_mjcf(m, name))
        print(f"wrote {len(ms)} MJCF files to {args.emit}")
        return 0
    try:
        import mujoco
    except ImportError:
        print("mujoco is not installed here: NOTHING WAS CHECKED.  (pip install mujoco on a machine with network "
              "access, then rerun; the oracle stays 'parity unpinned' against MuJoCo for capsule-box / box-box "
              "and for 6/7-DoF verdicts until this script has run green somewhere.)")
        return 0
    write_dir = os.path.join(ROOT, "tests", "golden") if args.write else None
    failed = False
    for name, m in ms.items():
        rep = check_model(name, m, args.n, args.seed, mujoco, write_dir)
        print(json.dumps(rep))
        fk_bad = max(rep["fk_max_abs_err"].values()) > 1e-12
        if rep["tables"] or rep["verdict_mismatches"] or fk_bad:
            failed = True
            cb = rep["mismatching_pairs_by_type"]
            print(f"  -> {name}: tables {rep['tables'] or 'ok'}; {rep['verdict_mismatches']} verdict mismatches "
                  f"(capsule-box {cb.get('box-capsule', 0)}, box-box {cb.get('box-box', 0)}, "
                  f"other {sum(v for k, v in cb.items() if k not in ('box-capsule', 'box-box'))}); FK err {rep['fk_max_abs_err']}")
    print("DISAGREEMENTS FOUND" if failed else "oracle == MuJoCo on everything checked")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
