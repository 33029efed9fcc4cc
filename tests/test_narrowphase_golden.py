"""The oracle's primitive narrowphase against tests/golden/narrowphase_*.json: per
routine 10 000 seeded poses whose verdicts were established by an independent brute-force statement in
extended precision (tests/narrowphase_cases.py, written by tools/make_narrowphase_golden.py) -- clearly apart,
clearly overlapping, and 1e-9 ... 1e-3 m on either side of touching; capsule pairs nearly and exactly
parallel.  No reference test pins capsule or box pairs and real MuJoCo cannot run here: this is
what stands behind those routines, including the two the oracle does not restate op for op
(capsule-box, box-box).  CPU only."""
import base64
import glob
import json
import os

import numpy as np
import pytest

import narrowphase_cases as nc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "narrowphase_*.json")))


def _load(path):
    fix = json.load(open(path))
    assert fix["recipe"] == nc.RECIPE, "fixture written by another sampling recipe: rerun tools/make_narrowphase_golden.py"
    n = fix["n"]
    d = nc.draw(fix["type1"], fix["type2"], n, fix["seed"])
    s = np.frombuffer(base64.b64decode(fix["s_le_f64_b64"]), dtype="<f8")
    c = nc.assemble(d, s)
    assert nc.digest(c) == fix["inputs_sha256"], "the seeded draws no longer reproduce the fixture's inputs"
    bits = lambda k: np.unpackbits(np.frombuffer(bytes.fromhex(fix[k]), np.uint8))[:n].astype(bool)  # noqa: E731
    cls = np.array([int(ch, 16) for ch in fix["cls_hex"]], np.uint8)
    assert np.array_equal(cls, d["cls"])
    return fix, c, bits("contact_bits"), cls, bits("near_parallel_bits")


def test_all_nine_routines_have_a_fixture():
    have = {os.path.basename(f) for f in FILES}
    assert have == {f"narrowphase_{nc.NAMES[a]}_{nc.NAMES[b]}.json" for a, b in nc.PAIRS}


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[12:-5] for f in FILES])
def test_oracle_agrees_with_the_independent_verdicts(oracle_mod, path):
    fix, c, want, cls, near = _load(path)
    t1, t2, n = fix["type1"], fix["type2"], fix["n"]
    assert n >= 10000 and 0.4 < want.mean() < 0.6
    for k in range(8):
        assert (cls == k).sum() > n // 16  # every band is populated
    got = np.array([oracle_mod.pair_test(t1, c["pos1"][i], c["mat1"][i], c["size1"][i], t2, c["pos2"][i], c["mat2"][i],
                                         c["size2"][i]) for i in range(n)])
    assert (got >= 0).all()
    bad = np.flatnonzero((got > 0) != want)
    assert len(bad) == 0, (f"{len(bad)} of {n} verdicts differ; classes {sorted(set(int(x) for x in cls[bad]))}, "
                           f"nearly parallel among them: {int(near[bad].sum())}; first: {bad[:5].tolist()}")
    # argument order is free: the routine table is indexed by type
    sw = np.array([oracle_mod.pair_test(t2, c["pos2"][i], c["mat2"][i], c["size2"][i], t1, c["pos1"][i], c["mat1"][i],
                                        c["size1"][i]) for i in range(0, n, 7)])
    assert np.array_equal(sw > 0, want[::7])
    # the poses written out in full are the first of the regenerated ones
    for i, p in enumerate(fix["first_poses"]):
        for k in ("pos1", "mat1", "size1", "pos2", "mat2", "size2"):
            assert np.array_equal(np.asarray(p[k]), c[k][i])
        assert p["contact"] == bool(want[i])


def test_the_independent_statement_reproduces_a_fixture_sample():
    """The brute-force statement itself, re-run on a slice (the full generation takes minutes): same ray
    parameters, same verdicts as the committed capsule-capsule and box-box fixtures."""
    for a, b in ((nc.CAPSULE, nc.CAPSULE), (nc.BOX, nc.BOX)):
        fix, c, want, cls, near = _load(os.path.join(GOLDEN, f"narrowphase_{nc.NAMES[a]}_{nc.NAMES[b]}.json"))
        k = 400
        LD = nc.LD
        g = nc.gap(a, c["pos1"][:k].astype(LD), c["mat1"][:k].reshape(k, 3, 3).astype(LD), c["size1"][:k].astype(LD),
                   b, c["pos2"][:k].astype(LD), c["mat2"][:k].reshape(k, 3, 3).astype(LD), c["size2"][:k].astype(LD))
        assert np.array_equal(np.asarray(g <= 0), want[:k])
