#!/usr/bin/env python3
"""Writes tests/golden/narrowphase_<a>_<b>.json: for each primitive narrowphase routine
10 000 seeded poses with the verdict an INDEPENDENT brute-force statement gives them in extended precision
(tests/narrowphase_cases.py: golden-section search over exact point-to-segment / point-to-box distances,
box-box by edge clipping; numpy longdouble) -- poses clearly apart, clearly overlapping, and in labelled bands
of 1e-9 ... 1e-3 m on either side of touching; capsule pairs also nearly and exactly parallel.

A fixture holds the seed and the recipe number (attitudes, sizes, rays and classes are plain draws of the
seeded generator), the ray parameter of every pose (the one input that takes work to find), a digest of
the assembled float64 inputs, the expected verdicts and the labels; the first poses are written out in full.

    python tools/make_narrowphase_golden.py [--n 10000] [--seed 20251003]

Takes a few minutes (the bisection runs in software extended precision over all poses at once)."""
import argparse
import base64
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import narrowphase_cases as nc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=20251003)
    args = ap.parse_args()
    for t1, t2 in nc.PAIRS:
        t0 = time.time()
        c = nc.make_cases(t1, t2, args.n, args.seed)
        name = f"narrowphase_{nc.NAMES[t1]}_{nc.NAMES[t2]}.json"
        full = []
        for i in range(8):
            full.append({k: np.asarray(c[k][i]).tolist() for k in ("pos1", "mat1", "size1", "pos2", "mat2", "size2")}
                        | {"contact": bool(c["contact"][i]), "cls": int(c["cls"][i])})
        fix = {"generator": "tools/make_narrowphase_golden.py (tests/narrowphase_cases.py, recipe %d): independent brute force in "
                            "extended precision, not the oracle" % nc.RECIPE,
               "type1": t1, "type2": t2, "n": args.n, "seed": args.seed, "recipe": nc.RECIPE,
               "criterion": "contact iff surface distance <= 0 (MuJoCo's, margin 0)",
               "classes": "one hex digit per pose: 0 clearly apart (0.01 .. 0.3 m), 1 clearly overlapping, 2 + k: within "
                          "[1e-(9-k), 1e-(8-k)) m of touching, k = 0 .. 5; `apart_side` says on which side",
               "s_le_f64_b64": base64.b64encode(np.ascontiguousarray(c["s"], dtype="<f8").tobytes()).decode(),
               "inputs_sha256": nc.digest(c),
               "contact_bits": np.packbits(c["contact"]).tobytes().hex(),
               "apart_side_bits": np.packbits(c["sign"]).tobytes().hex(),
               "near_parallel_bits": np.packbits(c["near_parallel"]).tobytes().hex(),
               "cls_hex": "".join("%x" % int(v) for v in c["cls"]),
               "first_poses": full}
        with open(os.path.join(ROOT, "tests", "golden", name), "w") as f:
            json.dump(fix, f)
        print(f"{name}: {args.n} poses, {c['contact'].mean():.3f} in contact, {int(c['near_parallel'].sum())} nearly parallel, "
              f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
