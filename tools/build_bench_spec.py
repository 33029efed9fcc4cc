#!/usr/bin/env python3
"""(Re)build the specialised kernels of bench.py's model (BASELINE configs[2]: franka_p + 16
obstacles, arm joints planned) -- optionally as timing-only variants for tools/time_variants.sh:
    python tools/build_bench_spec.py                 -> mjpl_amd/csrc/spec/libmjpl_spec_<hash>.so
    python tools/build_bench_spec.py NOCHECK NODRAIN -> variants/spec_<NAME>.so (-DMJPL_X_<NAME>)
    python tools/build_bench_spec.py NOSLP:-fno-slp-vectorize DIFF:cull=difference
                                                     -> variants/spec_<NAME>.so with the given compiler
                                                        flags / generator options instead of a -DMJPL_X_ define
Prints the library paths (the first is the real one's: time_variants.sh swaps the variants in there)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mjpl_amd import scenes, specialise  # noqa: E402


def main():
    m = scenes.franka_p(obstacles=True)
    arm = scenes.planning_index(m, scenes.FRANKA_ARM_JOINTS)
    base = m.keyframe("home").qpos.copy()
    if sys.argv[1:] == ["--path"]:  # where the real library lives (no compilation: usable on the GPU box)
        print(specialise.spec_path(specialise.dump_program(m, (), arm, base)[3].hash))
        return
    print(specialise.build(m, (), arm, base, force=not sys.argv[1:]))
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    for arg in sys.argv[1:]:
        name, _, opts = arg.partition(":")
        flags = [f"-DMJPL_X_{name}"] if not opts else [o for o in opts.split(",") if o.startswith("-")]
        os.environ.pop("MJPL_SPEC_CULL", None)
        os.environ.pop("MJPL_SPEC_CERT", None)
        for o in opts.split(","):
            if o.startswith("cull="):
                os.environ["MJPL_SPEC_CULL"] = o[5:]
            if o.startswith("env="):  # (NAME:env=KEY=VALUE -- a generator switch)
                k_, _, v_ = o[4:].partition("=")
                os.environ[k_] = v_
            if o.startswith("cert="):  # (CERT:cert=1 -- the check generated with the edge certificate, mjpl_fused.h)
                os.environ["MJPL_SPEC_CERT"] = o[5:]
        print(specialise.build(m, (), arm, base, force=True, extra_flags=flags,
                               output=os.path.join(ROOT, "variants", f"spec_{name}.so")))
        os.environ.pop("MJPL_SPEC_CULL", None)
        os.environ.pop("MJPL_SPEC_CERT", None)


if __name__ == "__main__":
    main()
