#!/usr/bin/env python3
"""(Re)build the specialised kernels of bench.py's model (BASELINE configs[2]: franka_p + 16
obstacles, arm joints planned) -- optionally as timing-only variants for tools/time_variants.sh:
    python tools/build_bench_spec.py                 -> mjpl_amd/csrc/spec/libmjpl_spec_<hash>.so
    python tools/build_bench_spec.py NOCHECK NODRAIN -> variants/spec_<NAME>.so (-DMJPL_X_<NAME>)
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
    for name in sys.argv[1:]:
        print(specialise.build(m, (), arm, base, force=True, extra_flags=[f"-DMJPL_X_{name}"],
                               output=os.path.join(ROOT, "variants", f"spec_{name}.so")))


if __name__ == "__main__":
    main()
