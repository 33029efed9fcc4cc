"""Rebuild the Franka-P robot's scene-generic library only (what __graft_entry__.build() makes among others):
for A/B timing of generator changes -- python tools/build_generic_spec.py; python bench.py --variant generic."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mjpl_amd import specialise  # noqa: E402
from spec_models import generic_robot  # noqa: E402

m, allowed, qidx, base = generic_robot()
print(specialise.build(m, allowed, qidx, base, force=True, generic=True))
