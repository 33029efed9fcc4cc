"""Build helpers: compile the gfx950 HIP library (and, for tests, the CPU oracle) in-tree.

``hipcc`` cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting ``mjpl_amd/csrc/libmjpl_hip.so`` travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(CSRC, "libmjpl_hip.so")
LIB_LDS_PATH = os.path.join(CSRC, "libmjpl_hip_ldstables.so")

# -ffp-contract=off: one IEEE rounding per operation, the contract the CPU path is compared under.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17",
               "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _stale(target: str) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(_ROOT, "include", "mjpl_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 library cannot be built on this machine")
    return exe


def build_hip(force: bool = False, lds_tables: bool = False, verbose: bool = False) -> str:
    """Compile libmjpl_hip.so (or the LDS-staged-tables A/B variant) for gfx950."""
    target = LIB_LDS_PATH if lds_tables else LIB_PATH
    if force or _stale(target):
        cmd = [hipcc(), *HIPCC_FLAGS, "-o", target, os.path.join(CSRC, "mjpl_hip.hip")]
        if lds_tables:
            cmd.insert(1, "-DMJPL_TABLES_LDS=1")
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.run(cmd, check=True)
    return target
