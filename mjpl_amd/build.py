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
_BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17",
               "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]

# Headers whose struct layouts, table layouts and kernel templates are shared by libmjpl_hip.so and
# the per-model specialised libraries (mjpl_amd/specialise.py).  Their digest is compiled into both
# (MJPL_SRC_STAMP) and mixed into the program hash a specialised library is named by, so a library
# built from other headers is neither found nor accepted (load_spec compares the stamps).
STAMPED_HEADERS = ("mjpl_filter.h", "mjpl_fused.h", "mjpl_device.h", "mjpl_trig.h", "mjpl_pose.h", "mjpl_project.h", "mjpl_rows.h")


def src_stamp() -> int:
    import hashlib
    h = hashlib.sha256()
    for name in STAMPED_HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return int.from_bytes(h.digest()[:8], "little")


def hipcc_flags() -> list[str]:
    return [*_BASE_FLAGS, f"-DMJPL_SRC_STAMP=0x{src_stamp():016x}ull"]


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
        tmp = f"{target}.tmp{os.getpid()}"  # (complete under its final name or not at all)
        cmd = [hipcc(), *hipcc_flags(), "-o", tmp, os.path.join(CSRC, "mjpl_hip.hip")]
        if lds_tables:
            cmd.insert(1, "-DMJPL_TABLES_LDS=1")
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        try:
            subprocess.run(cmd, check=True)
            os.replace(tmp, target)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return target
