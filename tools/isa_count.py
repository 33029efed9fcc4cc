#!/usr/bin/env python3
"""Static instruction counts of the kernels in a gfx950 library: `tools/isa_count.py LIB.so [substring ...]`.
For every kernel whose demangled name holds one of the substrings: vector / scalar / LDS / memory instructions of its
body (llvm-objdump of the embedded code object).  The Newton-chain kernels are one loop around straight-line code, so the
static count is close to what one iteration issues (the rarely taken blocks -- pinv, the sin / cos fall-back -- included)."""
import re
import subprocess
import sys
import tempfile
import os

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def code_object(lib: str, tmp: str) -> str:
    out = os.path.join(tmp, "co.o")
    subprocess.run([BUNDLER, "--unbundle", "--type=o", f"--input={lib}", f"--output={out}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def main():
    lib, subs = sys.argv[1], sys.argv[2:] or [""]
    with tempfile.TemporaryDirectory() as tmp:
        try:
            co = code_object(lib, tmp)
        except subprocess.CalledProcessError:
            # (a .so: the fat binary sits in .hip_fatbin)
            fat = os.path.join(tmp, "fat.bin")
            subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
            co = code_object(fat, tmp)
        dis = subprocess.run([OBJDUMP, "-d", "--demangle", co], check=True, capture_output=True, text=True).stdout
    cur, counts = None, {}
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            continue
        if cur is None or not any(s in cur for s in subs):
            continue
        t = line.strip().split()
        if not t or t[0].endswith(":"):
            continue
        op = t[0]
        c = counts.setdefault(cur, {"v": 0, "s": 0, "ds": 0, "mem": 0, "f64": 0, "trans": 0, "acc": 0})
        if op.startswith("v_accvgpr"):
            c["acc"] += 1
        elif op.startswith("v_"):
            c["v"] += 1
            if "_f64" in op:
                c["f64"] += 1
            if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_div_")):
                c["trans"] += 1
        elif op.startswith("s_"):
            c["s"] += 1
        elif op.startswith("ds_"):
            c["ds"] += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            c["mem"] += 1
    for k, c in counts.items():
        if k.endswith(".kd"):
            continue
        print(f"{k[:110]:110s} VALU {c['v']:6d} (f64 {c['f64']}, div/sqrt parts {c['trans']}) accvgpr {c['acc']:5d} SALU {c['s']:5d} LDS {c['ds']:4d} MEM {c['mem']:4d}")


if __name__ == "__main__":
    main()
