#!/bin/bash
# timing-only / diagnostic experiment builds: where does the time go?
# usage: tools/build_variants.sh [name=-Dflag ...]   (default: stamps, nonarrow, fkonly)
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -fPIC -shared -Wno-unused-function"
mkdir -p variants
if [ $# -eq 0 ]; then set -- stamps=-DMJPL_STAMPS nonarrow=-DMJPL_X_DRAIN_NONARROW fkonly=-DMJPL_X_Q_FKONLY; fi
for v in "$@"; do
  hipcc $F ${v#*=} -o variants/lib_${v%%=*}.so mjpl_amd/csrc/mjpl_hip.hip 2>/dev/null &
done
wait
ls -la variants
