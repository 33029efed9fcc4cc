#!/bin/bash
# timing-only experiment builds (wrong results by construction): where does the time go?
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -fPIC -shared -Wno-unused-function"
mkdir -p variants
hipcc $F -DMJPL_X_Q_NOPUSH -o variants/lib_q_nopush.so mjpl_amd/csrc/mjpl_hip.hip &
hipcc $F -DMJPL_X_Q_FKONLY -o variants/lib_q_fkonly.so mjpl_amd/csrc/mjpl_hip.hip &
wait
ls -la variants
