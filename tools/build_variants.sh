#!/bin/bash
# timing-only experiment builds (wrong results by construction): where does the time go?
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -fPIC -shared -Wno-unused-function"
mkdir -p variants
hipcc $F -DMJPL_X_SKIP_NARROW -o variants/lib_nonarrow.so mjpl_amd/csrc/mjpl_hip.hip &
hipcc $F -DMJPL_X_SKIP_NARROW -DMJPL_X_SKIP_WORLD -o variants/lib_nonarrow_noworld.so mjpl_amd/csrc/mjpl_hip.hip &
hipcc $F -DMJPL_X_SKIP_NARROW -DMJPL_X_SKIP_WORLD -DMJPL_X_SKIP_STORED -o variants/lib_fkonly.so mjpl_amd/csrc/mjpl_hip.hip &
hipcc $F -DMJPL_X_SKIP_WORLD -o variants/lib_noworld.so mjpl_amd/csrc/mjpl_hip.hip &
wait
ls -la variants
