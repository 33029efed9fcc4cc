#!/bin/bash
# timing-only / diagnostic experiment builds: where does the time go?
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -fPIC -shared -Wno-unused-function"
mkdir -p variants
hipcc $F -DMJPL_STAMPS -o variants/lib_stamps.so mjpl_amd/csrc/mjpl_hip.hip
ls -la variants
