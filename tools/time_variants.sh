#!/bin/bash
# Timing-only variants of the benchmark model's specialised kernels (variants/spec_<NAME>.so, built
# with -DMJPL_X_<NAME>): swaps each in for the real library, runs bench.py, restores the original.
# usage (GPU box): tools/time_variants.sh NAME...     -> gpurun_out/variant_<NAME>.json
set -u
LIB=$(python tools/build_bench_spec.py --path)
cp "$LIB" /tmp/spec_orig.so
python bench.py --steps 500 --no-cpu-baseline --no-variants > gpurun_out/variant_REAL.json 2> gpurun_out/variant_REAL.err
for v in "$@"; do
  cp "variants/spec_$v.so" "$LIB"
  python bench.py --steps 500 --no-cpu-baseline --no-variants > "gpurun_out/variant_$v.json" 2> "gpurun_out/variant_$v.err"
done
cp /tmp/spec_orig.so "$LIB"
