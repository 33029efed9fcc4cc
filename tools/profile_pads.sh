#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + counter passes of the moving-boxes model's fused kernel
# (bench.py --variant pads: Franka-P + 16 obstacles + the Panda's ten finger-pad boxes).
# usage: tools/profile_pads.sh <tag>   -> gpurun_out/<tag>_kernel_stats.csv, <tag>_bench.json, <tag>_pmc_k_edges_fused_pads.json
#        (copy them into profiles/, then `python tools/pmc_collect.py <tag> 262144 soa 1 1`)
set -u
TAG=${1:-r04pads}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --variant pads --streams 1 --steps 100 --warmup 20 --no-cpu-baseline --no-variants"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq1" -- $BENCH > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_kernel_stats.csv
grep '^{"metric"' "$OUT/trace.log" | tail -1 > gpurun_out/${TAG}_bench.json
python3 tools/pmc_summary.py "$OUT" k_edges_fused gpurun_out/${TAG}_pmc_k_edges_fused_pads.json
head -5 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
