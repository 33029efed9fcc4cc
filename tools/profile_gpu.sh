#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate PMC passes for bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the counter passes and their kernel durations: one engine, one stream (a kernel alone on the chip)
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 100 --warmup 20 --no-cpu-baseline --no-variants $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
# ... and the kernel trace of three engines taking the steps in turns (the line's `in_turns`: kernels of consecutive batches overlap)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_streams" -- python3 $GRAFT_REPO_ROOT/bench.py --streams 3 --steps 60 --warmup 6 --no-cpu-baseline --no-variants $* > "$OUT/trace_streams.log" 2>&1
# PMC passes: counters in their own runs (no trace domains), one block-limited group per pass
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq1" -- $BENCH > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -30
tail -2 "$OUT"/*.log | head -40
