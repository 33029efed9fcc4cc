#!/bin/bash
# Extra PMC passes (instruction / scalar caches, LDS conflicts) for bench.py; run through gpurun.
set -u
TAG=${1:-x}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcx_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants $*"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --output-format csv -d "$OUT/p1" -- $BENCH > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/p2" -- $BENCH > "$OUT/p2.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d "$OUT/p3" -- $BENCH > "$OUT/p3.log" 2>&1
tail -3 "$OUT"/*.log
