#!/bin/bash
# Round 6's lines and traces on the GPU box (after profiles/pmc.json holds the round's counters): tools/round6_lines.sh <tag>
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4 > gpurun_out/${TAG}_gpu_tests.txt
bash tools/final_lines.sh $TAG > gpurun_out/${TAG}_final_lines.txt 2>&1
python bench.py --workload rrt --steps 14 --no-cpu-baseline > gpurun_out/${TAG}_bench_rrt_14.json 2>/dev/null
MJPL_RRT_EARLY_NN=0 python bench.py --workload rrt --steps 14 --no-cpu-baseline > gpurun_out/${TAG}_bench_rrt_14_noearly.json 2>/dev/null
MJPL_RRT_TRACE=2 python bench.py --workload rrt --steps 14 --no-cpu-baseline > /dev/null 2> gpurun_out/${TAG}_rrt_chunk_trace.txt
for E in 1048576 4194304; do
  python bench.py --edges $E --steps 200 --warmup 20 --no-cpu-baseline --no-variants 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_edges_$E.json
  MJPL_FUSED_CERT_MIN_EDGES=0 python bench.py --edges $E --steps 200 --warmup 20 --no-cpu-baseline --no-variants 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_edges_${E}_nocert.json
done
python tools/time_nn_trees.py > gpurun_out/${TAG}_nn_trees.txt 2>&1
bash tools/profile_rrt_trace.sh ${TAG}t 14 > gpurun_out/${TAG}_profile_rrt_trace.log 2>&1
# the nearest-neighbour scan's counters in a planner round (the pattern of tools/profile_next_rows.sh, the rrt workload only)
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_${TAG}_rrtnn
BENCH="python3 $R/bench.py --workload rrt --steps 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT.trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT.pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT.pmc2.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT "k_nearest_mfma<7, false, true" gpurun_out/${TAG}_pmc_k_nearest_mfma_cells.json > /dev/null
python3 tools/pmc_summary.py $OUT "k_nn_candidates" gpurun_out/${TAG}_pmc_k_nn_candidates.json > /dev/null
python3 tools/pmc_summary.py $OUT "k_rrt_gen_project_rows" gpurun_out/${TAG}_5rounds_pmc_k_rrt_gen_project_rows.json > /dev/null
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_rrt5_kernel_stats.csv
rm -rf $OUT
cat gpurun_out/${TAG}_gpu_tests.txt gpurun_out/${TAG}_final_lines.txt | tail -14
