#!/bin/bash
# Counter passes for the kernels of the "next" rows: k_pose_apply (bench --workload pose),
# k_nearest_mfma and k_rrt_gen_project (--workload rrt), k_ik_solve (--workload ik); kernel-trace stats of each.  Run through gpurun.
# usage: tools/profile_next_rows.sh <tag>  -> gpurun_out/<tag>_pmc_<kernel>.json, gpurun_out/<tag>_<workload>_kernel_stats.csv
set -u
TAG=${1:-run}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for W in "pose:--steps 5 --warmup 1" "rrt:--steps 1" "ik:--steps 2 --warmup 1" "configs:--steps 5 --warmup 1"; do
  NAME=${W%%:*}; ARGS=${W#*:}
  OUT=$R/gpurun_out/prof_${TAG}_$NAME
  mkdir -p $OUT
  BENCH="python3 $R/bench.py --workload $NAME $ARGS --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_flops -- $BENCH > $OUT/pmc_flops.log 2>&1
  cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_${NAME}_kernel_stats.csv
done
cd $R
python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_pose k_pose_apply_rows gpurun_out/${TAG}_pmc_k_pose_apply_rows.json > /dev/null
python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_rrt "k_nearest_mfma<7, false" gpurun_out/${TAG}_pmc_k_nearest_mfma.json > /dev/null  # (the scan; <7, true> is the sample pass)
python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_rrt k_rrt_gen_project_rows gpurun_out/${TAG}_pmc_k_rrt_gen_project_rows.json > /dev/null
python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_rrt k_rrt_gen_project_ahead gpurun_out/${TAG}_pmc_k_rrt_gen_project_ahead.json > /dev/null  # (the tail's rows a step ahead)
python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_ik k_ik_solve_rows gpurun_out/${TAG}_pmc_k_ik_solve_rows.json > /dev/null
python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_configs k_filter_configs gpurun_out/${TAG}_pmc_k_filter_configs.json > /dev/null
# the float64 pool kernel's issued floating-point instruction mix (bench --variant f64: k_edges_fused_f64)
OUT=$R/gpurun_out/prof_${TAG}_f64flops
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --variant f64 > $OUT.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_f64flops k_edges_fused_f64 gpurun_out/${TAG}_pmc_flops_k_edges_fused_f64.json > /dev/null
ls gpurun_out/${TAG}_*
# (the raw rocprofv3 output is bulky and gpurun_out/ travels back only below 64 MiB: the summaries are what is kept)
rm -rf gpurun_out/prof_${TAG}_pose gpurun_out/prof_${TAG}_rrt gpurun_out/prof_${TAG}_ik gpurun_out/prof_${TAG}_configs gpurun_out/prof_${TAG}_f64flops
