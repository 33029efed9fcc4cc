cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_nn1 -- python3 $R/tools/time_nn_trees.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_nn2 -- python3 $R/tools/time_nn_trees.py > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $R/gpurun_out/pmc_nn3 -- python3 $R/tools/time_nn_trees.py > /dev/null 2>&1
cd $R
for d in pmc_nn1 pmc_nn2 pmc_nn3; do python3 tools/pmc_summary.py gpurun_out/$d "k_nearest_mfma<7, false, true>" gpurun_out/${d}_cells.json > /dev/null; python3 tools/pmc_summary.py gpurun_out/$d "k_nearest_mfma<7, false, false>" gpurun_out/${d}_full.json > /dev/null; done
rm -rf gpurun_out/pmc_nn1 gpurun_out/pmc_nn2 gpurun_out/pmc_nn3
cat gpurun_out/pmc_nn*_cells.json gpurun_out/pmc_nn*_full.json
