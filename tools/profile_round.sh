#!/bin/bash
# One profiling round on the GPU box (through gpurun): kernel-trace stats + separate PMC passes of bench.py
# for the three engine configurations of the headline line, summaries into gpurun_out/<tag>_*.json
# usage: tools/profile_round.sh <tag>      (then copy what is to be judged into profiles/ and run
#        python tools/pmc_collect.py <tag> 262144 soa 1 1 / <tag>f64 ... 0 1 / <tag>interp ... 1 0 / <tag>generic ... 1 2 on the
#        copied summaries; bench.py is run AFTER that, so that its line quotes this round's counters)
set -u
TAG=${1:-run}
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_gpu.sh ${TAG} > $R/gpurun_out/prof_${TAG}.log 2>&1
bash $R/tools/profile_gpu.sh ${TAG}_f64 --variant f64 > $R/gpurun_out/prof_${TAG}_f64.log 2>&1
bash $R/tools/profile_gpu.sh ${TAG}_interp --variant interpreter > $R/gpurun_out/prof_${TAG}_interp.log 2>&1
bash $R/tools/profile_gpu.sh ${TAG}_generic --variant generic > $R/gpurun_out/prof_${TAG}_generic.log 2>&1
# the two persistent kernels the fused one replaced (MJPL_FUSED=0), for the comparison in profiles/README.md
MJPL_FUSED=0 bash $R/tools/profile_gpu.sh ${TAG}_two > $R/gpurun_out/prof_${TAG}_two.log 2>&1
cd $R
for k in k_filter_items_pw k_filter_endpoints_pw k_tail; do
  python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_two $k gpurun_out/${TAG}two_pmc_$k.json > /dev/null
done
for k in k_edges_fused k_tail; do
  python3 tools/pmc_summary.py gpurun_out/prof_${TAG} $k gpurun_out/${TAG}_pmc_$k.json > /dev/null
done
python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_f64 k_edges_fused_f64 gpurun_out/${TAG}f64_pmc_k_edges_fused_f64.json > /dev/null
for k in k_edges_fused k_tail; do
  python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_interp $k gpurun_out/${TAG}interp_pmc_$k.json > /dev/null
done
for k in k_edges_fused k_tail; do
  python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_generic $k gpurun_out/${TAG}generic_pmc_$k.json > /dev/null
done
for v in "" _f64 _interp _generic _two; do
  f=$(ls gpurun_out/prof_${TAG}${v}/trace/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/${TAG}${v}_kernel_stats.csv
done
cp "$(ls gpurun_out/prof_${TAG}/trace_streams/*/*kernel_stats.csv | head -1)" gpurun_out/${TAG}_streams_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}_f64 gpurun_out/prof_${TAG}_interp gpurun_out/prof_${TAG}_generic gpurun_out/prof_${TAG}_two
ls gpurun_out/${TAG}*
