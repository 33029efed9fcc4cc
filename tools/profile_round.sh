#!/bin/bash
# One profiling round on the GPU box (through gpurun): kernel-trace stats + separate PMC passes of bench.py
# for the three engine configurations of the headline line, summaries into gpurun_out/<tag>_*.json
# usage: tools/profile_round.sh <tag>      (then copy what is to be judged into profiles/ and run
#        python tools/pmc_collect.py <tag> 262144 soa 1 1 / ... 0 1 / ... 1 0 on the copied summaries)
set -u
TAG=${1:-run}
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_gpu.sh ${TAG} > $R/gpurun_out/prof_${TAG}.log 2>&1
bash $R/tools/profile_gpu.sh ${TAG}_f64 --variant f64 > $R/gpurun_out/prof_${TAG}_f64.log 2>&1
bash $R/tools/profile_gpu.sh ${TAG}_interp --variant interpreter > $R/gpurun_out/prof_${TAG}_interp.log 2>&1
cd $R
for k in k_filter_items_pw k_filter_endpoints_pw k_tail; do
  python3 tools/pmc_summary.py gpurun_out/prof_${TAG} $k gpurun_out/${TAG}_pmc_$k.json > /dev/null
done
python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_f64 k_check_edges gpurun_out/${TAG}f64_pmc_k_check_edges.json > /dev/null
for k in k_filter_items k_filter_endpoints k_tail; do
  python3 tools/pmc_summary.py gpurun_out/prof_${TAG}_interp $k gpurun_out/${TAG}interp_pmc_$k.json > /dev/null
done
for v in "" _f64 _interp; do
  f=$(ls gpurun_out/prof_${TAG}${v}/trace/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/${TAG}${v}_kernel_stats.csv
done
python3 bench.py --steps 2000 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
ls gpurun_out/${TAG}*
