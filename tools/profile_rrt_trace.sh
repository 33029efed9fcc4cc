#!/bin/bash
# Kernel trace of one measured planner round (bench --workload rrt --steps 1): tools/profile_rrt_trace.sh <tag> -> gpurun_out/<tag>_rrt_kernel_stats.csv
set -u
TAG=${1:-run}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_${TAG}_rrt
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload rrt --steps 1 --no-cpu-baseline > $OUT/trace.log 2>&1
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_rrt_kernel_stats.csv
tail -1 $OUT/trace.log | cut -c1-300
rm -rf $OUT
python3 - <<P
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/${TAG}_rrt_kernel_stats.csv')))
for r in rows[:14]:
    print(r['Name'][:90], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3)
P
