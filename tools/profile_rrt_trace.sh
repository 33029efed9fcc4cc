#!/bin/bash
# Kernel trace of planner rounds (bench --workload rrt --steps N): tools/profile_rrt_trace.sh <tag> [steps]
#   -> gpurun_out/<tag>_rrt_kernel_stats.csv, gpurun_out/<tag>_rrt_nn_calls.txt (every nearest-neighbour kernel: start, duration, stream)
set -u
TAG=${1:-run}
STEPS=${2:-1}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_${TAG}_rrt
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload rrt --steps $STEPS --no-cpu-baseline > $OUT/trace.log 2>&1
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_rrt_kernel_stats.csv
tail -1 $OUT/trace.log | grep -o '"round_ms": \[[^]]*\]'
python3 - <<P
import csv, glob
rows=list(csv.DictReader(open('$R/gpurun_out/${TAG}_rrt_kernel_stats.csv')))
for r in rows[:14]:
    print(r['Name'][:90], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3)
f=[p for p in glob.glob('$OUT/trace/*/*kernel_trace.csv')][0]
tr=list(csv.DictReader(open(f)))
t0=min(int(r['Start_Timestamp']) for r in tr)
with open('$R/gpurun_out/${TAG}_rrt_nn_calls.txt','w') as o:
    for r in tr:
        n=r['Kernel_Name']
        if 'k_nearest' in n or 'k_nn_pack' in n or ('gen_project_rows' in n and int(r['End_Timestamp'])-int(r['Start_Timestamp'])>2000000):
            o.write('%10.3f ms  +%8.3f ms  queue %s  %s\n' % ((int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r.get('Queue_Id','?'), n[:70]))
P
tail -60 $R/gpurun_out/${TAG}_rrt_nn_calls.txt
rm -rf $OUT
