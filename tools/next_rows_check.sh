#!/bin/bash
# The next rows' lines and a chunk-by-chunk trace of a planner round, on the GPU box: tools/next_rows_check.sh <tag> [tests]
# -> gpurun_out/<tag>_bench_{pose,ik,rrt}.json, gpurun_out/<tag>_rrt_trace.err
TAG=${1:-run}
cd $GRAFT_REPO_ROOT
if [ "${2:-}" = "tests" ]; then
  python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log
fi
for w in pose ik rrt; do
  python bench.py --workload $w --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_$w.json").read().strip().splitlines()[-1])
print("$w", "%.4g" % d["value"], d["unit"], "%.4f ms" % d["ms_per_step"], d["config"].get("round_ms", ""))
PY
done
MJPL_RRT_TRACE=2 python tools/time_rrt_rounds.py 131072 3 > gpurun_out/${TAG}_rrt_trace.out 2> gpurun_out/${TAG}_rrt_trace.err
grep -c "chunk" gpurun_out/${TAG}_rrt_trace.err
