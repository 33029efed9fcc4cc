#!/bin/bash
# The round's lines on the GPU box, after the counters of the round are in profiles/pmc.json: tools/final_lines.sh <tag>
TAG=${1:-run}
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2> /dev/null
for w in configs pose ik rrt plan; do
  python bench.py --workload $w > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err
done
python tools/time_fused.py > gpurun_out/${TAG}_fused_sizes.json 2> /dev/null
python tools/time_host_latency.py > gpurun_out/${TAG}_host_latency.json 2> /dev/null
python tools/time_pose_latency.py > gpurun_out/${TAG}_pose_latency.json 2> /dev/null
for f in bench bench_driver_cmd bench_configs bench_pose bench_ik bench_rrt bench_plan; do
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_$f.json").read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print("$f", "%.4g %s" % (d["value"], d["unit"]), "%.4f ms" % d.get("ms_per_step", 0), "bound", r.get("bound"), "frac %.3g" % (r.get("frac") or 0), "kernel", r.get("kernel"))
except Exception as ex:
    print("$f", "FAILED", ex)
PY
done
