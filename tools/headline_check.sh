#!/bin/bash
# The headline line three ways on the GPU box -- default, the driver's command, 2 000 steps -- and (optionally) the
# profiling passes of the headline configuration: tools/headline_check.sh <tag> [profile]
TAG=${1:-run}
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2> gpurun_out/${TAG}_bench_driver_cmd.err
for f in bench bench_driver_cmd; do
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_$f.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$f", "%.4g edges/s" % d["value"], "%.4f ms/step" % d["ms_per_step"], "kernel", r.get("kernel"), "%.4f ms" % r.get("kernel_ms", 0), "frac %.3f" % r.get("frac", 0),
      "kernels", {k: round(v, 4) for k, v in r.get("kernels_ms", {}).items()}, "variants", {k: "%.3g" % v["value"] for k, v in d.get("variants", {}).items()}, "in_turns", (d.get("in_turns") or {}).get("value"))
PY
done
if [ "${2:-}" = "profile" ]; then
  tools/profile_gpu.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
  for k in k_edges_fused k_tail; do python3 tools/pmc_summary.py gpurun_out/prof_$TAG $k gpurun_out/${TAG}_pmc_$k.json > /dev/null; done
  cp $(ls gpurun_out/prof_$TAG/trace/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats.csv
  rm -rf gpurun_out/prof_$TAG
  python3 - <<PY
import json
for k in ("k_edges_fused", "k_tail"):
    r = json.load(open("gpurun_out/${TAG}_pmc_%s.json" % k))
    print(k, {x: r.get(x) for x in ("avg_ns", "SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE", "hbm_bytes_per_launch")})
PY
fi
