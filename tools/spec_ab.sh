#!/bin/bash
# A/B of a variant of the benchmark model's library (variants/spec_<NAME>.so, tools/build_bench_spec.py) against the default
# one on the GPU box: the headline, the projection, the IK seeds and planner rounds.  usage: tools/spec_ab.sh <NAME>
cd $GRAFT_REPO_ROOT
REAL=$(python tools/build_bench_spec.py --path)
cp $REAL /tmp/real_spec.so
# (whatever happens below, the library under the production name is the production build again when this script ends)
trap 'cp /tmp/real_spec.so $REAL' EXIT
run() {
  python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-variants 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  edges', d['ms_per_step'], 'ms', d['value'])"
  python bench.py --workload pose --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  pose', d['ms_per_step'], 'ms')"
  python bench.py --workload ik --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ik', d['ms_per_step'], 'ms')"
  python bench.py --workload rrt --steps 5 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"round_ms": \[[^]]*\]'
}
echo default; run
cp variants/spec_$1.so $REAL
echo $1; run
cp /tmp/real_spec.so $REAL
