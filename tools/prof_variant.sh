#!/bin/bash
# kernel-trace stats of bench.py with a timing-only library: tools/prof_variant.sh <name>
cd /tmp && export TMPDIR=/tmp
export MJPL_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib_$1.so
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_v_$1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob('$GRAFT_REPO_ROOT/gpurun_out/prof_v_$1/runc/*_kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'expand' in r['Name'] or 'items' in r['Name']: print('$1', r['Name'][:45], r['AverageNs'])
PY
