#!/bin/bash
# the planner's look-ups beside the first extension's last chunks: from how many lanes still extending on (same box)
for L in 4096 12288 32768 4096 12288 32768; do MJPL_RRT_EARLY_LANES=$L python bench.py --workload rrt --steps 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('early_lanes $L', '%.2f'%d['ms_per_step'], [round(x,1) for x in d['config']['round_ms']])"; done
MJPL_RRT_EARLY_NN=0 python bench.py --workload rrt --steps 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('early off', '%.2f'%d['ms_per_step'], [round(x,1) for x in d['config']['round_ms']])"
