#!/bin/bash
# the IK batch (16 384 seeds) and the projection batch (131 072 rows) with the lanes per row forced (option rows_g), same box
for G in 0 8 4 1 0 8; do
  MJPL_ROWS_G=$G python bench.py --workload ik --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ik rows_g $G', '%.3f ms'%d['ms_per_step'], 'kernel', d['roofline'].get('kernel_ms_per_step'))"
done
for G in 0 8 4 1; do
  MJPL_ROWS_G=$G python bench.py --workload pose --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pose rows_g $G', '%.4f ms'%d['ms_per_step'])"
done
