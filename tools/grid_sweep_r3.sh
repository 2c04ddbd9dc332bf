#!/bin/bash
# Spatial-index parameters again, now that a wave passes over a listed primitive with one compare
# (longer lists cost less than they did): the stress benchmark itself, one line per setting.
run() {
  printf "%s: " "$*"
  env "$@" timeout -k 10 200 python bench.py --workload stress --steps 4 --warmup 2 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms')" || exit 1
}
run RM_GRID_REACH=0.85
for r in 1.0 1.2 1.5 2.0; do run RM_GRID_REACH=$r RM_GRID_MAXLIST=64; done
for r in 1.0 1.2 1.5; do run RM_GRID_REACH=$r RM_GRID_MINLIST=24 RM_GRID_MAXLIST=64; done
for r in 1.2 1.5; do run RM_GRID_REACH=$r RM_GRID_MINLIST=24 RM_GRID_MAXLIST=96; done
run RM_GRID_REACH=1.2 RM_GRID_MAXLIST=64 RM_GRID_CELL=0.2
run RM_GRID_REACH=1.2 RM_GRID_MAXLIST=64 RM_GRID_CELL=0.33
run RM_GRID_REACH=1.2 RM_GRID_MAXLIST=64 RM_GRID_SHORTLIST=6
