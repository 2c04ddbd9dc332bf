#!/bin/bash
# Spatial-index parameters against the stress benchmark itself (7680x4320, smooth union): one bench
# line per setting.  Tuning aid; run on the GPU box from the repo root.
run() {
  printf "%s: " "$*"
  env "$@" timeout -k 10 200 python bench.py --workload stress --steps 3 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms')" || exit 1
}
run RM_GRID_SHORTLIST=4
for sl in 2 3 6 8; do run RM_GRID_SHORTLIST=$sl; done
for c in 0.18 0.35 0.5; do run RM_GRID_CELL=$c; done
for r in 0.6 0.7 1.0 1.2; do run RM_GRID_REACH=$r; done
for m in 4 10; do run RM_GRID_MINLIST=$m; done
for m in 24 32 64; do run RM_GRID_MAXLIST=$m; done
