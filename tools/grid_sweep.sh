#!/bin/bash
# Sweep of the spatial index parameters on the stress scenes (tuning aid).
for reach in ${REACHES:-0.25 0.4 0.55 0.75 1.0}; do
  for cell in ${CELLS:-0.2 0.3 0.4}; do
    echo "== reach x$reach cell x$cell"
    RM_VERBOSE=1 RM_GRID_REACH=$reach RM_GRID_CELL=$cell timeout -k 10 120 python tools/stress_probe.py 2>&1 | grep -v "amdgpu.ids\|primary steps" || exit 1
  done
done
