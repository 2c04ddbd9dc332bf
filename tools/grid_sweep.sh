#!/bin/bash
# Sweep of the spatial index parameters on the stress scenes (tuning aid).
for reach in ${REACHES:-0.3 0.5 0.7}; do
  for ml in ${MINLISTS:-4 8}; do
    for xl in ${MAXLISTS:-12 24 48}; do
      echo "== reach x$reach minlist $ml maxlist $xl"
      RM_VERBOSE=1 RM_GRID_REACH=$reach RM_GRID_MINLIST=$ml RM_GRID_MAXLIST=$xl timeout -k 10 120 python tools/stress_probe.py 2>&1 | grep -v "amdgpu.ids\|primary steps" || exit 1
    done
  done
done
