#!/bin/bash
# Sweep of the spatial index parameters on the stress scenes (tuning aid).
for sl in ${SHORTLISTS:-2 3 4 6 8}; do
  echo "== shortlist $sl"
  RM_VERBOSE=1 RM_GRID_SHORTLIST=$sl timeout -k 10 120 python tools/stress_probe.py 2>&1 | grep -v "amdgpu.ids\|primary steps" | grep -v "level 1" || exit 1
done
