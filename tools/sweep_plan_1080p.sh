#!/bin/bash
cd $GRAFT_REPO_ROOT
export RM_W=1920 RM_H=1080 RM_ROUNDS=6 RM_FRAMES=40
for us in 25 35 45 60; do for tm in 2 3 4; do
  echo -n "split_us=$us times_mean=$tm: "
  RM_SPLIT_US=$us RM_SPLIT_TIMES_MEAN=$tm timeout -k 10 100 python tools/quick_bench.py 2>/dev/null | grep median || exit 1
done; done
