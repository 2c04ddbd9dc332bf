#!/bin/bash
# Where the render kernel's wave-cycles go: parked on a wait, stalled at issue, or executing.
# usage (on the GPU box, from the repo root): bash tools/pmc_wait.sh lib.so...
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
    tag=$(basename $lib .so)
    out=gpurun_out/pmcw/$tag
    mkdir -p $out
    export RM_LIB=$lib
    timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_BUSY_CYCLES -d $out -o pmc -- python3 tools/profile_frame.py > $out.log 2>&1 || exit 1
    echo "== $tag"
    python3 tools/pmc_brief.py render_ $out/*.db
    out=gpurun_out/pmcw2/$tag
    mkdir -p $out
    timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -d $out -o pmc -- python3 tools/profile_frame.py > $out.log 2>&1 || exit 1
    python3 tools/pmc_brief.py render_ $out/*.db
done
