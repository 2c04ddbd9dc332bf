#!/bin/bash
# VALU/SALU instruction counts of the render kernel for the given builds of the library.
# usage (on the GPU box, from the repo root): bash tools/pmc_one.sh lib.so...
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
    tag=$(basename $lib .so)
    out=gpurun_out/pmc1/$tag
    mkdir -p $out
    export RM_LIB=$lib
    timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $out -o pmc -- python3 tools/profile_frame.py > $out.log 2>&1 || exit 1
    echo "== $tag"
    python3 tools/pmc_brief.py render_ $out/*.db
done
