#!/bin/bash
# VALU/SALU instruction counts of the render kernel for each ablation build under tools/abl/.
# usage (on the GPU box, from the repo root): bash tools/abl_pmc.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/abl
mkdir -p $out
for lib in tools/abl/librm_abl*.so; do
    tag=$(basename $lib .so)
    export RM_LIB=$lib
    timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS SQ_WAVES -d $out/$tag -o pmc -- python3 tools/profile_frame.py > $out/$tag.log 2>&1 || exit 1
    echo "== $tag"
    python3 tools/pmc_brief.py render_kernel $out/$tag/*.db | grep -v SQ_WAVES
done
