#!/bin/bash
# HBM traffic of the 3840x2160 default frame (tools/profile_frame.py) for one build of the library:
# FETCH_SIZE and WRITE_SIZE in passes of their own.  usage (on the GPU box): bash tools/pmc_frame_traffic.sh tag [lib.so]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
[ -n "$2" ] && export RM_LIB=$2
out=gpurun_out/pmcf/$tag
mkdir -p $out/fetch $out/write
timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o pmc -- python3 tools/profile_frame.py > $out/fetch.log 2>&1 || { tail -5 $out/fetch.log; exit 1; }
timeout -s KILL 100 rocprofv3 --pmc WRITE_SIZE -d $out/write -o pmc -- python3 tools/profile_frame.py > $out/write.log 2>&1 || { tail -5 $out/write.log; exit 1; }
echo "== $tag"
python3 tools/pmc_brief.py render_ $out/fetch/*.db $out/write/*.db | grep -v "^gpurun" 
