#!/bin/bash
# Where the stress kernel's wave-cycles go (parked on a wait, stalled at issue, executing) and what it
# issues.  usage (on the GPU box, from the repo root): bash tools/pmc_stress_wait.sh [lib.so]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RM_W=${RM_W:-3840} RM_H=${RM_H:-2160} RM_PATTERN=1 RM_FRAMES=3
[ -n "$1" ] && export RM_LIB=$1
out=gpurun_out/pmcs
mkdir -p $out/a $out/b
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_BUSY_CYCLES -d $out/a -o pmc -- python3 tools/stress_frame.py > $out/a.log 2>&1 || { tail -5 $out/a.log; exit 1; }
python3 tools/pmc_brief.py render_ $out/a/*.db
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU -d $out/b -o pmc -- python3 tools/stress_frame.py > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
python3 tools/pmc_brief.py render_ $out/b/*.db
