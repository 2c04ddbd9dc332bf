#!/bin/bash
# Copies the summaries of one tools/gpu_session2.sh run (gpurun_out/<tag>/) into profiles/round2_*.
# usage (from the repo root): bash tools/collect_profiles.sh <tag>
set -e
src=gpurun_out/${1:?tag}
for f in $src/bench*.json; do
    name=$(basename $f .json)
    tail -1 $f > profiles/round2_$name.json
done
: > profiles/round2_kernel_trace_summary.txt
for t in 4k 4k_fxaa 1080p_fxaa stress radiosity_m8 radiosity_m64 radiosity_m128 two_pass; do
    [ -f $src/trace_$t.txt ] || continue
    echo "== rocprofv3 --kernel-trace --stats: $t  (commands in tools/gpu_session2.sh)" >> profiles/round2_kernel_trace_summary.txt
    cat $src/trace_$t.txt >> profiles/round2_kernel_trace_summary.txt
done
sed "s#$src/##" $src/pmc_summary.txt > profiles/round2_pmc_summary.txt
cp $src/traffic.json profiles/round2_traffic.json
cp $src/stress_pmc.json profiles/round2_stress_pmc.json
cp $src/bounce_depths.txt profiles/round2_bounce_depths.txt
cp $src/step_overhead.txt profiles/round2_step_overhead.txt
echo "profiles/ refreshed from $src"
