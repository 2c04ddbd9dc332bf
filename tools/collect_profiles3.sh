#!/bin/bash
# Copies the summaries of one tools/gpu_session3.sh run (gpurun_out/<tag>/) into profiles/round3_*.
# usage (from the repo root): bash tools/collect_profiles3.sh <tag>
set -e
src=gpurun_out/${1:?tag}
for f in $src/bench*.json; do
    name=$(basename $f .json)
    tail -1 $f > profiles/round3_$name.json
done
: > profiles/round3_kernel_trace_summary.txt
for t in 4k 4k_fast 4k_fxaa 1080p_fxaa stress radiosity_m8; do
    [ -f $src/trace_$t.txt ] || continue
    echo "== rocprofv3 --kernel-trace --stats: $t  (commands in tools/gpu_session3.sh)" >> profiles/round3_kernel_trace_summary.txt
    cat $src/trace_$t.txt >> profiles/round3_kernel_trace_summary.txt
done
sed "s#$src/##;s#gpurun_out/pmct/[a-z0-9_]*/##" $src/pmc_summary.txt > profiles/round3_pmc_summary.txt
cp $src/traffic.json profiles/round3_traffic.json
cp $src/stress_pmc.json profiles/round3_stress_pmc.json
cp $src/exchange_costs.txt profiles/round3_exchange_costs.txt
cp $src/fast_tier.txt profiles/round3_fast_tier.txt
cp $src/step_overhead.txt profiles/round3_step_overhead.txt
echo "profiles/ refreshed from $src"
