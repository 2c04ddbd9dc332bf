#!/bin/bash
# Round-2 measurement session (run on the GPU box from the repo root: bash tools/gpu_session2.sh <tag>):
# bench lines of every configuration, rocprofv3 kernel traces of the same commands, PMC passes (each
# in a run of its own) for the render kernel at 4K, the FXAA kernel, the stress kernel and the
# radiosity kernels.  Results under gpurun_out/<tag>/; the summaries are copied to profiles/ by hand.
set -o pipefail
tag=${1:-r2}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 "$@" > $out/$name.json 2> $out/$name.err || { echo "FAILED: $name"; tail -5 $out/$name.err; exit 1; }; echo "ok $name"; }
run bench python bench.py
run bench_fxaa python bench.py --fxaa --no-cpu-baseline
run bench_1080p_fxaa python bench.py --width 1920 --height 1080 --fxaa --no-cpu-baseline
run bench_1080p_fxaa_2inflight python bench.py --width 1920 --height 1080 --fxaa --no-cpu-baseline --frames-in-flight 2
run bench_2inflight python bench.py --no-cpu-baseline --frames-in-flight 2
run bench_stress python bench.py --workload stress --steps 5 --warmup 1
run bench_radiosity_m8 python bench.py --workload radiosity
run bench_radiosity_m64 python bench.py --workload radiosity --map-size 64 --steps 5 --warmup 1 --no-cpu-baseline
run bench_radiosity_m128 python bench.py --workload radiosity --map-size 128 --steps 2 --warmup 1 --no-cpu-baseline
trace() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_$name -o trace -- "$@" > $out/trace_$name.json 2> $out/trace_$name.log || { echo "FAILED: trace $name"; tail -5 $out/trace_$name.log; exit 1; }
  python3 tools/rocpd_summary.py $out/trace_$name/*.db 2>&1 | grep -v "at::native\|rocclr\|^== " | head -12 > $out/trace_$name.txt; echo "ok trace $name"; }
trace 4k python3 bench.py --no-cpu-baseline
trace 4k_fxaa python3 bench.py --fxaa --no-cpu-baseline
trace 1080p_fxaa python3 bench.py --width 1920 --height 1080 --fxaa --no-cpu-baseline
trace stress python3 bench.py --workload stress --steps 5 --warmup 1
trace radiosity_m8 python3 bench.py --workload radiosity --no-cpu-baseline
trace radiosity_m64 python3 bench.py --workload radiosity --map-size 64 --steps 5 --warmup 1 --no-cpu-baseline
trace radiosity_m128 python3 bench.py --workload radiosity --map-size 128 --steps 2 --warmup 1 --no-cpu-baseline
trace two_pass python3 tools/two_pass_bench.py
pmc() { name=$1; counters=$2; shift 2; timeout -s KILL 150 rocprofv3 --pmc $counters -d $out/pmc_$name -o pmc -- "$@" > $out/pmc_$name.log 2>&1 || { echo "FAILED: pmc $name"; tail -5 $out/pmc_$name.log; exit 1; }; echo "ok pmc $name"; }
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS"
pmc sq "$SQ" python3 tools/profile_frame.py
pmc fetch "FETCH_SIZE" python3 tools/profile_frame.py
pmc write "WRITE_SIZE GRBM_GUI_ACTIVE" python3 tools/profile_frame.py
export RM_W=7680 RM_H=4320 RM_FRAMES=2
pmc stress_sq "$SQ" python3 tools/stress_frame.py
pmc stress_fetch "FETCH_SIZE" python3 tools/stress_frame.py
pmc stress_write "WRITE_SIZE GRBM_GUI_ACTIVE" python3 tools/stress_frame.py
unset RM_W RM_H RM_FRAMES
pmc fxaa_fetch "FETCH_SIZE" python3 tools/fxaa_bench.py
pmc fxaa_write "WRITE_SIZE" python3 tools/fxaa_bench.py
pmc rad64_fetch "FETCH_SIZE" python3 bench.py --workload radiosity --map-size 64 --steps 2 --warmup 1 --no-cpu-baseline
pmc rad64_write "WRITE_SIZE" python3 bench.py --workload radiosity --map-size 64 --steps 2 --warmup 1 --no-cpu-baseline
{
  for p in sq fetch write; do python3 tools/pmc_brief.py render_kernel $out/pmc_$p/*.db; done
  for p in stress_sq stress_fetch stress_write; do python3 tools/pmc_brief.py render_kernel $out/pmc_$p/*.db; done
  for p in fxaa_fetch fxaa_write; do python3 tools/pmc_brief.py fxaa_kernel $out/pmc_$p/*.db; done
  for p in rad64_fetch rad64_write; do python3 tools/pmc_brief.py rad_ $out/pmc_$p/*.db; done
} > $out/pmc_summary.txt 2>&1
timeout -k 10 120 python tools/step_overhead.py 2>&1 | grep "frames in flight" > $out/step_overhead.txt
echo "session done"
