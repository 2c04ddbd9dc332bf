#!/bin/bash
# One GPU session: parity tests, smoke, bench lines, rocprofv3 kernel trace and PMC passes.
# usage (on the GPU box, from the repo root): bash tools/gpu_session.sh <tag>
set -o pipefail
tag=${1:-r1}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread 2>&1 | tee $out/pytest_gpu.log | tail -4 || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $out/smoke.log || exit 1
timeout -k 10 200 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 200 python bench.py --fxaa --no-cpu-baseline > $out/bench_fxaa.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python bench.py --width 1920 --height 1080 --fxaa --no-cpu-baseline > $out/bench_1080p_fxaa.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python bench.py --width 1920 --height 1080 --fxaa --no-cpu-baseline --frames-in-flight 2 > $out/bench_1080p_fxaa_2inflight.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --frames-in-flight 2 > $out/bench_2inflight.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python bench.py --refraction --cpu-seconds 6 > $out/bench_refraction.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --workload stress --steps 5 --warmup 1 > $out/bench_stress.json 2>> $out/bench.err || exit 1
timeout -k 10 200 python bench.py --workload radiosity --no-cpu-baseline > $out/bench_radiosity.json 2>> $out/bench.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --no-cpu-baseline > $out/trace_bench.json 2> $out/trace.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS -d $out/pmc_sq -o pmc -- python3 tools/profile_frame.py > $out/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc -- python3 tools/profile_frame.py > $out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $out/pmc_write -o pmc -- python3 tools/profile_frame.py > $out/pmc_write.log 2>&1 || exit 1
python3 tools/rocpd_summary.py $out/trace/*.db 2>&1 | grep -v "at::native\|rocclr" > $out/kernel_trace_summary.txt
for p in pmc_sq pmc_fetch pmc_write; do python3 tools/pmc_brief.py render_kernel $out/$p/*.db; done > $out/pmc_summary.txt 2>&1
cat $out/bench.json; cat $out/kernel_trace_summary.txt; cat $out/pmc_summary.txt
