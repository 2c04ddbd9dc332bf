#!/bin/bash
# Round-3 measurement session (run on the GPU box from the repo root: bash tools/gpu_session3.sh <tag>):
# bench lines of every configuration (strict and fast tier, the several-GPU path with a one-rank RCCL
# group in both exchanges), rocprofv3 kernel traces of the same commands, PMC passes (each in a run of
# its own) for the render kernel at 4K and the stress kernel, and the per-rank cost of both exchanges.
# Results under gpurun_out/<tag>/; tools/collect_profiles3.sh copies the summaries to profiles/.
set -o pipefail
tag=${1:-r3}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 "$@" > $out/$name.json 2> $out/$name.err || { echo "FAILED: $name"; tail -5 $out/$name.err; exit 1; }; echo "ok $name"; }
run bench python bench.py
run bench_driver_shape python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run bench_fast python bench.py --fast --no-cpu-baseline
run bench_fxaa python bench.py --fxaa --no-cpu-baseline
run bench_1080p_fxaa python bench.py --width 1920 --height 1080 --fxaa --no-cpu-baseline
run bench_1080p_fxaa_fast python bench.py --width 1920 --height 1080 --fxaa --no-cpu-baseline --fast
run bench_1080p_fxaa_2inflight python bench.py --width 1920 --height 1080 --fxaa --no-cpu-baseline --frames-in-flight 2
run bench_stress python bench.py --workload stress --steps 5 --warmup 1
run bench_stress_fast python bench.py --workload stress --steps 5 --warmup 1 --fast
run bench_radiosity_m8 python bench.py --workload radiosity
run bench_radiosity_m64 python bench.py --workload radiosity --map-size 64 --steps 5 --warmup 1 --no-cpu-baseline
RM_BENCH_GATHER_ALONE=1 run bench_group_alone python bench.py --steps 60 --warmup 10
RM_BENCH_GATHER_ALONE=1 run bench_group_alone_presenter_fxaa python bench.py --steps 60 --warmup 10 --exchange presenter --fxaa
trace() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_$name -o trace -- "$@" > $out/trace_$name.json 2> $out/trace_$name.log || { echo "FAILED: trace $name"; tail -5 $out/trace_$name.log; exit 1; }
  python3 tools/rocpd_summary.py $out/trace_$name/*.db 2>&1 | grep -v "at::native\|rocclr\|^== " | head -12 > $out/trace_$name.txt; echo "ok trace $name"; }
trace 4k python3 bench.py --no-cpu-baseline --no-extras
trace 4k_fast python3 bench.py --no-cpu-baseline --no-extras --fast
trace 4k_fxaa python3 bench.py --fxaa --no-cpu-baseline --no-extras
trace 1080p_fxaa python3 bench.py --width 1920 --height 1080 --fxaa --no-cpu-baseline --no-extras
trace stress python3 bench.py --workload stress --steps 5 --warmup 1
trace radiosity_m8 python3 bench.py --workload radiosity --no-cpu-baseline
pmc() { name=$1; counters=$2; shift 2; timeout -s KILL 150 rocprofv3 --pmc $counters -d $out/pmc_$name -o pmc -- "$@" > $out/pmc_$name.log 2>&1 || { echo "FAILED: pmc $name"; tail -5 $out/pmc_$name.log; exit 1; }; echo "ok pmc $name"; }
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS"
pmc sq "$SQ" python3 tools/profile_frame.py
pmc fetch "FETCH_SIZE" python3 tools/profile_frame.py
pmc write "WRITE_SIZE GRBM_GUI_ACTIVE" python3 tools/profile_frame.py
{
  for p in sq fetch write; do python3 tools/pmc_brief.py render_ $out/pmc_$p/*.db; done
} > $out/pmc_summary.txt 2>&1
bash tools/pmc_stress_traffic.sh $tag > $out/stress_traffic.txt 2>&1 || { echo "FAILED: stress traffic"; tail -5 $out/stress_traffic.txt; exit 1; }
cp gpurun_out/pmct/$tag.json $out/stress_pmc.json
python3 tools/pmc_brief.py render_kernel gpurun_out/pmct/$tag/sq/*.db gpurun_out/pmct/$tag/fetch/*.db gpurun_out/pmct/$tag/write/*.db >> $out/pmc_summary.txt 2>&1
python3 - $out <<'PY'
import json, sqlite3, sys, glob
out = sys.argv[1]
def counters(db, needle):
    cur = sqlite3.connect(db).cursor()
    cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
    name = "kernel_name" if "kernel_name" in cols else "name"
    return {r[0]: r[2] / r[1] for r in cur.execute(f"select counter_name, count(*), sum(value) from counters_collection where {name} like ? group by 1", (f"%{needle}%",))}
fe = counters(glob.glob(f"{out}/pmc_fetch/*.db")[0], "render_dense_kernel")
wr = counters(glob.glob(f"{out}/pmc_write/*.db")[0], "render_dense_kernel")
json.dump({"kernel": "render_dense_kernel", "workload": "default scene 3840x2160", "fetch_size_kib": fe["FETCH_SIZE"], "write_size_kib": wr["WRITE_SIZE"],
           "hbm_bytes_per_launch": int((2 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024),
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE GRBM_GUI_ACTIVE in separate passes over tools/profile_frame.py; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 tallies 128-byte read requests at 64 bytes)"},
          open(f"{out}/traffic.json", "w"), indent=1)
PY
timeout -k 10 200 python tools/exchange_costs.py 2>&1 | grep -v amdgpu.ids > $out/exchange_costs.txt || { echo "FAILED: exchange costs"; exit 1; }
timeout -k 10 200 python tools/fast_ab.py 2>&1 | grep -v amdgpu.ids > $out/fast_tier.txt || { echo "FAILED: fast ab"; exit 1; }
timeout -k 10 120 python tools/step_overhead.py 2>&1 | grep "frames in flight" > $out/step_overhead.txt
echo "session done"
