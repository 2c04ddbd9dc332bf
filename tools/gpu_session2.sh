#!/bin/bash
# Measurement session (run on the GPU box from the repo root: bash tools/gpu_session2.sh <tag>):
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
export RM_FRAMES=2
pmc stress_sq "$SQ" python3 tools/stress_frame.py
pmc stress_fetch "FETCH_SIZE" python3 tools/stress_frame.py
pmc stress_write "WRITE_SIZE GRBM_GUI_ACTIVE" python3 tools/stress_frame.py
unset RM_FRAMES
pmc fxaa_fetch "FETCH_SIZE" python3 tools/fxaa_bench.py
pmc fxaa_write "WRITE_SIZE" python3 tools/fxaa_bench.py
pmc rad64_fetch "FETCH_SIZE" python3 bench.py --workload radiosity --map-size 64 --steps 2 --warmup 1 --no-cpu-baseline
pmc rad64_write "WRITE_SIZE" python3 bench.py --workload radiosity --map-size 64 --steps 2 --warmup 1 --no-cpu-baseline
{
  for p in sq fetch write; do python3 tools/pmc_brief.py render_ $out/pmc_$p/*.db; done
  for p in stress_sq stress_fetch stress_write; do python3 tools/pmc_brief.py render_kernel $out/pmc_$p/*.db; done
  for p in fxaa_fetch fxaa_write; do python3 tools/pmc_brief.py fxaa_kernel $out/pmc_$p/*.db; done
  for p in rad64_fetch rad64_write; do python3 tools/pmc_brief.py rad_ $out/pmc_$p/*.db; done
} > $out/pmc_summary.txt 2>&1
python3 - $out <<'PY'
import json, sqlite3, sys, glob
out = sys.argv[1]
def counters(db, needle):
    cur = sqlite3.connect(db).cursor()
    cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
    name = "kernel_name" if "kernel_name" in cols else "name"
    return {r[0]: r[2] / r[1] for r in cur.execute(f"select counter_name, count(*), sum(value) from counters_collection where {name} like ? group by 1", (f"%{needle}%",))}
sq = counters(glob.glob(f"{out}/pmc_stress_sq/*.db")[0], "render_kernel")
fe = counters(glob.glob(f"{out}/pmc_stress_fetch/*.db")[0], "render_kernel")
wr = counters(glob.glob(f"{out}/pmc_stress_write/*.db")[0], "render_kernel")
json.dump({"kernel": "render_kernel<GroupedLarge>", "workload": "stress scene 7680x4320 (tools/stress_frame.py = bench.py --workload stress)", "width": 7680, "height": 4320,
           "valu_lane_ops_per_launch": sq["SQ_THREAD_CYCLES_VALU"], "valu_instructions_per_launch": sq["SQ_INSTS_VALU"], "salu_instructions_per_launch": sq["SQ_INSTS_SALU"],
           "active_lanes_per_valu_instruction": sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"], "waves": sq["SQ_WAVES"],
           "fetch_size_kib": fe["FETCH_SIZE"], "write_size_kib": wr["WRITE_SIZE"], "hbm_bytes_per_launch": int((2 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024),
           "how": "separate rocprofv3 --pmc passes (SQ counters; FETCH_SIZE; WRITE_SIZE) over tools/stress_frame.py; FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes)"},
          open(f"{out}/stress_pmc.json", "w"), indent=1)
fe = counters(glob.glob(f"{out}/pmc_fetch/*.db")[0], "render_dense_kernel")
wr = counters(glob.glob(f"{out}/pmc_write/*.db")[0], "render_dense_kernel")
json.dump({"kernel": "render_dense_kernel", "workload": "default scene 3840x2160", "fetch_size_kib": fe["FETCH_SIZE"], "write_size_kib": wr["WRITE_SIZE"],
           "hbm_bytes_per_launch": int((2 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024),
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE GRBM_GUI_ACTIVE in separate passes over tools/profile_frame.py; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 tallies 128-byte read requests at 64 bytes)"},
          open(f"{out}/traffic.json", "w"), indent=1)
PY
timeout -k 10 120 python tools/bounce_ab.py 2>&1 | grep MAP_SIZE > $out/bounce_depths.txt
timeout -k 10 120 python tools/step_overhead.py 2>&1 | grep "frames in flight" > $out/step_overhead.txt
echo "session done"
