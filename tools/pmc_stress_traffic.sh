#!/bin/bash
# HBM traffic of the stress frame (7680x4320, as bench.py --workload stress renders it) for one build
# of the library: FETCH_SIZE and WRITE_SIZE in passes of their own, and the SQ counters the bench
# line's roofline takes its executed lane-operations from.  Writes gpurun_out/pmct/<tag>.json.
# usage (on the GPU box, from the repo root): bash tools/pmc_stress_traffic.sh tag [lib.so]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
[ -n "$2" ] && export RM_LIB=$2
export RM_FRAMES=4
out=gpurun_out/pmct/$tag
mkdir -p $out/sq $out/fetch $out/write
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES -d $out/sq -o pmc -- python3 tools/stress_frame.py > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o pmc -- python3 tools/stress_frame.py > $out/fetch.log 2>&1 || { tail -5 $out/fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/write -o pmc -- python3 tools/stress_frame.py > $out/write.log 2>&1 || { tail -5 $out/write.log; exit 1; }
python3 - "$out" "$tag" <<'PY'
import glob, json, sqlite3, sys
out, tag = sys.argv[1], sys.argv[2]
def counters(db, needle):
    con = sqlite3.connect(db); cur = con.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = cur.execute(f"select counter_name, count(*), sum(value) from counters_collection where {name_col} like ? group by 1", (f"%{needle}%",)).fetchall()
    k = cur.execute("select count(*), avg(end-start) from kernels where name like ?", (f"%{needle}%",)).fetchone()
    return {r[0]: r[2] / r[1] for r in rows}, k
sq, k = counters(glob.glob(f"{out}/sq/*.db")[0], "render_kernel")
fe, _ = counters(glob.glob(f"{out}/fetch/*.db")[0], "render_kernel")
wr, _ = counters(glob.glob(f"{out}/write/*.db")[0], "render_kernel")
res = {"kernel": "render_kernel<GroupedLarge>", "workload": "stress scene 7680x4320 (tools/stress_frame.py = bench.py --workload stress)", "width": 7680, "height": 4320,
       "valu_lane_ops_per_launch": sq["SQ_THREAD_CYCLES_VALU"], "valu_instructions_per_launch": sq["SQ_INSTS_VALU"], "salu_instructions_per_launch": sq["SQ_INSTS_SALU"],
       "active_lanes_per_valu_instruction": sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_INSTS_VALU"], "waves": sq["SQ_WAVES"],
       "valu_busy_quad_cycles": sq["SQ_ACTIVE_INST_VALU"], "wave_quad_cycles": sq["SQ_WAVE_CYCLES"], "wave_quad_cycles_waiting": sq["SQ_WAIT_ANY"],
       "fetch_size_kib": fe["FETCH_SIZE"], "write_size_kib": wr["WRITE_SIZE"],
       "hbm_bytes_per_launch": int(2 * fe["FETCH_SIZE"] * 1024 + wr["WRITE_SIZE"] * 1024),
       "kernel_avg_ns_under_the_profiler": k[1], "launches": k[0],
       "how": "separate rocprofv3 --pmc passes (SQ counters; FETCH_SIZE; WRITE_SIZE) over tools/stress_frame.py; FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes)"}
json.dump(res, open(f"gpurun_out/pmct/{tag}.json", "w"), indent=1)
print(tag, "write GB", round(wr["WRITE_SIZE"] * 1024 / 1e9, 3), "fetch GB", round(2 * fe["FETCH_SIZE"] * 1024 / 1e9, 3), "lanes", round(res["active_lanes_per_valu_instruction"], 1),
      "VALU instr", f"{sq['SQ_INSTS_VALU']:.3e}", "kernel ms", round(k[1] / 1e6, 2))
PY
