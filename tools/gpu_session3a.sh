#!/bin/bash
# Round-3 GPU session A: the group tests (eight ranks at 3840x2160, presenter exchange, one-rank RCCL
# with a communicator per frame in flight), the bench line as the driver runs it and with 200 steps,
# the several-GPU bench path with a one-rank RCCL group, and 8 processes sharing GPU 0 over gloo.
set -o pipefail
O=gpurun_out/r3a
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_group_gpu.py -x -v --timeout 300 --timeout-method thread > $O/test_group.log 2>&1 || { tail -30 $O/test_group.log; exit 1; }
tail -3 $O/test_group.log
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err || { tail -20 $O/bench_20.err; exit 1; }
timeout -k 10 200 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200.json 2> $O/bench_200.err || { tail -20 $O/bench_200.err; exit 1; }
RM_BENCH_GATHER_ALONE=1 timeout -k 10 200 python bench.py --steps 40 --warmup 5 > $O/alone.json 2> $O/alone.err || { tail -20 $O/alone.err; exit 1; }
RM_BENCH_GATHER_ALONE=1 timeout -k 10 200 python bench.py --steps 40 --warmup 5 --exchange presenter --fxaa > $O/alone_presenter_fx.json 2> $O/alone_presenter_fx.err || { tail -20 $O/alone_presenter_fx.err; exit 1; }
export RM_BENCH_BACKEND=gloo RM_BENCH_SAME_DEVICE=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 8 --steps 12 --warmup 3 > $O/g8.json 2> $O/g8.err || { tail -30 $O/g8.err; exit 1; }
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3a/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["n_gpus"], "ranks", round(d["ms_per_step"], 4), "ms/step", round(d["value"]), d["unit"], "checks", d.get("frame_checks"),
          "clock", (d.get("shader_clock_mhz") or {}).get("under_load"), "eff", d.get("efficiency_vs_n1"))
PY
