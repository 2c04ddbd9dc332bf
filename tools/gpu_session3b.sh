#!/bin/bash
# Round-3 GPU session B: the whole GPU suite, then 8 and 2 processes sharing GPU 0 through bench.py
# (the library's exchange replaced by gloo through host memory), both exchanges.
set -o pipefail
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/test_gpu.log 2>&1 || { tail -30 $O/test_gpu.log; exit 1; }
tail -3 $O/test_gpu.log
export RM_BENCH_BACKEND=gloo RM_BENCH_SAME_DEVICE=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 8 --steps 12 --warmup 3 --preroll-seconds 0 > $O/g8.json 2> $O/g8.err || { grep -v amdgpu.ids $O/g8.err | tail -30; exit 1; }
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 2 --steps 12 --warmup 3 --preroll-seconds 0 --exchange presenter --fxaa > $O/g2p.json 2> $O/g2p.err || { grep -v amdgpu.ids $O/g2p.err | tail -30; exit 1; }
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3b/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["n_gpus"], "ranks", round(d["ms_per_step"], 4), "ms/step", round(d["value"]), d["unit"], "checks", d.get("frame_checks"),
          "eff", d.get("efficiency_vs_n1"), "other", d.get("other_exchange", {}).get("ms_per_step"))
    assert d.get("gathered_frame_equals_single_gpu_render") is True and len(d["frame_checks"]) == 2, f
PY
