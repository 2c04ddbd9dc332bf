#!/bin/bash
# Rehearsal of the several-GPU bench path on a box with one GPU (run there, from the repo root):
# the packed-strip parity test, the RCCL path (rm_group_frame) with a one-rank group, and 2, 4 and 3
# ranks sharing GPU 0 with the library's all-gather replaced by gloo through host memory (with and
# without FXAA), and the patch-sharded lightmap precompute with 2 ranks.  Each run's JSON line lands in gpurun_out/reh/;
# "gathered_frame_equals_single_gpu_render" must be true in every one.
set -o pipefail
mkdir -p gpurun_out/reh
timeout -k 10 300 python -u -m pytest tests/test_render_gpu.py -x -q --timeout 120 --timeout-method thread -k "strips" 2>&1 | tail -3 || exit 1
RM_BENCH_GATHER_ALONE=1 timeout -k 10 200 python bench.py --steps 40 --warmup 5 > gpurun_out/reh/alone.json 2> gpurun_out/reh/alone.err || exit 1
RM_BENCH_GATHER_ALONE=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --fxaa --width 1920 --height 1080 > gpurun_out/reh/alone_fx.json 2> gpurun_out/reh/alone_fx.err || exit 1
export RM_BENCH_BACKEND=gloo RM_BENCH_SAME_DEVICE=1
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 4 > gpurun_out/reh/g2.json 2> gpurun_out/reh/g2.err || exit 1
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 22 --warmup 4 > gpurun_out/reh/g4.json 2> gpurun_out/reh/g4.err || exit 1
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 3 --steps 10 --warmup 2 --fxaa --width 1920 --height 1080 > gpurun_out/reh/g3fx.json 2> gpurun_out/reh/g3fx.err || exit 1
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --workload radiosity --steps 5 --warmup 1 > gpurun_out/reh/rad2.txt 2> gpurun_out/reh/rad2.err || exit 1
grep -q '"n_gpus": 2' gpurun_out/reh/rad2.txt || exit 1
echo done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/reh/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["n_gpus"], "ranks", round(d["ms_per_step"], 4), "ms", d["config"].get("bytes_per_pixel_exchanged"), "B/pixel",
          "gathered == whole:", d.get("gathered_frame_equals_single_gpu_render"))
    assert d.get("gathered_frame_equals_single_gpu_render") is True, f
PY
