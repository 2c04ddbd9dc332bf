#!/bin/bash
# Round-3 GPU session C: the fast tier's tests, the whole GPU suite, the default bench line with its
# fast-tier field, and the fast tier at the sizes where the instruction count is the frame time.
set -o pipefail
O=gpurun_out/r3c
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_fast_tier_gpu.py -x -q -s --timeout 300 --timeout-method thread > $O/test_fast.log 2>&1 || { tail -30 $O/test_fast.log; exit 1; }
grep -h "fast tier" $O/test_fast.log; tail -2 $O/test_fast.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/test_gpu.log 2>&1 || { tail -30 $O/test_gpu.log; exit 1; }
tail -2 $O/test_gpu.log
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err || { tail -20 $O/bench_20.err; exit 1; }
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --fast --no-cpu-baseline --width 1920 --height 1080 --fxaa > $O/bench_fast_1080p_fxaa.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --width 1920 --height 1080 --fxaa > $O/bench_strict_1080p_fxaa.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
timeout -k 10 300 python bench.py --workload stress --steps 10 --warmup 3 --fast > $O/bench_fast_stress.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
timeout -k 10 300 python bench.py --workload stress --steps 10 --warmup 3 > $O/bench_strict_stress.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3c/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 4), "ms/step", round(d["value"]), d["unit"], d["config"].get("precision"), "fast_tier:", d.get("fast_tier"))
PY
