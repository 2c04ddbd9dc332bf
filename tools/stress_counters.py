"""Algorithmic flops of the stress frame (bench.py --workload stress) from the CPU oracle's counting
build: eight rows of the 7680x4320 frame - the rows tests/test_bench_sizes_gpu.py holds the GPU
against - rendered with every event counted.  The oracle folds all 256 primitives at every step, as
the reference's loop over its objects does (ray.rs:19-40); the spatial index of the GPU kernel skips
most of that work, so these are the flops of the algorithm, not of the kernel.  Writes the counters
into tests/golden/counters.json ("stress_rows_7680x4320").  CPU only; about a minute on 8 threads."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from raymarching_amd.flops import frame_flops  # noqa: E402
from raymarching_amd.scene import Box, stress_lightmap_pattern, stress_scene  # noqa: E402

W, H = 7680, 4320
ROWS = [5, 611, 1500, 2160, 2701, 3333, 3900, 4319]
scene = stress_scene(250, boxes=True, smooth_union=0.08)
e = oracle.OracleEngine(scene, counting=True)
lm = stress_lightmap_pattern()
for i, obj in enumerate(scene.objects):
    if not isinstance(obj, Box):
        e.set_lightmap(i, lm)
e.reset_counters()
for y in ROWS:
    e.render(W, H, rows=(y, y + 1), threads=min(os.cpu_count() or 1, 16))
c = e.counters()
e.close()
c["rows"] = ROWS
c["note"] = "eight rows of the frame (0.19 % of its pixels), pattern lightmaps, march cap off (no march of the oracle reaches 1024 steps here)"
path = os.path.join(ROOT, "tests", "golden", "counters.json")
with open(path) as f:
    all_counters = json.load(f)
all_counters[f"stress_rows_{W}x{H}"] = c
with open(path, "w") as f:
    json.dump(all_counters, f, indent=1, sort_keys=True)
flops = frame_flops(c, smooth_union=True)
print(f"{c['pixels']} pixels, {flops / c['pixels']:.0f} algorithmic flops per primary ray, "
      f"{c['march_steps'] / c['pixels']:.1f} march steps and {c['shadow_steps'] / c['pixels']:.1f} shadow steps per pixel, "
      f"deepest reflection chain {c['max_depth']}, longest march {c['max_march_steps']} steps")
