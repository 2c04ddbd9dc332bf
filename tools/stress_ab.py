"""Times the smooth-union stress frame (bench.py --workload stress: 250 spheres and boxes + the room,
k = 0.08, march cap 1024) for one or more builds of librm_hip.so in one process, interleaved, and
checks that every build renders the same bytes.
usage: stress_ab.py [lib.so ...]   (RM_W, RM_H: frame size, default 3840x2160)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import _lib
from raymarching_amd.scene import Box, stress_lightmap_pattern, stress_scene

W, H = int(os.environ.get("RM_W", 3840)), int(os.environ.get("RM_H", 2160))
libs = sys.argv[1:] or [_lib.LIB_PATH]
scene = stress_scene(250, boxes=True, smooth_union=0.08)
lm = stress_lightmap_pattern()
engines = []
for path in libs:
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    from raymarching_amd.engine import Engine

    e = Engine(scene, W, H)
    for i, obj in enumerate(scene.objects):
        if not isinstance(obj, Box):
            e.set_lightmap(i, lm)
    e.set_march_cap(1024)
    e.input(0, 0, (0.0, 0.0, 0.0), 0)
    engines.append(e)
bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in engines]
for e, b in zip(engines, bufs):
    e.render_rows_device(b.data_ptr(), 0, H)
    e.synchronize()
times = [[] for _ in engines]
for r in range(int(os.environ.get("RM_ROUNDS", 5))):
    for i, (e, b) in enumerate(zip(engines, bufs)):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        e.render_rows_device(b.data_ptr(), 0, H)
        ev1.record()
        torch.cuda.synchronize()
        times[i].append(ev0.elapsed_time(ev1))
for path, t, b, e in zip(libs, times, bufs, engines):
    t.sort()
    same = bool(torch.equal(b, bufs[0]))
    print(f"{os.path.basename(path):32s} median {t[len(t) // 2]:8.3f} ms  min {t[0]:8.3f} ms  {W * H / t[len(t) // 2] / 1e3:7.0f} Mrays/s  "
          f"{'same bytes' if same else 'DIFFERENT BYTES'}  {e.stats()}", flush=True)
