"""Loop-iteration statistics of a -DRM_DIAG build: python tools/diag_iters.py tools/librm_diag.so"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

from raymarching_amd import _lib, default_scene
from raymarching_amd.scene import Box, stress_scene

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from raymarching_amd.engine import Engine

W, H = int(os.environ.get("RM_W", 3840)), int(os.environ.get("RM_H", 2160))
if os.environ.get("RM_STRESS"):
    import numpy as np

    scene = stress_scene(250, boxes=True, smooth_union=float(os.environ["RM_STRESS"]))
    e = Engine(scene, W, H)
    lm = np.full((8, 8, 3), 0.1, dtype=np.float32)
    for i, obj in enumerate(scene.objects):
        if not isinstance(obj, Box):
            e.set_lightmap(i, lm)
else:
    e = Engine(default_scene(), W, H)
    e.compute_lightmaps(4)
e.reset_stats()
e.render(None, 0, 0, (0, 0, 0), 0)
lib = _lib.load()
import ctypes

raw = (ctypes.c_uint64 * 32)()
lib.rm_diag_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
assert lib.rm_diag_read(e.handle, raw) == 0
waves = (W // 8) * (H // 8)
it, mx, lanes = raw[4], raw[5], raw[6]
print(f"waves {waves}: loop iterations per wave {it / waves:.1f}, busiest lane {mx / waves:.1f}, mean lane {lanes / waves / 64:.1f}")
print(f"lane utilisation now {lanes / 64 / it:.3f}; one merged loop per pixel would need >= {mx / it:.3f} of the iterations")
names = ["setup", "march loops", "closest_index", "shade (colour, normal, lightmap, phong)", "shadow march", "reflection bookkeeping", "tone map + store"]
laps = [raw[8 + k] for k in range(len(names))]
tot = sum(laps)
for n, c in zip(names, laps):
    print(f"  {n:42s} {c / waves:9.1f} cycles/wave  {100.0 * c / tot:5.1f} %")
kinds = ["", "march-loop iterations", "shadow-loop iterations", "  of both: with the sphere group evaluated", "  shadow: with the penumbra division", "marches (primary + reflections)",
         "steps through the spatial index", "  of them: settled by a full walk"]
for k in range(1, len(kinds)):
    print(f"  {kinds[k]:46s} {raw[16 + k] / waves:7.2f} per wave")
print(f"  {'  of them: the short lists left the value open':46s} {raw[16] / waves:7.2f} per wave")
print(f"  lanes in use per march-loop iteration {raw[15] / max(raw[17], 1):.1f} of 64, per shadow-loop iteration {raw[31] / max(raw[18], 1):.1f} of 64")
print(f"  longest wave {raw[24] / 100:.1f} us, mean wave {raw[25] / waves / 100:.2f} us, grid span {(raw[27] - ((~raw[26]) & 0xFFFFFFFFFFFFFFFF)) / 100:.1f} us")
print(f"  waves longer than 20 us: {raw[28]}, 50 us: {raw[29]}, 100 us: {raw[30]}")
