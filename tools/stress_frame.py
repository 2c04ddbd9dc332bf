"""Renders a few frames of the smooth-union stress scene; target of rocprofv3 runs (RM_W, RM_H, RM_STRESS_K)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raymarching_amd import _lib

if os.environ.get("RM_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["RM_LIB"])
from raymarching_amd.engine import Engine
from raymarching_amd.scene import Box, stress_lightmap_pattern, stress_scene

W, H = int(os.environ.get("RM_W", 7680)), int(os.environ.get("RM_H", 4320))  # bench.py --workload stress
scene = stress_scene(250, boxes=True, smooth_union=float(os.environ.get("RM_STRESS_K", "0.08")))
eng = Engine(scene, W, H)
eng.set_march_cap(1024)
if os.environ.get("RM_PATTERN") == "1":
    lm = stress_lightmap_pattern()
    for i, obj in enumerate(scene.objects):
        if not isinstance(obj, Box):
            eng.set_lightmap(i, lm)
else:
    eng.compute_lightmaps(4)  # as bench.py does, before the frames
eng.input(0, 0, (0, 0, 0), 0)
buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
for _ in range(int(os.environ.get("RM_FRAMES", 3))):
    eng.render_rows_device(buf.data_ptr(), 0, H)
eng.synchronize()
print("checksum", int(buf.sum().item()))
