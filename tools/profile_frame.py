"""Renders a few 4K frames of the default scene; the target of rocprofv3 runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import _lib, default_scene

if os.environ.get("RM_LIB"):  # another build of the library (ablation and diagnostic variants)
    _lib.LIB_PATH = os.path.abspath(os.environ["RM_LIB"])
from raymarching_amd.engine import Engine

W, H = int(os.environ.get("RM_W", 3840)), int(os.environ.get("RM_H", 2160))
frames = int(os.environ.get("RM_FRAMES", 5))
eng = Engine(default_scene(), W, H)
eng.compute_lightmaps(4)
eng.input(0, 0, (0, 0, 0), 0)
buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
for _ in range(frames):
    eng.render_rows_device(buf.data_ptr(), 0, H)
eng.synchronize()
print("checksum", int(buf.sum().item()))
