"""How many tiles of the 1920x1080 default frame have a slowest quarter above a threshold: the number of
tiles the launch plan lists (rm_engine_split_tiles) for a range of RM_SPLIT_US, the relative threshold
switched off.  One engine per threshold (the environment is read when the first plan is made)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raymarching_amd import Engine, default_scene

W, H = int(os.environ.get("RM_W", 1920)), int(os.environ.get("RM_H", 1080))
os.environ["RM_SPLIT_TIMES_MEAN"] = "0"
os.environ["RM_PLAN_EVERY"] = "1"
buf = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
for us in [int(x) for x in os.environ.get("RM_THRESHOLDS", "45,55,65,70,75,80,85,90,95,100").split(",")]:
    os.environ["RM_SPLIT_US"] = str(us)
    e = Engine(default_scene(), W, H)
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    e.compute_lightmaps(4)
    e.input(0, 0, (0, 0, 0), 0)
    counts = []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(12):
        if i == 6:
            ev0.record()
        e.render_rows_device(buf.data_ptr(), 0, H)
        if i >= 6:
            counts.append(e.split_tiles())
    ev1.record()
    torch.cuda.synchronize()
    print(f"threshold {us:3d} us: tiles listed in frames 7-12: {counts}")
    e.close()
